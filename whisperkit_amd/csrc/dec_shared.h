// Device helpers shared by the two decoder paths (decoder.hip: GEMV kernels, attention, samplers; decoder32.hip: MFMA projections).
#pragma once
#include "kernels.h"

namespace wh {

__device__ __forceinline__ bool slot_live(const SeqState* s) { return s->active && !s->done; }

// running (max, sum exp(x - max), argmax) of a softmax, mergeable in any fixed order; equal maxima keep the smaller index
struct SoftStat { float m, s; int i; };
__device__ __forceinline__ void stat_merge(SoftStat& a, float em, float es, int ei) {
    if (em == -INFINITY) return;
    if (a.m == -INFINITY || em > a.m) {
        a.s = (a.m == -INFINITY ? 0.0f : a.s * __expf(a.m - em)) + es;
        a.m = em; a.i = ei;
    } else if (em == a.m) {
        a.s += es; a.i = min(a.i, ei);
    } else {
        a.s += es * __expf(em - a.m);
    }
}

// "timestamp mass beats every text token" (TimestampRulesFilter, LogitsFilter.swift:144-242) from the merged softmax statistics of
// the unmasked text ids (t) and timestamp ids (u).  fp32: logsumexp(ts) > max(text).  Float16 mode: the reference takes logSoftmax
// over all logits, then logSumExp(ts) and max(text), each a FloatType (Float16) value - emulated as the fp32 quantities relative to
// the common log-sum-exp, rounded to Float16 before the comparison (BNNS' internal precision is not specified).
__device__ __forceinline__ bool timestamp_mass_wins(const SoftStat& t, const SoftStat& u, bool f16_mode) {
    if (u.m == -INFINITY) return false;
    const float ts = u.m + logf(u.s);
    if (!f16_mode) return ts > t.m;
    if (t.m == -INFINITY) return true;
    SoftStat g = t;
    stat_merge(g, u.m, u.s, u.i);
    const float lse = g.m + logf(g.s);
    return (float)(f16)(ts - lse) > (float)(f16)(t.m - lse);
}

// element (slot b, channel n) of an activation plane of K channels: Z[b / 32][n / 16][(n / 8) & 1][b & 31][n & 7]
__device__ __forceinline__ size_t plane_index(int b, int n, int K) {
    return ((((size_t)(b >> 5) * (K >> 4) + (n >> 4)) * 2 + ((n >> 3) & 1)) * 32 + (b & 31)) * 8 + (n & 7);
}
// f32 -> f16 hi + scaled f16 lo (z ~ hi + lo / 2048, 22 mantissa bits; the scale keeps lo out of the f16 subnormals)
__device__ __forceinline__ void split_hilo(float z, f16& hi, f16& lo) {
    hi = (f16)z;
    lo = (f16)((z - (float)hi) * 2048.0f);
}


// Cross-attention gate (scheduling only, never correctness): sessions of one model that decode concurrently take turns at the ONE
// kernel of a decoder layer that saturates the HBM.  Free-running streams fall into convoys - their cross-attention launches overlap
// (each then takes twice as long), finish together, and then all of them run their latency-bound projection chains with the HBM idle
// (kernel trace of three sessions in flight, profiles/r03g_inflight_overlap.txt: no cross-attention kernel running 34 - 39 % of the
// time).  The cross-query projection's workgroup 0 takes the gate before it exits (so the session's cross-attention launch starts
// behind it), the LAST workgroup of the cross-attention grid gives it back when it is dispatched (the grid is draining from there:
// the next session's launch ramps up under the tail).  A bounded wait: streams that share a hardware queue cannot deadlock, they
// only lose the staggering.
constexpr unsigned long long kGateTimeoutTicks = 50000;     // 500 us of the 100 MHz wall clock
__device__ __forceinline__ void xattn_gate_acquire(int* g) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        int expected = 0;
        if (__hip_atomic_compare_exchange_strong(g, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (wall_clock64() - t0 > kGateTimeoutTicks) { __hip_atomic_fetch_add(g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        __builtin_amdgcn_s_sleep(16);
    }
}
__device__ __forceinline__ void xattn_gate_release(int* g) { __hip_atomic_fetch_add(g, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace wh
