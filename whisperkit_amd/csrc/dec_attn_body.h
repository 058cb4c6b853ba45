// Decoder attention bodies (self-attention over the cached positions, flash-decoding split cross-attention) as device functions:
// decoder.hip wraps them into the stand-alone kernels, decoder_fused.hip into launches that also carry the projection whose
// output they consume (QKV -> self-attention, cross query -> cross-attention) with an in-launch hand-off per (batch tile, head).
// Replaces the attention inside the per-token CoreML TextDecoder call (Sources/WhisperKit/Core/TextDecoder.swift:381-418) and
// updateAlignmentWeights (:272-296).
#pragma once
#include "dec_shared.h"

namespace wh {

struct AttnArgs {
    int batch, d, n_head, layer, n_layer, n_split;
    const float* q;          // [B][d]
    const f16* self_k; const f16* self_v;     // layer base [Bmax][H][224][64]
    const f16* cross_k; const f16* cross_v;   // layer base [Bmax][H][1500][64]
    f16 *att_hi, *att_lo;    // attention output (before the out projection) as an f16 hi | lo pair in B-fragment plane order (decoder32.hip)
    float* part;             // [B][H][n_split][kPartStride]: (m, l, o[64]) of every key split, one 128-byte-aligned slot each
    int* ticket;             // [B][H] arrival counters (zero between launches)
    float* align; const int* align_slot; int n_align;   // [B][224][n_align][1500] raw score rows of the alignment heads
    SeqState* seq;
    int no_fence;
    unsigned long long* dbg;   // optional timeline probe (WH_DBG=1)
    // in-launch hand-off (fused launches only): arrival counters [n_bt][n_head] the projection's finishers bump, the count that
    // means "this head's rows are published", and the sticky give-up word of the bounded spin
    const int* ready; int ready_need; int* poison;
};

__device__ __forceinline__ void store_att(const AttnArgs& a, int b, int n, float v) {
    f16 hi, lo;
    split_hilo(v, hi, lo);
    const size_t o = plane_index(b, n, a.d);
    a.att_hi[o] = hi;
    a.att_lo[o] = lo;
}

// Consumer side of the hand-off: ONE lane polls the (batch tile, head) counter with relaxed agent-scope (sc1) loads, the workgroup
// meets at a barrier, the payload is then read with sc1 loads (L1 bypassed; the producer stored write-through).  The spin is bounded:
// a counter that never arrives (a producer that was never dispatched would be a bug, not a schedule - producers carry the lowest
// workgroup ids of the launch) sets a sticky poison word that makes every later poll of the session return at once; the host reports
// it as an error after the step (capi / host.hip) - a wrong result, never a hung GPU.
__device__ __forceinline__ void wait_ready(const int* counter, int need, int* poison) {
    if (threadIdx.x == 0) {
        bool ok = false;
        for (int it = 0; it < (1 << 21); ++it) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = true; break; }
            if ((it & 255) == 255 && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) __hip_atomic_store(poison, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// One query against keys [t0, t0 + n) of a head-major K/V block (rows of 64 halves).  Thread layout: 8 lanes per
// key (16 bytes = 8 channels each), 32 keys per pass, PASSES passes; all K and V rows of the block are in flight
// before the first use.  Returns this block's softmax statistics (m, l) and leaves the unnormalised output
// o[64] = sum_t exp(s_t - m) V[t] in o_out (LDS, valid for tid < 64).  raw_scores (optional, global) gets s_t.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 load_kv16(const f16* p) {      // 16 bytes of a K / V row; NT: non-temporal (streamed once per step)
    if constexpr (NT) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return uint4{v[0], v[1], v[2], v[3]};
    } else return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ uint4 load_kv16_sc1(const f16* p) {  // the same through L2 (a row another workgroup of this launch published)
    uint4 v;
    const unsigned* w = reinterpret_cast<const unsigned*>(p);
    v.x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.w = __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// HANDOFF: the rows [0, n_load) are fetched first (they do not depend on this launch), then the workgroup waits for the producer
// (`wait`), then row `late_key` (< 0: none) and the query are read through L2.
template <int PASSES, bool NT, bool HANDOFF, typename GetN, typename Wait>
__device__ __forceinline__ bool attend_block(const float* __restrict__ qg, const f16* __restrict__ kb, const f16* __restrict__ vb, int n_load,
                                             GetN get_n, Wait wait, int late_key, float* const* raw_pp, float* red /* [16] */,
                                             float* osum /* [4][64] */, float* o_out /* [64] */, float* m_out, float* l_out,
                                             unsigned long long* stamp = nullptr) {
    // n_load rows are FETCHED right away; how many of them count (n = get_n(), < 0: slot not live) is only looked at
    // afterwards, so the slot-state loads and the K/V stream share one memory round trip instead of two.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = tid & 7, kg = tid >> 3;
    uint4 kreg[PASSES], vreg[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = kg + 32 * i;
        kreg[i] = key < n_load ? load_kv16<NT>(kb + (size_t)key * kHeadDim + part * 8) : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = kg + 32 * i;
        vreg[i] = key < n_load ? load_kv16<NT>(vb + (size_t)key * kHeadDim + part * 8) : uint4{0, 0, 0, 0};
    }
    float qv[8];
    if constexpr (!HANDOFF) {
        float4 q0 = *reinterpret_cast<const float4*>(qg + part * 8);
        float4 q1 = *reinterpret_cast<const float4*>(qg + part * 8 + 4);
        qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    }
    const int n = get_n();
    if (n < 0) return false;            // workgroup-uniform
    if constexpr (HANDOFF) {
        wait();                         // the producer's rows of this head are published (or the spin gave up: poison is set)
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = __hip_atomic_load(qg + part * 8 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            if (kg + 32 * i == late_key) {
                kreg[i] = load_kv16_sc1(kb + (size_t)late_key * kHeadDim + part * 8);
                vreg[i] = load_kv16_sc1(vb + (size_t)late_key * kHeadDim + part * 8);
            }
        }
    }
    float* raw_scores = *raw_pp;
    float s[PASSES];
    float lmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        f16x8 k8 = *reinterpret_cast<f16x8*>(&kreg[i]);
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t = fmaf((float)k8[j], qv[j], t);
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        const int key = kg + 32 * i;
        if (key < n) {
            if (raw_scores && part == 0) raw_scores[key] = t;
            lmax = fmaxf(lmax, t);
        } else t = -INFINITY;
        s[i] = t;
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    if (stamp && threadIdx.x == 0) stamp[2] = wall_clock64();
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.0f;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const bool valid = kg + 32 * i < n;
        const float p = valid ? __expf(s[i] - m) : 0.0f;
        if (part == 0) lsum += p;
        if (!valid) vreg[i] = uint4{0, 0, 0, 0};          // rows past n were fetched speculatively: keep 0 * garbage out
        f16x8 v8 = *reinterpret_cast<f16x8*>(&vreg[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(p, (float)v8[j], o[j]);
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = o[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        o[j] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) osum[wave * 64 + lane * 8 + j] = o[j];
    }
    __syncthreads();
    if (tid < 64) o_out[tid] = (osum[tid] + osum[64 + tid]) + (osum[128 + tid] + osum[192 + tid]);
    if (stamp && threadIdx.x == 0) stamp[3] = wall_clock64();
    *m_out = m;
    *l_out = (red[4] + red[5]) + (red[6] + red[7]);
    return true;
}

// ---- self-attention of (head h, slot b).  Stand-alone (HANDOFF = false): PASSES x 32 cached positions are FETCHED speculatively,
// before the slot state is known; the launcher passes the smallest bound that covers every live slot's position, so the cache traffic
// follows the decoded length (PMC, 32 slots at positions < 9: 37 MB fetched per launch with the fixed 7 passes against 1.5 MB needed).
// Fused with the QKV projection (HANDOFF = true): the slot state is read first, exactly the rows < token_index are fetched while the
// projection still streams its weights, and this step's row and the query are read once the head's six row tiles have arrived.
template <int PASSES, bool HANDOFF>
__device__ __forceinline__ void dec_self_attn_body(const AttnArgs& a, int h, int b) {
    __shared__ float red[16], osum[256], o_l[64];
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;     // stand-alone: looked at after the K/V loads are issued
    const int d = a.d;
    const size_t base = ((size_t)b * a.n_head + h) * kMaxTok * kHeadDim;
    float m, l;
    float* raw = nullptr;
    const int pos = min(max(s_ti, 0), kMaxTok - 1);
    auto get_n = [&]() { return (s_act && !s_done) ? min(pos + 1, PASSES * 32) : -1; };
    auto wait = [&]() { if constexpr (HANDOFF) wait_ready(a.ready + (b >> 5) * a.n_head + h, a.ready_need, a.poison); };
    const int n_load = HANDOFF ? ((s_act && !s_done) ? min(pos, PASSES * 32) : 0) : PASSES * 32;
    if (!attend_block<PASSES, false, HANDOFF>(a.q + (size_t)b * d + h * kHeadDim, a.self_k + base, a.self_v + base, n_load, get_n, wait,
                                              HANDOFF ? pos : -1, &raw, red, osum, o_l, &m, &l))
        return;
    if (threadIdx.x < 64) store_att(a, b, h * kHeadDim + threadIdx.x, o_l[threadIdx.x] / l);
}

// ---- cross-attention of (key split sp, head h, slot b); `stamp_slot`: linear workgroup index of the timeline probe
template <int PASSES, bool NT, bool HANDOFF>
__device__ __forceinline__ void dec_cross_attn_body(const AttnArgs& a, int sp, int h, int b, unsigned stamp_slot) {
    constexpr int KPB = PASSES * 32;
    __shared__ float red[16], osum[256], o_l[64];
    __shared__ int last_flag;
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;     // looked at after the K/V loads are issued
    const int d = a.d, S = a.n_split;
    const int t0 = sp * KPB, n = min(KPB, kCtx - t0);
    const size_t base = (((size_t)b * a.n_head + h) * kCtx + t0) * kHeadDim;
    int slot = -1;
    if (a.align) slot = a.align_slot[a.layer * a.n_head + h];
    float m, l;
    unsigned long long* stamp = a.dbg ? a.dbg + (size_t)(stamp_slot % 4096) * 8 : nullptr;
    if (stamp && threadIdx.x == 0) stamp[0] = wall_clock64();
    // alignment-head row: DecodingCache.alignmentWeights row tokenIndex + 1 (TextDecoder.swift:272-296), raw scores here,
    // softmax + head mean in alignment_mean_kernel
    float* raw = nullptr;
    auto get_n = [&]() {
        if (!(s_act && !s_done)) return -1;
        const int pos = min(max(s_ti, 0), kMaxTok - 1);
        if (slot >= 0 && pos + 1 < kMaxTok) raw = a.align + (((size_t)b * kMaxTok + pos + 1) * a.n_align + slot) * kCtx + t0;
        return n;
    };
    auto wait = [&]() { if constexpr (HANDOFF) wait_ready(a.ready + (b >> 5) * a.n_head + h, a.ready_need, a.poison); };
    if (!attend_block<PASSES, NT, HANDOFF>(a.q + (size_t)b * d + h * kHeadDim, a.cross_k + base, a.cross_v + base, n, get_n, wait, -1, &raw, red, osum,
                                           o_l, &m, &l, stamp))
        return;
    // ---- publish this split's partial, take a ticket; the last arriver combines all splits in index order
    const int tid = threadIdx.x;
    float* mine = a.part + (((size_t)b * a.n_head + h) * S + sp) * kPartStride;
    // write-through (sc1) stores + drained ticket: no per-workgroup L2 write-back (MI355X_MICROARCH.md "publish-large")
    if (tid < 64) __hip_atomic_store(mine + 2 + tid, o_l[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 64) {
        __hip_atomic_store(mine, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int* cnt = a.ticket + b * a.n_head + h;
        int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int last = (t == S - 1);
        if (last) {
            // agent-scope acquire on the combining CU: the partial slots are rewritten by every layer's launch, and a copy
            // left in this XCD's L2 by an earlier combine must not be served to the sc1 loads below (the recipe of
            // MI355X_MICROARCH.md: one relaxed ticket, one agent acquire).  WH_XATT_NOFENCE=1 drops it (A/B knob).
            if (!a.no_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        }
        last_flag = last;
    }
    __syncthreads();
    if (stamp && tid == 0) stamp[4] = wall_clock64();
    if (last_flag && tid == 0 && a.dbg) stamp[6] = 1;
    if (last_flag) {      // workgroup-uniform
        // all S partials (S x 66 floats) are fetched by the whole workgroup in ONE round of independent sc1 loads into LDS
        // (a per-thread loop over the splits is S dependent L2 round trips: 24 us at S = 24) and combined from there
        __shared__ float pl[kMaxSplit * 66];
        const float* p0 = a.part + ((size_t)b * a.n_head + h) * S * kPartStride;
        constexpr int NLD = (kMaxSplit * 66 + 255) / 256;
        float tmp[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {         // issue every load before the first use
            const int i = tid + 256 * k, sp_i = i / 66, e = i - sp_i * 66;
            tmp[k] = i < S * 66 ? __hip_atomic_load(p0 + sp_i * kPartStride + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;   // sc1
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            if (i < S * 66) pl[i] = tmp[k];
        }
        __syncthreads();
        if (tid < 64) {
            float mg = -INFINITY;
            for (int i = 0; i < S; ++i) mg = fmaxf(mg, pl[i * 66]);
            float lg = 0.0f, og = 0.0f;
            for (int i = 0; i < S; ++i) {
                const float w = __expf(pl[i * 66] - mg);
                lg = fmaf(w, pl[i * 66 + 1], lg);
                og = fmaf(w, pl[i * 66 + 2 + tid], og);
            }
            store_att(a, b, h * kHeadDim + tid, og / lg);
        }
    }
    if (stamp && tid == 0) stamp[5] = wall_clock64();
}

}  // namespace wh
