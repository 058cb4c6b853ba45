// The decoder projection kernel body (see decoder32.hip for the design notes) as a device function, so that the stand-alone kernel
// (decoder32.hip) and the fused projection + attention launches (decoder_fused.hip) share one implementation - and one set of bits.
#pragma once
#include "dec_shared.h"

namespace wh {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------- in-launch hand-off (producer side)
// A fused launch (decoder_fused.hip) runs a projection's workgroups and the attention workgroups that consume its output in ONE grid:
// the producer publishes with write-through (sc1) stores, drains them, and bumps the arrival counter of the (batch tile, head) its row
// tile belongs to; the attention workgroups of that head poll the counter (MI355X_MICROARCH.md "handoff-flag", recipe R1: sc1 payload
// -> vmcnt(0) -> relaxed agent-scope flag; the consumer reads the payload with sc1 loads).  Without a consumer (signal == nullptr) the
// stores are the plain ones of the stand-alone kernel.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(float* p, f32x4 v, bool write_through) {
    if (write_through) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store8(f16* p, f16x4 v, bool write_through) {
    if (write_through) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<f16x4*>(p) = v;
}
__device__ __forceinline__ void signal_head(int* counter, int tid) {      // workgroup-uniform call
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------- residual tail
// Shared by the RESID finisher and the embedding kernel: thread (slot j, channels n..n+3) holds the new residual values.
// Stores x, the planes z = gamma_next * x for the next LayerNorm consumer, and this row tile's (mean, M2) per slot.
__device__ __forceinline__ void d32_resid_tail(const float (&xn)[4], bool valid, int bt, int rt, int n_rt, int n, int j, int gb, int tid,
                                               int d, float* x, const float* gamma_next, f16* zhi, f16* zlo, float2* stat_out,
                                               float (*xs)[33]) {
    const float4 gm = *reinterpret_cast<const float4*>(gamma_next + n);
    if (valid) {
        *reinterpret_cast<float4*>(x + (size_t)gb * d + n) = float4{xn[0], xn[1], xn[2], xn[3]};
        const float z[4] = {gm.x * xn[0], gm.y * xn[1], gm.z * xn[2], gm.w * xn[3]};
        f16x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { f16 h_, l_; split_hilo(z[i], h_, l_); hi[i] = h_; lo[i] = l_; }
        const size_t o = plane_index(gb, n, d);
        *reinterpret_cast<f16x4*>(zhi + o) = hi;
        *reinterpret_cast<f16x4*>(zlo + o) = lo;
    }
    const int nl = n & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) xs[nl + i][j] = xn[i];
    __syncthreads();
    if (tid < 32) {      // slot tid: two-pass statistics of this tile's 32 channels, fixed order
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += xs[r][tid];
        const float mean = s * (1.0f / 32.0f);
        float m2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) { const float e = xs[r][tid] - mean; m2 = fmaf(e, e, m2); }
        stat_out[((size_t)bt * n_rt + rt) * 32 + tid] = float2{mean, m2};
    }
}

// Chan's pairwise update of (count, mean, M2).  chan32_k: the k-th 32-sample partial (mean_b, M2_b) joins k earlier ones, so the
// weights 32 / n and n_a 32 / n are the constants 1 / (k + 1) and 32 k / (k + 1) - no division on the dependent chain.
template <int K>
__device__ __forceinline__ void chan32_k(float& cm, float& cM2, float mb, float M2b) {
    constexpr float w = 1.0f / (float)(K + 1), w2 = 32.0f * (float)K / (float)(K + 1);
    const float delta = mb - cm;
    cm = fmaf(delta, w, cm);
    cM2 += fmaf(delta * delta, w2, M2b);
}
__device__ __forceinline__ void chan_merge(float& cn, float& cm, float& cM2, float nb, float mb, float M2b) {
    if (nb == 0.0f) return;
    const float nn = cn + nb, rn = __frcp_rn(nn);
    const float delta = mb - cm;
    cm = fmaf(delta, nb * rn, cm);
    cM2 += fmaf(delta * delta, cn * nb * rn, M2b);
    cn = nn;
}

// ---------------------------------------------------------------------------------------------- the projection kernel
// Weights and planes stream in double-buffered chunks of TC k-tiles (tw is a multiple of TC); sched_barriers keep the loads of the
// next chunk ahead of the current chunk's MFMAs.  Measured and rejected: requesting a wave's whole weight slab (20 tiles, 80 VGPRs) in
// one burst ahead of the MFMAs - 2-3 % slower at 8 and 32 slots (profiles/r02j_*): the per-launch latency is not the weight round trips.
// __launch_bounds__(256, 2): capping the wave at 256 unified registers keeps the accumulators in arch VGPRs (with 512 allowed the
// compiler parked them in AccVGPRs and copied all 32 of them out and back in every loop iteration: 160 v_accvgpr moves per launch);
// every instantiation fits (88 - 204 VGPRs, no scratch), two workgroups can share a CU.
template <int MODE, bool HILO, int TC>
__device__ __forceinline__ void dec32_proj_body(const P32Args& a, const int block_id) {
    constexpr bool kLN = MODE == P32_QKV || MODE == P32_Q || MODE == P32_FC1 || MODE == P32_LOGITS;
    __shared__ float red[4][16][64];                 // the four waves' partial tiles
    __shared__ float st_l[8][32][3];                 // LayerNorm statistics: 8 partial (n, mean, M2) per slot
    __shared__ float xs_raw[MODE == P32_LOGITS ? 256 * 6 : 32 * 33];   // RESID: the tile's new residual values; LOGITS: sampler records
    float (*xs)[33] = reinterpret_cast<float (*)[33]>(xs_raw);
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_rt = (a.N + 31) >> 5;
    // Workgroup id -> (row tile, K slice, batch tile): ids x + 8 t of one group of 8 share the weight slab x and differ in the batch
    // tile t, so the readers of a slab are dispatched back to back onto the SAME XCD (id % 8) and the slab crosses HBM once, whatever
    // the parity of the tile count (PMC at 64 slots before this: the 1621 logits tiles fetched 273 MB for 133 MB of weights).
    const int grp8 = block_id / (8 * a.n_bt), in8 = block_id % (8 * a.n_bt);
    const int xw = grp8 * 8 + (in8 & 7), bt = in8 >> 3;
    // re-arm the arrival flags of the in-launch hand-off of the kernel BEFORE this one (decoder_fused.hip): it is complete (kernel boundary)
    if (a.clear_flags && block_id == 0 && tid < a.n_clear) __hip_atomic_store(a.clear_flags + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (xw >= n_rt * a.ks) return;          // padding of the last group (workgroup-uniform)
    const int rt = xw % n_rt, ksi = xw / n_rt;
    const int KT = a.K >> 4;
    const int kt0 = (ksi * 4 + wave) * a.tw;
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.Wt) + ((size_t)rt * KT + kt0) * 64 + lane;
    const size_t zoff = ((size_t)bt * KT + kt0) * 64 + lane;
    const u32x4* hp = reinterpret_cast<const u32x4*>(a.zhi) + zoff;
    const u32x4* lp = HILO ? reinterpret_cast<const u32x4*>(a.zlo) + zoff : nullptr;
    // epilogue coordinates of this thread: slot j, channels n .. n + 3 (the accumulator rows 4 wave + i of half-wave h)
    const int j = tid & 31, sub = tid >> 5;
    const int n = rt * 32 + 4 * sub;
    const int gb = bt * 32 + j;
    const bool valid = gb < a.batch;

#define D32_STAMP(i) do { if (a.dbg && tid == 0 && bt == 0) a.dbg[(size_t)(xw & 4095) * 8 + (i)] = wall_clock64(); } while (0)
    D32_STAMP(0);
    // ---- small epilogue operands, requested first (memory returns are in order per wave: they arrive under the weight stream)
    float2 sp[5] = {};
    if constexpr (kLN) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int idx = min(sub + 8 * i, a.n_stat - 1);
            sp[i] = a.stat_in[((size_t)bt * a.n_stat + idx) * 32 + j];
        }
    }
    float4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};    // LN modes: g, c;  RESID: bias, old x
    int pos_l = 0, live_l = 0;
    if constexpr (kLN) {
        e0 = *reinterpret_cast<const float4*>(a.fold_g + n);
        e1 = *reinterpret_cast<const float4*>(a.fold_c + n);
    } else {
        e0 = *reinterpret_cast<const float4*>(a.bias + n);
        e1 = *reinterpret_cast<const float4*>(a.x + (size_t)gb * a.d + n);
    }
    if (valid) { live_l = slot_live(a.seq + gb); if constexpr (MODE == P32_QKV) pos_l = a.seq[gb].token_index; }
    int rules[6] = {0, 0, 0, 0, 0, 0};
    unsigned masked4 = 0xffffffffu;
    int tb = 0, ws_tok = 0, eot_tok = 0, nots_tok = 0, r16 = 0;
    if constexpr (MODE == P32_LOGITS) {
        if (a.cfg) r16 = a.cfg->f16_logits;
        if (a.stats) {
            if (valid) {
#pragma unroll
                for (int i = 0; i < 6; ++i) rules[i] = a.seq[gb].f_rules[i];
            }
            if (n + 3 < a.N) masked4 = *reinterpret_cast<const unsigned*>(a.sup_mask + n);
            else {
                masked4 = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) masked4 |= (unsigned)(n + i < a.N ? a.sup_mask[n + i] : 1) << (8 * i);
            }
            tb = a.cfg->time_token_begin; ws_tok = a.cfg->whitespace_token; eot_tok = a.cfg->end_token; nots_tok = a.cfg->no_timestamps_token;
        }
    }

    // ---- weight stream x activation planes on the matrix cores
    f32x16 acc_h = {0}, acc_l = {0};
    auto stats_to_lds = [&]() {
        // LayerNorm statistics: each thread Chan-combines its <= 5 row-tile partials (ascending), the 8 threads of a slot meet in LDS
        if constexpr (kLN) {
            float cm = sp[0].x, cM2 = sp[0].y;                     // n_stat >= 8 is not required: a thread without partials writes count 0
            const int mine = sub < a.n_stat ? (a.n_stat - sub + 7) >> 3 : 0;
            if (mine > 1) chan32_k<1>(cm, cM2, sp[1].x, sp[1].y);
            if (mine > 2) chan32_k<2>(cm, cM2, sp[2].x, sp[2].y);
            if (mine > 3) chan32_k<3>(cm, cM2, sp[3].x, sp[3].y);
            if (mine > 4) chan32_k<4>(cm, cM2, sp[4].x, sp[4].y);
            st_l[sub][j][0] = 32.0f * (float)mine; st_l[sub][j][1] = cm; st_l[sub][j][2] = cM2;
        }
    };
    {
        u32x4 wa[TC], ha[TC], la[HILO ? TC : 1], wb[TC], hb[TC], lb[HILO ? TC : 1];
        auto ld = [&](u32x4 (&w)[TC], u32x4 (&h)[TC], u32x4 (&l)[HILO ? TC : 1], int c) {
#pragma unroll
            for (int i = 0; i < TC; ++i) w[i] = __builtin_nontemporal_load(wp + (size_t)(c * TC + i) * 64);     // streamed once: nt
#pragma unroll
            for (int i = 0; i < TC; ++i) h[i] = hp[(size_t)(c * TC + i) * 64];
            if constexpr (HILO) {
#pragma unroll
                for (int i = 0; i < TC; ++i) l[i] = lp[(size_t)(c * TC + i) * 64];
            }
        };
        auto mm = [&](const u32x4 (&w)[TC], const u32x4 (&h)[TC], const u32x4 (&l)[HILO ? TC : 1]) {
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const f16x8 wf = __builtin_bit_cast(f16x8, w[i]);
                acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(f16x8, h[i]), acc_h, 0, 0, 0);
                if constexpr (HILO) acc_l = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(f16x8, l[i]), acc_l, 0, 0, 0);
            }
        };
        const int nch = a.tw / TC;
        ld(wa, ha, la, 0);
        D32_STAMP(1);
#pragma unroll 1
        for (int c = 0; c < nch; c += 2) {
            if (c + 1 < nch) ld(wb, hb, lb, c + 1);
            __builtin_amdgcn_sched_barrier(0);      // loads of the next chunk stay ahead of this chunk's MFMAs
            mm(wa, ha, la);
            if (c == 0) D32_STAMP(2);
            if (c + 2 < nch) ld(wa, ha, la, c + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nch) mm(wb, hb, lb);
        }
    }
    stats_to_lds();       // after the stream: the statistics are epilogue operands (timeline probe: waiting for them up front cost 1 us per launch)
    D32_STAMP(3);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = HILO ? fmaf(acc_l[r], 1.0f / 2048.0f, acc_h[r]) : acc_h[r];
    __syncthreads();
    D32_STAMP(4);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((red[0][4 * wave + i][lane] + red[1][4 * wave + i][lane]) + red[2][4 * wave + i][lane]) + red[3][4 * wave + i][lane];

    // ---- K split across workgroups: publish, ticket, the last arriver sums the slices in index order.  Write-through (sc1)
    // 16-byte stores, a drained vmcnt in every storing wave, one relaxed ticket; the finisher reads the slabs with sc1 loads, which
    // bypass its L1 and are served by L2 - no agent-scope fence on either side (MI355X_MICROARCH.md "handoff-flag", R1).
    if (a.ks > 1) {
        float* base = a.part + (((size_t)bt * n_rt + rt) * a.ks) * 1024 + tid * 4;
        {
            const f32x4 pv4 = {v[0], v[1], v[2], v[3]};
            float* mine = base + (size_t)ksi * 1024;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine), "v"(pv4) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int* cnt = a.ticket + bt * n_rt + rt;
            const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == a.ks - 1);
            if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next launch
            last_flag = last;
        }
        __syncthreads();
        if (!last_flag) return;             // workgroup-uniform
        float pv[8][4];
#pragma unroll
        for (int s = 0; s < 8; ++s) {       // every load is issued before the first add; slices past ks re-read slice 0 and are dropped
            const float* p = base + (size_t)(s < a.ks ? s : 0) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[s][i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = pv[0][i];
#pragma unroll
            for (int s = 1; s < 8; ++s) t += (s < a.ks) ? pv[s][i] : 0.0f;
            v[i] = t;
        }
    }

    D32_STAMP(5);
    // ---- epilogues
    float y[4];
    if constexpr (kLN) {
        float cn = 0.0f, cm = 0.0f, cM2 = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) chan_merge(cn, cm, cM2, st_l[s][j][0], st_l[s][j][1], st_l[s][j][2]);
        const float mu = cm, rstd = rsqrtf(cM2 / (float)a.d + 1e-5f);
        const float g4[4] = {e0.x, e0.y, e0.z, e0.w}, c4[4] = {e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = fmaf(rstd, v[i] - mu * g4[i], c4[i]);
    }
    if constexpr (MODE == P32_QKV) {
        if (valid && live_l) {
            const int d = a.d;
            if (n < d) store16(a.q + (size_t)gb * d + n, f32x4{y[0], y[1], y[2], y[3]}, a.signal != nullptr);
            else {
                int c = n - d;
                f16* dst = a.self_k;
                if (c >= d) { c -= d; dst = a.self_v; }
                const int pos = min(max(pos_l, 0), kMaxTok - 1);
                store8(dst + (((size_t)gb * a.n_head + (c >> 6)) * kMaxTok + pos) * kHeadDim + (c & 63), f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]},
                       a.signal != nullptr);
            }
        }
        if (a.signal) signal_head(a.signal + bt * a.n_head + ((n % a.d) >> 6), tid);
    } else if constexpr (MODE == P32_Q) {
        if (valid && live_l) store16(a.q + (size_t)gb * a.d + n, f32x4{y[0], y[1], y[2], y[3]}, a.signal != nullptr);
        if (a.signal) signal_head(a.signal + bt * a.n_head + (n >> 6), tid);
    } else if constexpr (MODE == P32_FC1) {
        if (valid)
            *reinterpret_cast<f16x4*>(a.h_out + plane_index(gb, n, a.N)) =
                f16x4{(f16)gelu_erf(y[0]), (f16)gelu_erf(y[1]), (f16)gelu_erf(y[2]), (f16)gelu_erf(y[3])};
    } else if constexpr (MODE == P32_RESID) {
        const float b4[4] = {e0.x, e0.y, e0.z, e0.w}, x4[4] = {e1.x, e1.y, e1.z, e1.w};
        float xn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xn[i] = x4[i] + (v[i] + b4[i]);
        d32_resid_tail(xn, valid && live_l, bt, rt, n_rt, n, j, gb, tid, a.d, a.x, a.gamma_next, a.zhi_out, a.zlo_out, a.stat_out, xs);
    } else {    // P32_LOGITS
        if (r16) {      // reference-numerics switch: the TextDecoder output is a Float16 array (Core/Models.swift:1041)
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (float)(f16)y[i];
        }
        if (a.logits && valid && live_l) {
            float* lo = a.logits + (size_t)gb * a.N + n;
            if (n + 3 < a.N) {
                *reinterpret_cast<float2*>(lo) = float2{y[0], y[1]};
                *reinterpret_cast<float2*>(lo + 2) = float2{y[2], y[3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n + i < a.N) lo[i] = y[i];
            }
        }
        if (a.stats) {
            // fused greedy sampler, part 1 (decoder.hip logits_block_stats): the index-predicate filters of LogitsFilter.swift on this
            // thread's 4 ids, then (max, sum exp, argmax) separately for text and timestamp ids; the 8 threads of a slot meet in LDS
            SoftStat t{-INFINITY, 0.0f, 0x7fffffff}, u{-INFINITY, 0.0f, 0x7fffffff};
            const int blank = rules[0], ts_active = rules[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = n + i;
                bool masked = ((masked4 >> (8 * i)) & 0xff) != 0 || id >= a.N;                                   // SuppressTokensFilter
                masked |= blank && (id == ws_tok || id == eot_tok);                                               // SuppressBlankFilter
                masked |= ts_active && (id == nots_tok || (id >= rules[2] && id < rules[3]) || (id >= rules[4] && id < rules[5]));   // TimestampRulesFilter
                if (!masked) { if (id < tb) stat_merge(t, y[i], 1.0f, id); else stat_merge(u, y[i], 1.0f, id); }
            }
            float* rec = xs_raw + (size_t)(sub * 32 + j) * 6;
            rec[0] = t.m; rec[1] = t.s; rec[2] = __int_as_float(t.i); rec[3] = u.m; rec[4] = u.s; rec[5] = __int_as_float(u.i);
            __syncthreads();
            if (tid < 32 && valid && live_l) {
                SoftStat T{-INFINITY, 0.0f, 0x7fffffff}, U{-INFINITY, 0.0f, 0x7fffffff};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float* r_ = xs_raw + (size_t)(s * 32 + tid) * 6;
                    stat_merge(T, r_[0], r_[1], __float_as_int(r_[2]));
                    stat_merge(U, r_[3], r_[4], __float_as_int(r_[5]));
                }
                float* o = a.stats + ((size_t)gb * kStatBlocks + rt) * 8;
                *reinterpret_cast<float4*>(o) = float4{T.m, T.s, __int_as_float(T.i), U.m};
                *reinterpret_cast<float2*>(o + 4) = float2{U.s, __int_as_float(U.i)};
            }
        }
    }
    D32_STAMP(6);
}


}  // namespace wh
