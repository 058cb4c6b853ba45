// Decoder projections on the matrix cores for batch tiles of 32 sequences (gfx950): the weight-streaming half of the token step
// that replaces the per-token CoreML TextDecoder call of Sources/WhisperKit/Core/TextDecoder.swift:381-418.
//
// Why: the lane-per-K GEMVs of decoder.hip do B x K x N FMAs on the vector ALUs - at 32 sequences per step they take 18-29 us
// per launch for 3-13 MB of weights (0.44 TB/s, profiles/r01g_bench_largev3_b32_kernels.json).  With the batch as the 32-wide
// N side of v_mfma_f32_32x32x16_f16 a 1 KB weight tile costs two MFMAs whatever the batch, and the kernel is a weight stream.
//
// Data layout (everything is laid out so that ONE coalesced 1 KB wave load is ONE MFMA fragment, no LDS staging, no shuffles):
//   weights      Wt[row tile rt][k tile kt][lane 64][8 halves]   lane l = weight row rt*32 + (l & 31), k = kt*16 + 8*(l >> 5) + 0..7
//                (re-tiled from the blob's [N][K] once, at model load: d32_tile_weights_kernel)
//   activations  Z[batch tile][k tile][k half][slot 32][8 halves], an f16 hi plane and an f16 lo plane with z ~ hi + lo / 2048
//                (22 mantissa bits: f32-level accuracy, tests/estimate_f16_activation_error.py), written by the PRODUCING kernel
//   accumulators D[weight row (r & 3) + 8 (r >> 2) + 4 (l >> 5)][slot l & 31]: a lane owns ONE slot, so every per-slot scalar
//                (LayerNorm statistics, cache position, liveness) is a per-lane scalar
//
// LayerNorm never runs as a pass: W (gamma (x - mu) rstd + beta) + b = rstd (W (gamma x) - mu (W gamma)) + (W beta + b), so the
// producer of x writes z = gamma_next * x (it knows which LayerNorm comes next), the consumer multiplies raw planes and applies
// (mu, rstd) in its epilogue with g = W gamma and c = W beta + b precomputed in f64 at model load.  The statistics themselves
// are per-row-tile partial (mean, M2) pairs emitted by the producer's finishing workgroups and Chan-combined in a fixed order by
// every consumer - bit-deterministic, batch-invariant, never a two-pass read of the row.
//
// Work split: a workgroup = 4 waves = one 32-row tile x one K slice; the waves split the slice (tw k tiles each, loads of the
// next chunk of TC tiles in flight under the MFMAs of the current one) and meet in LDS.  N = d projections (out projections,
// cross query, fc2) split K across `ks` workgroups as well: each publishes its 32 x 32 partial tile with write-through stores,
// takes a ticket, and the last arriver sums the slices in index order (MI355X_MICROARCH.md "splitk-seam": cheaper here than a
// kernel boundary because the combine is 4-32 KB and the finisher also owns the epilogue).
#include "dec_shared.h"

namespace wh {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------- model-load helpers
__global__ void d32_tile_weights_kernel(const f16* __restrict__ W, int N, int K, u32x4* __restrict__ out, size_t n_out) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int KT = K >> 4;
    const int lane = (int)(o & 63);
    const size_t tile = o >> 6;
    const int kt = (int)(tile % KT), rt = (int)(tile / KT);
    const int row = rt * 32 + (lane & 31), k = kt * 16 + 8 * (lane >> 5);
    u32x4 v = {0, 0, 0, 0};
    if (row < N) v = *reinterpret_cast<const u32x4*>(W + (size_t)row * K + k);
    out[o] = v;
}
void dec32_tile_weights(const f16* W, int N, int K, f16* out, hipStream_t st) {
    const size_t n_out = (size_t)((N + 31) / 32) * (K / 16) * 64;
    d32_tile_weights_kernel<<<(unsigned)((n_out + 255) / 256), 256, 0, st>>>(W, N, K, reinterpret_cast<u32x4*>(out), n_out);
}

__global__ __launch_bounds__(256) void d32_fold_kernel(const f16* __restrict__ W, int N, int K, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ bias,
                                                       float* __restrict__ g, float* __restrict__ c) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    double sg = 0.0, sc = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double w = (double)(float)W[(size_t)row * K + k];
        sg += w * (double)gamma[k];
        sc += w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o, 64); sc += __shfl_xor(sc, o, 64); }
    if (lane == 0) { g[row] = (float)sg; c[row] = (float)(sc + (bias ? (double)bias[row] : 0.0)); }
}
void dec32_fold_vectors(const f16* W, int N, int K, const float* gamma, const float* beta, const float* bias, float* g, float* c, hipStream_t st) {
    d32_fold_kernel<<<(N + 3) / 4, 256, 0, st>>>(W, N, K, gamma, beta, bias, g, c);
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
// (Round 5, measured and rejected, profiles/r05c_projection_two_batch_tiles_per_workgroup_ab_rejected.jsonl: ONE workgroup per weight slab for TWO
// batch tiles - two accumulator sets, the planes of both tiles in every chunk, chunks of 2 k-tiles to stay under the register cap; built,
// bit-identical (tests/test_gpu_round5.py at 70 slots), and slower: every N = d launch 10.5 -> 13.9 us alone (the wave's MFMA chain doubles
// and ten chunk round trips replace four), 5.44 -> 5.98 ms per 64-slot step, 2314 -> 2150 audio-s/s at 64 slots x 3 in flight, 2602 -> 2579 at
// 128 x 3: beside other sessions the launches are not priced in workgroups-in-flight either.  Code in git history, commit "decoder
// projections: two batch tiles per workgroup".)
// NTW: non-temporal weight loads.  A slab is read by n_bt workgroups of ONE XCD (ids x + 8 t): with a single batch tile it is streamed
// once (nt keeps it from displacing the activations in L2); with two or more, the later readers are meant to hit the first one's lines.
// RT (round 6): weight-row tiles per workgroup.  With RT = 2 a wave multiplies every plane fragment it loads by the fragments of TWO consecutive row tiles
// (rt0, rt0 + 1): half as many workgroups per launch, each reading its activation planes once for 64 output channels instead of 32 (1.0 instead of 1.5 KB
// through the CU's L1 per matrix instruction pair).  Why: with 5 - 8 batch tiles a launch is 320 - 1280 workgroups of a few microseconds of latency each, more than
// the CUs that the cross-attention streams of the other sessions leave can hold at once; on half of the CUs the projection chain takes twice as long
// (profiles/r06s_cu_partition_sweep.jsonl), i.e. its cost in flight is workgroup ROUNDS, and two row tiles per workgroup halve them.  The k-tiles of a row tile
// are multiplied by the same wave in the same order as with RT = 1 and meet in LDS in the same order: the same bits (tests/test_gpu_round6.py).
template <int MODE, bool HILO, int TC, bool NTW, int RT>
__global__ __launch_bounds__(256, 2) void dec32_proj_kernel(const P32Args a) {
    constexpr bool kLN = MODE == P32_QKV || MODE == P32_Q || MODE == P32_FC1 || MODE == P32_LOGITS;
    constexpr int kXs = MODE == P32_LOGITS ? 256 * 6 : 32 * 33;
    __shared__ float red[RT][4][16][64];             // the four waves' partial tiles
    __shared__ float st_l[8][32][3];                 // LayerNorm statistics: 8 partial (n, mean, M2) per slot
    __shared__ float xs_raw[kXs];                    // RESID: the tile's new residual values; LOGITS: sampler records (one buffer: the row tiles of a workgroup use it in turn)
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_rt = (a.N + 31) >> 5;
    const int n_rtp = (n_rt + RT - 1) / RT;           // row-tile groups (pairs with RT = 2; the last one may hold a single tile)
    // Workgroup id -> (row tile, K slice, batch tile): ids x + 8 t of one group of 8 share the weight slab x and differ in the batch
    // tile t, so the readers of a slab are dispatched back to back onto the SAME XCD (id % 8) and the slab crosses HBM once, whatever
    // the parity of the tile count (PMC at 64 slots before this: the 1621 logits tiles fetched 273 MB for 133 MB of weights).
    const int grp8 = blockIdx.x / (8 * a.n_bt), in8 = blockIdx.x % (8 * a.n_bt);
    const int xw = grp8 * 8 + (in8 & 7), bt = in8 >> 3;
    if (xw >= n_rtp * a.ks) return;         // padding of the last group (workgroup-uniform)
    const int rt0 = (xw % n_rtp) * RT, ksi = xw / n_rtp;
    const int n_mine = min(RT, n_rt - rt0);  // row tiles of this workgroup that exist (workgroup-uniform)
    const int KT = a.K >> 4;
    const int kt0 = (ksi * 4 + wave) * a.tw;
    const u32x4* wp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) wp[t] = reinterpret_cast<const u32x4*>(a.Wt) + ((size_t)min(rt0 + t, n_rt - 1) * KT + kt0) * 64 + lane;     // (a missing second tile re-reads the first: never stored)
    const size_t zoff = ((size_t)bt * KT + kt0) * 64 + lane;
    const u32x4* hp = reinterpret_cast<const u32x4*>(a.zhi) + zoff;
    const u32x4* lp = HILO ? reinterpret_cast<const u32x4*>(a.zlo) + zoff : nullptr;
    // epilogue coordinates of this thread: slot j, channels n .. n + 3 of every row tile (the accumulator rows 4 wave + i of half-wave h)
    const int j = tid & 31, sub = tid >> 5;
    const int gb = bt * 32 + j;
    const bool valid = gb < a.batch;
    const int xw_dbg = xw;

#define D32_STAMP(i) do { if (a.dbg && tid == 0 && bt == 0) a.dbg[(size_t)(xw_dbg & 4095) * 8 + (i)] = wall_clock64(); } while (0)
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
    // (four row tiles: the 32 registers of these operands are needed by the third stage of the weight ring; they are requested behind the last chunk instead)
    constexpr bool kLateOperands = RT >= 4;
    constexpr int NB = RT >= 4 ? 3 : 2;             // stages of the weight / plane ring: NB - 1 chunks in flight under the MFMAs of the current one
    float4 e0[RT], e1[RT];                          // LN modes: g, c;  RESID: bias, old x
    int pos_l = 0, live_l = 0;
    auto load_operands = [&]() {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int n = min(rt0 + t, n_rt - 1) * 32 + 4 * sub;
            if constexpr (kLN) {
                e0[t] = *reinterpret_cast<const float4*>(a.fold_g + n);
                e1[t] = *reinterpret_cast<const float4*>(a.fold_c + n);
            } else {
                e0[t] = *reinterpret_cast<const float4*>(a.bias + n);
                e1[t] = *reinterpret_cast<const float4*>(a.x + (size_t)gb * a.d + n);
            }
        }
    };
    if constexpr (!kLateOperands) load_operands();
    if (valid) { live_l = slot_live(a.seq + gb); if constexpr (MODE == P32_QKV) pos_l = a.seq[gb].token_index; }
    int rules[6] = {0, 0, 0, 0, 0, 0};
    unsigned masked4[RT];
    int tb = 0, ws_tok = 0, eot_tok = 0, nots_tok = 0, r16 = 0;
#pragma unroll
    for (int t = 0; t < RT; ++t) masked4[t] = 0xffffffffu;
    if constexpr (MODE == P32_LOGITS) {
        if (a.cfg) r16 = a.cfg->f16_logits;
        if (a.stats) {
            if (valid) {
#pragma unroll
                for (int i = 0; i < 6; ++i) rules[i] = a.seq[gb].f_rules[i];
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int n = min(rt0 + t, n_rt - 1) * 32 + 4 * sub;
                if (n + 3 < a.N) masked4[t] = *reinterpret_cast<const unsigned*>(a.sup_mask + n);
                else {
                    masked4[t] = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) masked4[t] |= (unsigned)(n + i < a.N ? a.sup_mask[n + i] : 1) << (8 * i);
                }
            }
            tb = a.cfg->time_token_begin; ws_tok = a.cfg->whitespace_token; eot_tok = a.cfg->end_token; nots_tok = a.cfg->no_timestamps_token;
        }
    }

    // ---- weight stream x activation planes on the matrix cores
    f32x16 acc_h[RT], acc_l[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[t][r] = 0.0f; acc_l[t][r] = 0.0f; }
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
        u32x4 w_[NB][RT][TC], h_[NB][TC], l_[NB][HILO ? TC : 1];
        auto ld = [&](u32x4 (&w)[RT][TC], u32x4 (&h)[TC], u32x4 (&l)[HILO ? TC : 1], int c) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < TC; ++i) w[t][i] = NTW ? __builtin_nontemporal_load(wp[t] + (size_t)(c * TC + i) * 64) : wp[t][(size_t)(c * TC + i) * 64];
#pragma unroll
            for (int i = 0; i < TC; ++i) h[i] = hp[(size_t)(c * TC + i) * 64];
            if constexpr (HILO) {
#pragma unroll
                for (int i = 0; i < TC; ++i) l[i] = lp[(size_t)(c * TC + i) * 64];
            }
        };
        auto mm = [&](const u32x4 (&w)[RT][TC], const u32x4 (&h)[TC], const u32x4 (&l)[HILO ? TC : 1]) {
#pragma unroll
            for (int i = 0; i < TC; ++i) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const f16x8 wf = __builtin_bit_cast(f16x8, w[t][i]);
                    acc_h[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(f16x8, h[i]), acc_h[t], 0, 0, 0);
                    if constexpr (HILO) acc_l[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(f16x8, l[i]), acc_l[t], 0, 0, 0);
                }
            }
        };
        // ring of NB stages, the loop unrolled by NB so that every stage index is a constant: chunk c lives in stage c % NB; while chunk c is multiplied the
        // chunks c + 1 .. c + NB - 1 are in flight (NB = 2: the double buffer of rounds 2 - 5, instruction for instruction)
        const int nch = a.tw / TC;
#pragma unroll
        for (int u = 0; u + 1 < NB; ++u)
            if (u < nch) ld(w_[u], h_[u], l_[u], u);
        D32_STAMP(1);
#pragma unroll 1
        for (int c = 0; c < nch; c += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                constexpr int kAhead = NB - 1;
                if (c + u + kAhead < nch) ld(w_[(u + kAhead) % NB], h_[(u + kAhead) % NB], l_[(u + kAhead) % NB], c + u + kAhead);
                __builtin_amdgcn_sched_barrier(0);      // loads of the next chunks stay ahead of this chunk's MFMAs
                if (c + u < nch) mm(w_[u], h_[u], l_[u]);
                if (u == 0 && c == 0) D32_STAMP(2);
            }
        }
    }
    if constexpr (kLateOperands) load_operands();
    stats_to_lds();       // after the stream: the statistics are epilogue operands (timeline probe: waiting for them up front cost 1 us per launch)
    D32_STAMP(3);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[t][wave][r][lane] = HILO ? fmaf(acc_l[t][r], 1.0f / 2048.0f, acc_h[t][r]) : acc_h[t][r];
    __syncthreads();
    D32_STAMP(4);
    float v[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[t][i] = ((red[t][0][4 * wave + i][lane] + red[t][1][4 * wave + i][lane]) + red[t][2][4 * wave + i][lane]) + red[t][3][4 * wave + i][lane];

    // ---- K split across workgroups: publish, ticket, the last arriver sums the slices in index order.  Write-through (sc1)
    // 16-byte stores, a drained vmcnt in every storing wave, one relaxed ticket; the finisher reads the slabs with sc1 loads, which
    // bypass its L1 and are served by L2 - no agent-scope fence on either side (MI355X_MICROARCH.md "handoff-flag", R1).
    if (a.ks > 1) {
        // (the slices of row tile rt0 + t live where a workgroup of its own would put them; ONE ticket per group of row tiles: its tiles always travel together)
        float* base = a.part + (((size_t)bt * n_rt + rt0) * a.ks) * 1024 + tid * 4;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (t < n_mine) {
                const f32x4 pv4 = {v[t][0], v[t][1], v[t][2], v[t][3]};
                float* mine = base + ((size_t)t * a.ks + ksi) * 1024;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine), "v"(pv4) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int* cnt = a.ticket + bt * n_rt + rt0;
            const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == a.ks - 1);
            if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next launch
            last_flag = last;
        }
        __syncthreads();
        if (!last_flag) {                   // workgroup-uniform
            if constexpr (MODE == P32_Q) { if (a.gate && blockIdx.x == 0 && tid == 0) xattn_gate_acquire(a.gate); }
            return;
        }
        float pv[RT][8][4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int s = 0; s < 8; ++s) {       // every load is issued before the first add; slices past ks re-read slice 0 and are dropped
                const float* p = base + ((size_t)(t < n_mine ? t : 0) * a.ks + (s < a.ks ? s : 0)) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) pv[t][s][i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x_ = pv[t][0][i];
#pragma unroll
                for (int s = 1; s < 8; ++s) x_ += (s < a.ks) ? pv[t][s][i] : 0.0f;
                v[t][i] = x_;
            }
    }

    D32_STAMP(5);
    // ---- epilogues (per row tile of this workgroup; every condition below is workgroup-uniform or guards stores only)
    float mu = 0.0f, rstd = 0.0f;
    if constexpr (kLN) {
        float cn = 0.0f, cm = 0.0f, cM2 = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) chan_merge(cn, cm, cM2, st_l[s][j][0], st_l[s][j][1], st_l[s][j][2]);
        mu = cm; rstd = rsqrtf(cM2 / (float)a.d + 1e-5f);
    }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    if (t >= n_mine) break;
    if constexpr (MODE == P32_RESID || MODE == P32_LOGITS) { if (t > 0) __syncthreads(); }      // the previous tile's readers are done with xs_raw
    const int rt = rt0 + t, n = rt * 32 + 4 * sub;
    float y[4];
    if constexpr (kLN) {
        const float g4[4] = {e0[t].x, e0[t].y, e0[t].z, e0[t].w}, c4[4] = {e1[t].x, e1[t].y, e1[t].z, e1[t].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = fmaf(rstd, v[t][i] - mu * g4[i], c4[i]);
    }
    if constexpr (MODE == P32_QKV) {
        if (valid && live_l) {
            const int d = a.d;
            if (n < d) *reinterpret_cast<float4*>(a.q + (size_t)gb * d + n) = float4{y[0], y[1], y[2], y[3]};
            else {
                int c = n - d;
                f16* dst = a.self_k;
                if (c >= d) { c -= d; dst = a.self_v; }
                const int pos = min(max(pos_l, 0), kMaxTok - 1);
                *reinterpret_cast<f16x4*>(dst + (((size_t)gb * a.n_head + (c >> 6)) * kMaxTok + pos) * kHeadDim + (c & 63)) =
                    f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
            }
        }
    } else if constexpr (MODE == P32_Q) {
        if (valid && live_l) *reinterpret_cast<float4*>(a.q + (size_t)gb * a.d + n) = float4{y[0], y[1], y[2], y[3]};
    } else if constexpr (MODE == P32_FC1) {
        if (valid) {      // hidden activations as an f16 hi | lo pair: rounding them to ONE f16 plane (round 2) was the largest single term
            f16x4 hi, lo;  // of the decoder's logits error at 32 layers (1e-3 of the fp32 oracle; tests/test_gpu_fulldepth.py)
#pragma unroll
            for (int i = 0; i < 4; ++i) { f16 h_, l_; split_hilo(gelu_erf(y[i]), h_, l_); hi[i] = h_; lo[i] = l_; }
            const size_t o = plane_index(gb, n, a.N);
            *reinterpret_cast<f16x4*>(a.h_out + o) = hi;
            *reinterpret_cast<f16x4*>(a.h_out_lo + o) = lo;
        }
    } else if constexpr (MODE == P32_RESID) {
        const float b4[4] = {e0[t].x, e0[t].y, e0[t].z, e0[t].w}, x4[4] = {e1[t].x, e1[t].y, e1[t].z, e1[t].w};
        float xn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xn[i] = x4[i] + (v[t][i] + b4[i]);
        d32_resid_tail(xn, valid && live_l, bt, rt, n_rt, n, j, gb, tid, a.d, a.x, a.gamma_next, a.zhi_out, a.zlo_out, a.stat_out,
                       reinterpret_cast<float (*)[33]>(xs_raw));
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
            SoftStat t_{-INFINITY, 0.0f, 0x7fffffff}, u_{-INFINITY, 0.0f, 0x7fffffff};
            const int blank = rules[0], ts_active = rules[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = n + i;
                bool masked = ((masked4[t] >> (8 * i)) & 0xff) != 0 || id >= a.N;                                   // SuppressTokensFilter
                masked |= blank && (id == ws_tok || id == eot_tok);                                               // SuppressBlankFilter
                masked |= ts_active && (id == nots_tok || (id >= rules[2] && id < rules[3]) || (id >= rules[4] && id < rules[5]));   // TimestampRulesFilter
                if (!masked) { if (id < tb) stat_merge(t_, y[i], 1.0f, id); else stat_merge(u_, y[i], 1.0f, id); }
            }
            float* xr = xs_raw;
            float* rec = xr + (size_t)(sub * 32 + j) * 6;
            rec[0] = t_.m; rec[1] = t_.s; rec[2] = __int_as_float(t_.i); rec[3] = u_.m; rec[4] = u_.s; rec[5] = __int_as_float(u_.i);
            __syncthreads();
            if (tid < 32 && valid && live_l) {
                SoftStat T{-INFINITY, 0.0f, 0x7fffffff}, U{-INFINITY, 0.0f, 0x7fffffff};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float* r_ = xr + (size_t)(s * 32 + tid) * 6;
                    stat_merge(T, r_[0], r_[1], __float_as_int(r_[2]));
                    stat_merge(U, r_[3], r_[4], __float_as_int(r_[5]));
                }
                float* o = a.stats + ((size_t)gb * kStatBlocks + rt) * 8;
                *reinterpret_cast<float4*>(o) = float4{T.m, T.s, __int_as_float(T.i), U.m};
                *reinterpret_cast<float2*>(o + 4) = float2{U.s, __int_as_float(U.i)};
            }
        }
    }
  }     // row tiles of this workgroup
    D32_STAMP(6);
    if constexpr (MODE == P32_Q) { if (a.gate && blockIdx.x == 0 && tid == 0) xattn_gate_acquire(a.gate); }    // dec_shared.h
}

// (Round 5, measured and rejected, profiles/r05l_*: the cross-query projection and xabs_qk as ONE launch - a workgroup per (head, batch tile): phase 1
// = this kernel's arithmetic for the head's two row tiles of W_cq with q_h kept in LDS, phase 2 = xabs_qk's arithmetic over the head's d / 32
// channel tiles.  Bit-identical (tests at 70 / 33 slots), and no faster: 23.3 us against 10.4 + 10.3 us - 80 workgroups that each stream 328 KB in two
// dependent phases instead of 160 + 400 short ones - 8.23 -> 8.40 ms per 128-slot step alone, 2610 -> 2615 audio-s/s in flight.  Code in git
// history, commit "cross query + absorbed query as one launch".)

// ---------------------------------------------------------------------------------------------- embedding
// x = token_embedding[next_token] + positional_embedding[token_index] (openai/whisper TextDecoder.forward), the head of the
// residual chain: same tail as a RESID finisher (x, gamma_1 x planes of layer 0, row-tile statistics).
__global__ __launch_bounds__(256) void dec32_embed_kernel(const f16* __restrict__ emb, const float* __restrict__ pos, const SeqState* __restrict__ seq,
                                                          int batch, int d, int n_vocab, float* x, const float* gamma_next, f16* zhi, f16* zlo,
                                                          float2* stat_out) {
    __shared__ float xs[32][33];
    const int tid = threadIdx.x, j = tid & 31, sub = tid >> 5;
    const int rt = blockIdx.x, bt = blockIdx.y, n_rt = gridDim.x;
    const int n = rt * 32 + 4 * sub, gb = bt * 32 + j;
    const bool valid = gb < batch;
    float xn[4] = {0, 0, 0, 0};
    bool live = false;
    if (valid) {
        live = slot_live(seq + gb);
        const int tok = min(max(seq[gb].next_token, 0), n_vocab - 1);
        const int p = min(max(seq[gb].token_index, 0), kMaxTok - 1);
        const f16x4 e = *reinterpret_cast<const f16x4*>(emb + (size_t)tok * d + n);
        const float4 pz = *reinterpret_cast<const float4*>(pos + (size_t)p * d + n);
        xn[0] = (float)e[0] + pz.x; xn[1] = (float)e[1] + pz.y; xn[2] = (float)e[2] + pz.z; xn[3] = (float)e[3] + pz.w;
    }
    d32_resid_tail(xn, valid && live, bt, rt, n_rt, n, j, gb, tid, d, x, gamma_next, zhi, zlo, stat_out, xs);
}
void launch_dec32_embed(const f16* emb, const float* pos, const SeqState* seq, int batch, int d, int n_vocab, int n_bt, float* x,
                        const float* gamma_next, f16* zhi, f16* zlo, float2* stat, hipStream_t st) {
    ProfScope ps_(KK_DEC_EMBED, st);
    dec32_embed_kernel<<<dim3(d / 32, n_bt), 256, 0, st>>>(emb, pos, seq, batch, d, n_vocab, x, gamma_next, zhi, zlo, stat);
}

// ---------------------------------------------------------------------------------------------- launcher
static int env_int32(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// K splits across workgroups: only the N = d projections need them (40 row tiles at d = 1280 would leave 5/6 of the chip idle);
// the split must divide the K / 64 tile groups.  WH_D32_KS_* override (tuning).
int dec32_ksplit(int mode, int N, int K, bool f16_input) {
    static const int ks_resid = env_int32("WH_D32_KS_RESID", 0), ks_fc2 = env_int32("WH_D32_KS_FC2", 0), ks_q = env_int32("WH_D32_KS_Q", 0),
                     ks_wide = env_int32("WH_D32_KS_WIDE", 1), tile_kb = env_int32("WH_D32_TILE_KB", 96);
    int want = (mode == P32_RESID) ? (f16_input ? ks_fc2 : ks_resid) : (mode == P32_Q ? ks_q : ks_wide);
    // default for the N = d projections: as many K slices as keep a workgroup's weight slab at <= tile_kb KB (a 32-row tile of K
    // columns is K / 16 KB): d = 384 -> 1 (no ticket at all), d = 1280 -> 1 for K = d, 4 for fc2's K = 4d
    if (want <= 0) want = (K / 16 + tile_kb - 1) / tile_kb;
    const int groups = K / 64;
    want = max(1, min(want, 8));
    while (want > 1 && groups % want) --want;
    if ((long long)((N + 31) / 32) * want * 1024 > kD32PartFloats) want = 1;
    return want;
}

template <int MODE, bool HILO, bool NTW>
static void launch_tc_w(const P32Args& a, dim3 grid, hipStream_t st) {
    const int tw = a.tw;
    // chunk size of the weight stream: 5 k-tiles (196 - 204 registers: two workgroups per CU) up to four batch tiles; beyond (160- to 256-slot device batches: 200 - 1280 workgroups
    // per launch, which must find room while two cross-attention streams hold most CUs) chunks of 2 (114 - 132 registers: three to four workgroups per CU): on the driver's line
    // 2661 -> 2681 (from 8 tiles) -> 2689 - 2701 audio-s/s (from 6 / 5 tiles; profiles/r06r_projection_chunk_threshold.jsonl, r06q_*; no difference at 128 slots, r05n).  The
    // chunking does not touch the order of the matrix instructions: same bits.  WH_D32_TC (chunk cap) and WH_D32_TC_BT (first batch-tile count with small chunks) override.
    static const int tc_env = env_int32("WH_D32_TC", 0);
    static const int tc_bt = env_int32("WH_D32_TC_BT", 5);
    const int tc_cap = tc_env > 0 ? tc_env : (a.n_bt >= tc_bt ? 2 : 5);
    if constexpr (MODE != P32_LOGITS && MODE != P32_Q) {
        if (a.rt == 4) {    // four row tiles per workgroup: chunks of 1 k-tile (230 - 244 registers: two workgroups per CU)
            dec32_proj_kernel<MODE, HILO, 1, NTW, 4><<<grid, 256, 0, st>>>(a);
            return;
        }
    }
    if (a.rt >= 2) {        // two row tiles per workgroup (launch_dec32_proj decides): chunks of 4 k-tiles (236 - 256 registers, two workgroups per CU like chunks of 2: 2823 -> 2834 audio-s/s), else 2 or 1; WH_D32_RT2_TC: A/B
        static const int tc2 = env_int32("WH_D32_RT2_TC", 4);
        if (tw % 4 == 0 && tc2 >= 4) dec32_proj_kernel<MODE, HILO, 4, NTW, 2><<<grid, 256, 0, st>>>(a);
        else if (tw % 2 == 0 && tc2 >= 2) dec32_proj_kernel<MODE, HILO, 2, NTW, 2><<<grid, 256, 0, st>>>(a);
        else dec32_proj_kernel<MODE, HILO, 1, NTW, 2><<<grid, 256, 0, st>>>(a);
        return;
    }
    // chunks of <= 5 k-tiles: 6 would put the LOGITS instantiation at 226 VGPRs + accumulators = one wave per SIMD (tiny.en: 22 -> 47 us)
    if (tw % 5 == 0 && tc_cap >= 5) dec32_proj_kernel<MODE, HILO, 5, NTW, 1><<<grid, 256, 0, st>>>(a);
    else if (tw % 4 == 0 && tc_cap >= 4) dec32_proj_kernel<MODE, HILO, 4, NTW, 1><<<grid, 256, 0, st>>>(a);
    else if (tw % 3 == 0 && tc_cap >= 3) dec32_proj_kernel<MODE, HILO, 3, NTW, 1><<<grid, 256, 0, st>>>(a);
    else if (tw % 2 == 0) dec32_proj_kernel<MODE, HILO, 2, NTW, 1><<<grid, 256, 0, st>>>(a);
    else dec32_proj_kernel<MODE, HILO, 1, NTW, 1><<<grid, 256, 0, st>>>(a);
}
template <int MODE, bool HILO>
static void launch_tc(const P32Args& a, dim3 grid, hipStream_t st) {
    static const int ntw = env_int32("WH_D32_NTW", -1);      // -1: nt for a single batch tile only; 0 never; 1 always (the behaviour before round 4's last change)
    const bool nt = ntw < 0 ? a.n_bt == 1 : ntw != 0;
    if (nt) launch_tc_w<MODE, HILO, true>(a, grid, st); else launch_tc_w<MODE, HILO, false>(a, grid, st);
}

unsigned long long* debug_buffer();
void launch_dec32_proj(int mode, const P32Args& a_in, int n_bt, hipStream_t st) {
    P32Args a = a_in;
    a.dbg = (debug_buffer() && a.prof_kind >= 0) ? debug_buffer() + (size_t)a.prof_kind * 4096 * 8 : nullptr;   // WH_DBG=1 timeline probe
    const bool hilo = a.zlo != nullptr;
    a.ks = dec32_ksplit(mode, a.N, a.K, a.K > a.N);      // K > N: the fc2 shape (its own split knob)
    a.tw = a.K / (64 * a.ks);
    a.n_bt = n_bt;
    // Row tiles per workgroup (round 6).  From four to five batch tiles on (128- to 256-slot device batches) a launch is 320 - 1280 workgroups and its time follows the CUs it gets
    // (alone at 256 slots, whole chip / 128 / 64 CUs: qkv 21 / 33 / 56 us, fc1 23 / 36 / 64, fc2 28 / 44 / 80, profiles/r06t_chain_on_cus_ab.jsonl): the wide projections are bound by
    // the bytes their workgroups pull through the CUs' L1s - 3 KB per pair of matrix instructions (1 KB weight tile + 2 KB hi | lo planes) - and beside two cross-attention
    // streams they have half of the chip or less.  Two row tiles per workgroup share the planes (2 KB per pair), four (qkv, fc1, fc2) 1.5 KB: headline 2738 -> 2825 (two) -> 2837
    // audio-s/s (four), profiles/r06u .. r06w_*.  Same bits (a row tile's k-tiles meet the same wave in the same order).  WH_D32_RT_BT / WH_D32_RT4_BT (first batch-tile count
    // with 2 / 4 row tiles; 99 = never) and WH_D32_RT4_MODES (bit 0 qkv, 1 fc1, 2 fc2) are the A/B knobs.  Thresholds (profiles/r06ad_*): 128-slot batches x 3 in flight 2666 -> 2744
    // with two row tiles (four: 2737), 64-slot batches 2350 -> 2345 with two, 2229 with four: two from four batch tiles on, four from five.
    static const int rt_bt = env_int32("WH_D32_RT_BT", 4), rt4_bt = env_int32("WH_D32_RT4_BT", 5), rt4_modes = env_int32("WH_D32_RT4_MODES", 7);
    a.rt = n_bt >= rt_bt ? 2 : 1;
    {
        const int bit = mode == P32_QKV ? 1 : mode == P32_FC1 ? 2 : (mode == P32_RESID && a.K > a.N) ? 4 : 0;
        if (n_bt >= rt4_bt && (rt4_modes & bit) && ((a.N + 31) / 32) % 4 == 0) a.rt = 4;
    }
    const int nx = (((a.N + 31) / 32 + a.rt - 1) / a.rt) * a.ks;
    const dim3 grid((unsigned)(((nx + 7) / 8) * 8 * n_bt));
    ProfScope ps_(a.prof_kind, st);
    switch (mode) {
        case P32_QKV: launch_tc<P32_QKV, true>(a, grid, st); break;
        case P32_Q: launch_tc<P32_Q, true>(a, grid, st); break;
        case P32_FC1: launch_tc<P32_FC1, true>(a, grid, st); break;
        case P32_LOGITS: launch_tc<P32_LOGITS, true>(a, grid, st); break;
        default:
            if (hilo) launch_tc<P32_RESID, true>(a, grid, st);
            else launch_tc<P32_RESID, false>(a, grid, st);
    }
}

}  // namespace wh
