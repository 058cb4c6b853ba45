// Decoder token step for gfx950: replaces the per-token CoreML TextDecoder call of
// Sources/WhisperKit/Core/TextDecoder.swift:381-418 plus the host-side K4..K8 work around it
// (updateKVCache :218-270, updateAlignmentWeights :272-296, LogitsFilter.swift, TokenSampler.swift)
// and the loop bookkeeping of decodeText (:573-757).  Everything the loop needs lives in device
// memory (SeqState), so a step is a fixed kernel chain with no host round trip: the next input token,
// the cache position and the stop flag are read from / written to SeqState by the kernels themselves.
//
// Every grid-wide dependency of a decoder layer is one kernel boundary (cheaper on gfx950 than an in-kernel grid
// barrier, MI355X_MICROARCH.md "boundary" vs "barrier-xcd").  The projections run on the matrix cores with the
// batch as the 32-wide N side of the MFMA tile (decoder32.hip); this file holds the attention kernels, the
// samplers and the step launcher:
//
//   embed         x = token_embedding[next_token] + positional_embedding[token_index]
//   proj<QKV>     LN1 (folded) -> q (f32); k, v straight into the head-major self-attention cache at `pos`
//   self_attn     one workgroup per (head, slot): softmax(q K^T) V over <= 224 cached positions -> att planes
//   proj<RESID>   x += W_o att + b_o
//   proj<Q>       LN2 (folded) -> cross-attention query
//   cross_attn    workgroups per (key split, head, slot) over the 1500 cached cross K/V rows (flash-decoding
//                 split; the last-arriving split combines the partials in a fixed order -> att planes); alignment
//                 heads also store their raw score row (DecodingCache.alignmentWeights row tokenIndex + 1)
//   proj<RESID>   x += W_co att + b_co
//   proj<FC1>     LN3 (folded) -> GELU(fc1) (f16 plane)
//   proj<RESID>   x += W_2 h + b_2
//   ... per layer, then
//   proj<LOGITS>  LN_f (folded) . E^T (tied embedding) + the index-predicate logits filters and per-tile softmax statistics
//   sampler       merge of the statistics, greedy / top-k sample, decodeText state advance
//
// All results are bit-deterministic (fixed summation orders, no float atomics) and batch-invariant: a slot decodes to the
// same bits alone or among 31 others (tests/test_gpu_dims.py).
#include <algorithm>
#include <cstdlib>

#include "dec_shared.h"

namespace wh {

// ---------------------------------------------------------------------------------------------- fused greedy sampler, part 1
// Filter rules of one sampling step as scalars (restating LogitsFilter.swift): r[0] SuppressBlank active,
// r[1] TimestampRules active, [r2, r3) and [r4, r5) id ranges masked by the timestamp rules.
// last_ts_hint: -2 = scan the history for the last timestamp token (TimestampRulesFilter walks it backwards); >= -1 = the caller already knows the
// index of the last token >= time_token_begin in [0, n_tok) (-1: none) - sampler_final_kernel finds it with all of its threads, because with
// text-only histories (random-init weights: the benchmark) the one-thread backward walk over <= 223 LDS words was 9 of the kernel's 12 us
__device__ __forceinline__ void compute_filter_rules(const SamplerCfg& cfg, const SeqState* sq, int n_tok, int V, int* r, int last_ts_hint = -2) {
    const int tb = cfg.time_token_begin;
    int blank = 0, ts_active = 0, r1lo = 0, r1hi = 0, r2lo = 0, r2hi = 0;
    blank = cfg.suppress_blank && (n_tok == cfg.prefilled_index);            // SuppressBlankFilter :44-50
    if (cfg.timestamp_rules) {                                               // TimestampRulesFilter :72-129
        int sb = -1;
        if (cfg.is_multilingual) {                                           // :131-142
            for (int i = 0; i < 3 && i < n_tok; ++i)
                if (sq->tokens[i] == cfg.transcribe_token || sq->tokens[i] == cfg.translate_token) { sb = max(i + 1, cfg.initial_prompt_index); break; }
        } else sb = cfg.initial_prompt_index;
        if (sb >= 0 && sb <= n_tok) {
            ts_active = 1;
            if (n_tok > sb) {
                int cnt = n_tok - sb;
                bool lastTs = sq->tokens[n_tok - 1] >= tb;
                bool penTs = cnt < 2 || sq->tokens[n_tok - 2] >= tb;
                if (lastTs) {
                    if (penTs) { r1lo = tb; r1hi = V; }          // has to be non-timestamp
                    else { r1lo = 0; r1hi = cfg.end_token; }     // cannot be normal text
                }
                int lastTimestamp = -1;
                if (last_ts_hint >= -1) {
                    if (last_ts_hint >= sb) lastTimestamp = sq->tokens[last_ts_hint];
                } else {
                    for (int i = n_tok - 1; i >= sb; --i)
                        if (sq->tokens[i] >= tb) { lastTimestamp = sq->tokens[i]; break; }
                }
                if (lastTimestamp >= 0) {
                    int tl = (lastTs && !penTs) ? lastTimestamp : lastTimestamp + 1;
                    r2lo = tb; r2hi = tl;
                }
            }
        }
    }
    r[0] = blank; r[1] = ts_active; r[2] = r1lo; r[3] = r1hi; r[4] = r2lo; r[5] = r2hi;
}

// ---------------------------------------------------------------------------------------------- attention
struct AttnArgs {
    int batch, d, n_head, layer, n_layer, n_split;
    int cross_div;           // > 1: slot b attends over the cross K / V of slot b / cross_div (beams of one audio share one copy)
    const float* q;          // [B][d]
    const f16* self_k; const f16* self_v;     // layer base [Bmax][H][224][64]
    const f16 *cross_k_hi, *cross_v_hi;       // layer base [Bmax][H][1500][64]: 24-bit rows (round 5: Float16 rows cost 7e-3 sigma under a sharp softmax) =
    const signed char *cross_k_lo, *cross_v_lo;   //   a Float16 and a signed 8-bit residual in units of ulp(hi) / 256 per element (kernels.h, hr24)
    f16 *att_hi, *att_lo;    // attention output (before the out projection) as an f16 hi | lo pair in B-fragment plane order (decoder32.hip)
    float* part;             // [B][H][n_split][kPartStride]: (m, l, o[64]) of every key split, one 128-byte-aligned slot each
    int* ticket;             // [B][H] arrival counters (zero between launches)
    float* align; const int* align_slot; int n_align;   // [B][224][n_align][1500] raw score rows of the alignment heads
    SeqState* seq;
    int no_fence;
    int self_rows;             // self-attention: cache rows fetched (DecodeBuffers.self_rows)
    const int* self_owner;     // self-attention: row -> owning slot table of beam search, or null
    int* gate; int gate_wg;    // cross-attention gate (dec_shared.h): the workgroup with linear id gate_wg gives it back at entry
    unsigned long long* dbg;   // optional timeline probe (WH_DBG=1)
};
__device__ __forceinline__ void gate_release_if_mine(const AttnArgs& a) {
    if (a.gate && threadIdx.x == 0 && (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) == a.gate_wg) xattn_gate_release(a.gate);
}
#define ATT_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 4096 * 8 + (i)] = (unsigned long long)wall_clock64(); } while (0)

__device__ __forceinline__ void store_att(const AttnArgs& a, int b, int n, float v) {
    f16 hi, lo;
    split_hilo(v, hi, lo);
    const size_t o = plane_index(b, n, a.d);
    a.att_hi[o] = hi;
    a.att_lo[o] = lo;
}

// One query against keys [t0, t0 + n) of a head-major K/V block (rows of 64 channels; T = f16: the self-attention cache, T = hr24: the
// cross-attention rows, a Float16 plus an 8-bit residual per element = 19 mantissa bits in 3 bytes).  Thread layout: 8 lanes per key (8
// channels each: 16 bytes of f16, + 8 bytes of residuals), 32 keys per pass, PASSES passes; all K and V rows of the block are in flight
// before the first use.  Returns this block's softmax statistics (m, l) and leaves the unnormalised output
// o[64] = sum_t exp(s_t - m) V[t] in o_out (LDS, valid for tid < 64).  raw_scores (optional, global) gets s_t.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <typename T> struct KvSrc;
template <> struct KvSrc<f16> { const f16* p; };
template <> struct KvSrc<hr24> { const f16* hi; const signed char* lo; };
template <typename T> struct KvPiece;                    // 8 channels of one row
template <> struct KvPiece<f16> { uint4 h; };
template <> struct KvPiece<hr24> { uint4 h; uint2 l; };
template <bool NT>
__device__ __forceinline__ uint4 load16(const void* p) {      // NT: non-temporal (streamed once per step)
    if constexpr (NT) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return uint4{v[0], v[1], v[2], v[3]};
    } else return *reinterpret_cast<const uint4*>(p);
}
template <bool NT>
__device__ __forceinline__ uint2 load8(const void* p) {
    if constexpr (NT) {
        const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
        return uint2{v[0], v[1]};
    } else return *reinterpret_cast<const uint2*>(p);
}
template <bool NT>
__device__ __forceinline__ void load_kv8(const KvSrc<f16>& s, size_t o, KvPiece<f16>& r) { r.h = load16<NT>(s.p + o); }
template <bool NT>
__device__ __forceinline__ void load_kv8(const KvSrc<hr24>& s, size_t o, KvPiece<hr24>& r) { r.h = load16<NT>(s.hi + o); r.l = load8<NT>(s.lo + o); }
__device__ __forceinline__ void kv8_to_float(const KvPiece<f16>& k, float (&f)[8]) {
    const f16x8 h = *reinterpret_cast<const f16x8*>(&k.h);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)h[j];
}
__device__ __forceinline__ void kv8_to_float(const KvPiece<hr24>& k, float (&f)[8]) {
    const f16x8 h = *reinterpret_cast<const f16x8*>(&k.h);
    const unsigned hw[4] = {k.h.x, k.h.y, k.h.z, k.h.w}, lw[2] = {k.l.x, k.l.y};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned hb = (hw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        const int lo = (int)(lw[j >> 2] << (24 - 8 * (j & 3))) >> 24;              // sign-extended byte j
        f[j] = fmaf((float)lo, hr24_unit(hb), (float)h[j]);
    }
}

// attend_fetch: every K and V row of the block is requested before anything is used.  Rows past n_load are not skipped but
// re-read row n_load - 1 (clamped address): straight-line loads, no exec-mask branches; attend_compute ignores them (key >= n).
template <int PASSES, bool NT, typename T>
__device__ __forceinline__ void attend_fetch(const KvSrc<T>& kb, const KvSrc<T>& vb, int n_load, KvPiece<T> (&kreg)[PASSES], KvPiece<T> (&vreg)[PASSES]) {
    const int part = threadIdx.x & 7, kg = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = min(kg + 32 * i, n_load - 1);
        load_kv8<NT>(kb, (size_t)key * kHeadDim + part * 8, kreg[i]);
    }
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = min(kg + 32 * i, n_load - 1);
        load_kv8<NT>(vb, (size_t)key * kHeadDim + part * 8, vreg[i]);
    }
}
__device__ __forceinline__ void attend_load_q(const float* __restrict__ qg, float (&qv)[8]) {
    const int part = threadIdx.x & 7;
    float4 q0 = *reinterpret_cast<const float4*>(qg + part * 8);
    float4 q1 = *reinterpret_cast<const float4*>(qg + part * 8 + 4);
    qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
}
// attend_compute: the fetched rows against ONE query; n of them count (rows past n: score -inf, V zeroed - they may be clamped re-reads
// or, in the self-attention cache, rows no step has written yet).  Shared by the self- and cross-attention kernels.
template <int PASSES, typename T>
__device__ __forceinline__ void attend_compute(const float (&qv)[8], KvPiece<T> (&kreg)[PASSES], KvPiece<T> (&vreg)[PASSES], int n, float* raw_scores,
                                               float* red /* [16] */, float* osum /* [4][64] */, float* o_out /* [64] */, float* m_out, float* l_out,
                                               unsigned long long* stamp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = tid & 7, kg = tid >> 3;
    float s[PASSES];
    float lmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        float k8[8];
        kv8_to_float(kreg[i], k8);
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t = fmaf(k8[j], qv[j], t);
        t = group8_sum(t);
        const int key = kg + 32 * i;
        const bool in = key < n;
        if (raw_scores && part == 0 && in) raw_scores[key] = t;
        t = in ? t : -INFINITY;
        lmax = fmaxf(lmax, t);
        s[i] = t;
    }
    lmax = wave_max_dpp(lmax);
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
        float v8[8];
        kv8_to_float(vreg[i], v8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(p, valid ? v8[j] : 0.0f, o[j]);      // rows past n were fetched speculatively: keep 0 * garbage out
    }
    lsum = wave_sum_dpp(lsum);
    if (lane == 0) red[4 + wave] = lsum;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = stride8_sum(o[j]);
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) osum[wave * 64 + lane * 8 + j] = o[j];
    }
    __syncthreads();
    if (tid < 64) o_out[tid] = (osum[tid] + osum[64 + tid]) + (osum[128 + tid] + osum[192 + tid]);
    if (stamp && threadIdx.x == 0) stamp[3] = wall_clock64();
    *m_out = m;
    *l_out = (red[4] + red[5]) + (red[6] + red[7]);
}

template <int PASSES, bool NT, typename T, typename GetN, typename QFix>
__device__ __forceinline__ bool attend_block(const float* __restrict__ qg, const KvSrc<T>& kb, const KvSrc<T>& vb, int n_load,
                                             GetN get_n, QFix qfix, float* const* raw_pp, float* red /* [16] */, float* osum /* [4][64] */,
                                             float* o_out /* [64] */, float* m_out, float* l_out, unsigned long long* stamp = nullptr) {
    // n_load rows are FETCHED right away; how many of them count (n = get_n(), < 0: slot not live) is only looked at
    // afterwards, so the slot-state loads and the K/V stream share one memory round trip instead of two.
    KvPiece<T> kreg[PASSES], vreg[PASSES];
    attend_fetch<PASSES, NT, T>(kb, vb, n_load, kreg, vreg);
    float qv[8];
    attend_load_q(qg, qv);
    const int n = get_n();
    if (n < 0) return false;            // workgroup-uniform
    qfix(qv, threadIdx.x & 7);          // hook for a query fix-up (identity today)
    attend_compute<PASSES, T>(qv, kreg, vreg, n, *raw_pp, red, osum, o_out, m_out, l_out, stamp);
    return true;
}

// self_rows cached positions are FETCHED (speculatively, before the slot state is known): the launcher passes the smallest bound that
// covers every live slot's position in its launches (the host replays one step graph per 8 positions), so the cache traffic follows
// the decoded length (PMC: 37 MB per launch at 32 slots for 1.5 MB needed with a fixed 224-row fetch, 3.4 x the needed bytes with a
// bound per 32-row band; rows past the bound re-read the last row: cache hits).
template <int PASSES>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const AttnArgs a) {
    __shared__ float red[16], osum[256], o_l[64];
    const int h = blockIdx.x, b = blockIdx.y;
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;     // looked at after the K/V loads are issued
    const int d = a.d;
    const int n_load = min(PASSES * 32, a.self_rows > 0 ? a.self_rows : PASSES * 32);
    const size_t base = ((size_t)b * a.n_head + h) * kMaxTok * kHeadDim;
    float m, l;
    float* raw = nullptr;
    auto get_n = [&]() { return (s_act && !s_done) ? min(min(max(s_ti, 0), kMaxTok - 1) + 1, n_load) : -1; };
    auto qfix = [](float (&)[8], int) {};
    if (!attend_block<PASSES, false, f16>(a.q + (size_t)b * d + h * kHeadDim, KvSrc<f16>{a.self_k + base}, KvSrc<f16>{a.self_v + base}, n_load, get_n, qfix, &raw, red, osum, o_l, &m, &l))
        return;
    if (threadIdx.x < 64) store_att(a, b, h * kHeadDim + threadIdx.x, o_l[threadIdx.x] / l);
}

// Beam search: row r of slot b's history lives in the cache of slot owner[b][r] (the beam that computed it, or the audio's pre-fill slot);
// a new beam that continues another beam's sequence inherits that beam's owner row on the host - nothing is copied on the device
// (openai/whisper's rearrange_kv_cache moved ~8 MB per re-parented beam and position at large-v3: 18 % of the beam pass, profiles/r03i_*).
// One dependent load more than dec_self_attn_kernel (the owners), the same arithmetic.  No reference behaviour (there is no beam search).
template <int PASSES>
__global__ __launch_bounds__(256) void dec_self_attn_owner_kernel(const AttnArgs a) {
    __shared__ float red[16], osum[256], o_l[64];
    const int h = blockIdx.x, b = blockIdx.y;
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;
    const int d = a.d, H = a.n_head;
    const int n_load = min(PASSES * 32, a.self_rows > 0 ? a.self_rows : PASSES * 32);
    const int part = threadIdx.x & 7, kg = threadIdx.x >> 3;
    size_t off[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = min(kg + 32 * i, n_load - 1);
        const int own = min(max(a.self_owner[(size_t)b * kMaxTok + key], 0), a.batch - 1);
        off[i] = (((size_t)own * H + h) * kMaxTok + key) * kHeadDim + part * 8;
    }
    KvPiece<f16> kreg[PASSES], vreg[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) load_kv8<false>(KvSrc<f16>{a.self_k}, off[i], kreg[i]);
#pragma unroll
    for (int i = 0; i < PASSES; ++i) load_kv8<false>(KvSrc<f16>{a.self_v}, off[i], vreg[i]);
    float qv[8];
    attend_load_q(a.q + (size_t)b * d + h * kHeadDim, qv);
    if (!(s_act && !s_done)) return;            // workgroup-uniform
    const int n = min(min(max(s_ti, 0), kMaxTok - 1) + 1, n_load);
    float m, l;
    attend_compute<PASSES, f16>(qv, kreg, vreg, n, nullptr, red, osum, o_l, &m, &l, nullptr);
    if (threadIdx.x < 64) store_att(a, b, h * kHeadDim + threadIdx.x, o_l[threadIdx.x] / l);
}

// The last arriver of a (slot, head): all S partials (S x 66 floats) are fetched by the whole workgroup in ONE round of independent sc1
// loads into LDS (a per-thread loop over the splits is S dependent L2 round trips: 24 us at S = 24) and combined from there in index order.
__device__ __forceinline__ void combine_splits(const AttnArgs& a, int b, int h, int S) {
    __shared__ float pl[kMaxSplit * 66];
    const int tid = threadIdx.x;
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

// (Round 5, measured and rejected, profiles/r05o_*: at width <= 384 the cross-QUERY projection folded into this kernel - a (split, head, slot)
// workgroup computing its head's 64 query channels itself from the LayerNorm-2 planes (49 KB of W_cq at d = 384, four threads per channel) while its
// K / V rows are in flight, so that dec32_proj<P32_Q>'s launch disappears.  Parity-green (the whole GPU suite), and slower: the query phase is twelve
// dependent L2 round trips that the row stream does not hide, and it is redundant across key splits and slots - tiny.en at 1 slot 10.8 -> 17.7 us
// per launch (0.2266 -> 0.2365 ms per decoder step with the cross-query launch gone), at 8 slots 18.3 -> 42.4 us.  Code in git history.)
template <int PASSES, bool NT>
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const AttnArgs a) {
    constexpr int KPB = PASSES * 32;
    __shared__ float red[16], osum[256], o_l[64];
    __shared__ int last_flag;
    const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    gate_release_if_mine(a);
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;     // looked at after the K/V loads are issued
    const int d = a.d, S = a.n_split;
    const int t0 = sp * KPB, n = min(KPB, kCtx - t0);
    const int bc = a.cross_div > 1 ? b / a.cross_div : b;       // the slot whose cross K / V this slot reads
    const size_t base = (((size_t)bc * a.n_head + h) * kCtx + t0) * kHeadDim;
    int slot = -1;
    if (a.align) slot = a.align_slot[a.layer * a.n_head + h];
    float m, l;
    ATT_STAMP(0);
    unsigned long long* stamp = a.dbg ? a.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 4096 * 8 : nullptr;
    // alignment-head row: DecodingCache.alignmentWeights row tokenIndex + 1 (TextDecoder.swift:272-296), raw scores here,
    // softmax + head mean in alignment_mean_kernel
    float* raw = nullptr;
    auto get_n = [&]() {
        if (!(s_act && !s_done)) return -1;
        const int pos = min(max(s_ti, 0), kMaxTok - 1);
        if (slot >= 0 && pos + 1 < kMaxTok) raw = a.align + (((size_t)b * kMaxTok + pos + 1) * a.n_align + slot) * kCtx + t0;
        return n;
    };
    auto qfix = [](float (&)[8], int) {};
    if (!attend_block<PASSES, NT, hr24>(a.q + (size_t)b * d + h * kHeadDim, KvSrc<hr24>{a.cross_k_hi + base, a.cross_k_lo + base}, KvSrc<hr24>{a.cross_v_hi + base, a.cross_v_lo + base},
                                        n, get_n, qfix, &raw, red, osum, o_l, &m, &l, stamp))
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
    ATT_STAMP(4);
    if (last_flag && tid == 0 && a.dbg) stamp[6] = 1;
    if (last_flag) combine_splits(a, b, h, S);      // workgroup-uniform
    ATT_STAMP(5);
}

// ---------------------------------------------------------------------------------------------- sampler
constexpr int SAMP_T = 1024;
constexpr int SAMP_E = 51;   // ceil(51866 / 1024)
static_assert(SAMP_T * SAMP_E >= kMaxVocab, "sampler_kernel keeps the whole logits row in registers");

struct BlockRed {
    float f[32];
    int i[32];
};

__device__ __forceinline__ float block_max(float v, BlockRed* br) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) br->f[wave] = v;
    __syncthreads();
    float r = br->f[0];
    for (int w = 1; w < SAMP_T / 64; ++w) r = fmaxf(r, br->f[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, BlockRed* br) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) br->f[wave] = v;
    __syncthreads();
    float r = 0.0f;
    for (int w = 0; w < SAMP_T / 64; ++w) r += br->f[w];
    return r;
}
// argmax with ties -> smallest index
__device__ __forceinline__ void block_argmax(float v, int idx, BlockRed* br, float* vout, int* iout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, 64);
        int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (lane == 0) { br->f[wave] = v; br->i[wave] = idx; }
    __syncthreads();
    float bv = br->f[0];
    int bi = br->i[0];
    for (int w = 1; w < SAMP_T / 64; ++w) {
        float ov = br->f[w];
        int oi = br->i[w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    *vout = bv;
    *iout = bi;
}

__device__ __forceinline__ float uniform01(unsigned long long seed, int counter) {
    unsigned long long z = seed + (unsigned long long)(counter + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// decodeText bookkeeping after one sampled token (TextDecoder.swift:573-757), then the filter rules of the next step
// last_ts_pre: index of the last timestamp token in [0, n_tok) BEFORE this step's token is appended (-1 none), or -2 = unknown (scan)
__device__ __forceinline__ void advance_decode_state(const SamplerCfg& cfg, SeqState* sq, int tok, float lp, int n_tok, int last_ts_pre = -2) {
    const int tb = cfg.time_token_begin;
    const int ti_cur = sq->token_index;
    const bool isFirstToken = ti_cur == cfg.prefilled_index;
    const bool tooLow = isFirstToken && cfg.has_first_token_threshold && lp < cfg.first_token_log_prob_threshold;
    const bool completed = tok == cfg.end_token;
    const bool segDone = completed || n_tok >= kMaxTok - 1 || tooLow;
    sq->steps += 1;
    sq->first_token_too_low = tooLow ? 1 : 0;
    if (segDone) {
        sq->done = 1;
        return;
    }
    const bool isPrefill = ti_cur < sq->prompt_len - 1;
    int nt = n_tok;
    if (!isPrefill) { sq->tokens[nt] = tok; sq->logprobs[nt] = lp; nt += 1; sq->n_tokens = nt; }
    const int ti_next = ti_cur + 1;
    if (ti_next >= cfg.loop_count) {
        sq->done = 1;
        return;
    }
    int next = tok;
    if (ti_next < sq->prompt_len) {                                   // :581-594 (loop top of the next iteration)
        const bool isLast = ti_next == sq->prompt_len - 1;
        const bool isTs = sq->tokens[ti_next] >= tb;
        const bool predTs = next >= tb;
        if (!(isLast && isTs && predTs)) next = sq->tokens[ti_next];
        else sq->tokens[ti_next] = next;
    }
    sq->token_index = ti_next;
    sq->next_token = next;
    // (the prompt-timestamp replacement above writes a timestamp over a timestamp: an index found before it still points at one)
    int hint = last_ts_pre;
    if (hint >= -1 && nt > n_tok && tok >= tb) hint = nt - 1;          // the token appended by this step is the newest timestamp
    compute_filter_rules(cfg, sq, nt, cfg.n_vocab, sq->f_rules, hint);
}

// MODE bit 0: apply filters; bit 1: sample; bit 2: advance decodeText state; bit 3: write filtered logits back
// TOPK > 0 (beam search, BeamSearchTokenSampler.update's device part, no reference behaviour): after the filters, the log-softmax of the
// row and its TOPK best entries, descending, ties to the lower id (openai/whisper decoding.py:361 logprobs[idx].topk(beam_size + 1)) go to
// logprob_out / token_out[b * kBeamTopK + k]; the row stays in registers (the two-kernel form re-read it K + 3 times through L2).
template <int DO_FILTER, int DO_SAMPLE, int DO_ADVANCE, int WRITE_BACK, int TOPK = 0>
__global__ __launch_bounds__(SAMP_T) void sampler_kernel(const SamplerCfg* __restrict__ cfgp, const int* __restrict__ suppress,
                                                        SeqState* __restrict__ seqs, float* __restrict__ logits_all,
                                                        int counter_override, int* __restrict__ token_out, float* __restrict__ logprob_out) {
    __shared__ BlockRed br;
    __shared__ int sh_i[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    SeqState* sq = seqs + b;
    if ((DO_ADVANCE || TOPK) && !slot_live(sq)) return;
    const SamplerCfg cfg = *cfgp;
    const int V = cfg.n_vocab;
    float* logits = logits_all + (size_t)b * V;
    const int n_tok = sq->n_tokens;
    const int tb = cfg.time_token_begin;

    // ---- scalar filter parameters (thread 0), restating LogitsFilter.swift
    // sh_i: 0 blank_active, 1 ts_active, 2 r1_lo, 3 r1_hi, 4 r2_lo, 5 r2_hi
    if (tid == 0) {
        int r[6] = {0, 0, 0, 0, 0, 0};
        if (DO_FILTER) compute_filter_rules(cfg, sq, n_tok, V, r);
#pragma unroll
        for (int i = 0; i < 6; ++i) sh_i[i] = r[i];
    }
    if (DO_FILTER && !cfg.language_filter) {
        for (int i = tid; i < cfg.n_suppress; i += SAMP_T) {                       // SuppressTokensFilter :21-24
            int t = suppress[i];
            if (t >= 0 && t < V) logits[t] = -INFINITY;
        }
    }
    __syncthreads();
    const int blank = sh_i[0], ts_active = sh_i[1], r1lo = sh_i[2], r1hi = sh_i[3], r2lo = sh_i[4], r2hi = sh_i[5];

    float x[SAMP_E];
    float mx_text = -INFINITY, mx_ts = -INFINITY;
#pragma unroll
    for (int e = 0; e < SAMP_E; ++e) {
        int n = tid + SAMP_T * e;
        float v = -INFINITY;
        if (n < V) {
            v = logits[n];
            if (cfg.f16_logits) v = (float)(f16)v;          // idempotent for logits the projection kernel already rounded
            if (DO_FILTER) {
                if (cfg.language_filter) {                                           // LanguageLogitsFilter :259-265
                    if (n < cfg.language_token_begin || n >= cfg.language_token_begin + cfg.n_language_tokens) v = -INFINITY;
                } else {
                    if (blank && (n == cfg.whitespace_token || n == cfg.end_token)) v = -INFINITY;
                    if (ts_active && (n == cfg.no_timestamps_token || (n >= r1lo && n < r1hi) || (n >= r2lo && n < r2hi))) v = -INFINITY;
                }
            }
            if (n < tb) mx_text = fmaxf(mx_text, v); else mx_ts = fmaxf(mx_ts, v);
        }
        x[e] = v;
    }
    if (DO_FILTER && ts_active && !cfg.language_filter) {
        // sumOfProbabilityOverTimestampsIsAboveAnyOtherToken (:144-242): logsumexp(ts) > max(text)
        const float m_text = block_max(mx_text, &br);
        const float m_ts = block_max(mx_ts, &br);
        float s = 0.0f, st = 0.0f;
#pragma unroll
        for (int e = 0; e < SAMP_E; ++e) {
            int n = tid + SAMP_T * e;
            if (n < V && x[e] != -INFINITY) { if (n >= tb) s += expf(x[e] - m_ts); else st += expf(x[e] - m_text); }
        }
        s = block_sum(s, &br);
        st = block_sum(st, &br);
        const SoftStat t_{m_text, st, 0}, u_{m_ts, s, 0};
        const bool cond = timestamp_mass_wins(t_, u_, cfg.f16_logits != 0);
        if (cond) {
#pragma unroll
            for (int e = 0; e < SAMP_E; ++e) {
                int n = tid + SAMP_T * e;
                if (n < tb) x[e] = -INFINITY;
            }
        }
    }
    if (WRITE_BACK) {
#pragma unroll
        for (int e = 0; e < SAMP_E; ++e) {
            int n = tid + SAMP_T * e;
            if (n < V) logits[n] = x[e];
        }
    }
    if (TOPK) {
        const int K = min(counter_override, kBeamTopK);      // (the counter argument carries K in this mode)
        float lm = -INFINITY;
        int li = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < SAMP_E; ++e) {
            const int n = tid + SAMP_T * e;
            if (n < V && (x[e] > lm || (x[e] == lm && n < li))) { lm = x[e]; li = n; }
        }
        float gmax; int gidx;
        block_argmax(lm, li, &br, &gmax, &gidx);
        float se = 0.0f;
#pragma unroll
        for (int e = 0; e < SAMP_E; ++e) {
            const int n = tid + SAMP_T * e;
            if (n < V && x[e] != -INFINITY) se += expf(x[e] - gmax);
        }
        se = block_sum(se, &br);
        const float lse = gmax + logf(se);
        float bv = gmax; int bi = gidx;
        unsigned long long taken = 0;                     // this thread's ids already reported (bit e)
        for (int k = 0; k < K; ++k) {
            if (tid == 0) { logprob_out[(size_t)b * kBeamTopK + k] = bv - lse; token_out[(size_t)b * kBeamTopK + k] = bi; }
            if (k + 1 == K) break;
            float m2 = -INFINITY; int i2 = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < SAMP_E; ++e) {
                const int n = tid + SAMP_T * e;
                if (n == bi) taken |= 1ull << e;
                if (n < V && !((taken >> e) & 1) && (x[e] > m2 || (x[e] == m2 && n < i2))) { m2 = x[e]; i2 = n; }
            }
            block_argmax(m2, i2, &br, &bv, &bi);
        }
        return;
    }
    if (!DO_SAMPLE) return;

    // ---- GreedyTokenSampler (TokenSampler.swift:29-252)
    const float temp = sq->temperature;
    if (temp != 0.0f) {
        const float alpha = 1.0f / temp;
#pragma unroll
        for (int e = 0; e < SAMP_E; ++e) x[e] *= alpha;
    }
    float lm = -INFINITY;
    int li = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < SAMP_E; ++e) {
        int n = tid + SAMP_T * e;
        if (n < V && (x[e] > lm || (x[e] == lm && n < li))) { lm = x[e]; li = n; }
    }
    float gmax; int gidx;
    block_argmax(lm, li, &br, &gmax, &gidx);
    float se = 0.0f;
#pragma unroll
    for (int e = 0; e < SAMP_E; ++e) {
        int n = tid + SAMP_T * e;
        if (n < V && x[e] != -INFINITY) se += expf(x[e] - gmax);
    }
    se = block_sum(se, &br);
    const float lse = gmax + logf(se);
    int tok = gidx;
    float lp = gmax - lse;
    if (temp != 0.0f) {
        // top-k multinomial (BNNS path :140-180): k block-wide argmax passes, descending, ties -> lower id
        const int k = min(cfg.top_k, 8);
        float tv[8]; int ti[8];
        tv[0] = gmax; ti[0] = gidx;
        for (int j = 1; j < k; ++j) {
            float m2 = -INFINITY; int i2 = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < SAMP_E; ++e) {
                int n = tid + SAMP_T * e;
                if (n >= V) continue;
                bool taken = false;
                for (int q = 0; q < j; ++q) taken |= (ti[q] == n);
                if (!taken && (x[e] > m2 || (x[e] == m2 && n < i2))) { m2 = x[e]; i2 = n; }
            }
            block_argmax(m2, i2, &br, &tv[j], &ti[j]);
        }
        float pr[8], total = 0.0f;
        for (int j = 0; j < k; ++j) { pr[j] = expf(tv[j] - lse); total += pr[j]; }
        const int counter = DO_ADVANCE ? sq->token_index : counter_override;
        const float rnd = uniform01(cfg.seed + (unsigned long long)b * 0x632BE59BD9B4E019ull, counter) * total;
        float accp = 0.0f;
        int chosen = 0;
        for (int j = 0; j < k; ++j) {
            accp += pr[j];
            if (rnd < accp) { chosen = j; break; }
        }
        tok = ti[chosen];
        lp = tv[chosen] - lse;
    }
    if (tid == 0) {
        if (token_out) { token_out[b] = tok; logprob_out[b] = lp; }
        if (DO_ADVANCE) advance_decode_state(cfg, sq, tok, lp, n_tok);
    }
}

// ---------------------------------------------------------------------------------------------- fused greedy sampler, part 2
__device__ __forceinline__ void stat_wave_reduce(SoftStat& a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float om = __shfl_xor(a.m, o, 64), os = __shfl_xor(a.s, o, 64);
        int oi = __shfl_xor(a.i, o, 64);
        stat_merge(a, om, os, oi);
    }
}

// One workgroup per slot: merge the per-workgroup (text, timestamp) statistics of the logits kernel, apply the
// "timestamp mass beats every text token" rule (LogitsFilter.swift:144-242), take the greedy token and its log-prob
// (TokenSampler.swift:29-252, T = 0) and advance the decodeText state.
__global__ __launch_bounds__(256) void sampler_final_kernel(const SamplerCfg* __restrict__ cfgp, SeqState* __restrict__ seqs,
                                                            const float* __restrict__ stats, int nblk) {
    __shared__ float sm[4][4];
    __shared__ int si[4][3];
    __shared__ SeqState sq_l;     // the slot's whole decode state: thread 0's bookkeeping (token history scans, appends) runs on
                                  // this LDS copy instead of a chain of dependent global round trips
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SeqState* sq = seqs + b;
    if (!slot_live(sq)) return;
    constexpr int kWords = sizeof(SeqState) / 4;
    static_assert(kMaxTok <= 256, "one history token per thread");
    for (int i = tid; i < kWords; i += 256) reinterpret_cast<int*>(&sq_l)[i] = reinterpret_cast<const int*>(sq)[i];
    // index of the last timestamp token of the history, found by all threads (the rules of the NEXT step need it: compute_filter_rules)
    const int n_hist = sq->n_tokens;
    const int tbeg = cfgp->time_token_begin;
    int last_ts = (tid < n_hist && tid < kMaxTok && sq->tokens[tid] >= tbeg) ? tid : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last_ts = max(last_ts, __shfl_xor(last_ts, o, 64));
    SoftStat t{-INFINITY, 0.0f, 0x7fffffff}, u{-INFINITY, 0.0f, 0x7fffffff};
    constexpr int NR = kStatBlocks / 256;      // records per thread: all loads are issued before the first merge (one L2 round
    float4 lo[NR];                             // trip instead of NR dependent ones)
    float2 hi[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int i = tid + 256 * k;
        const float* e = stats + ((size_t)b * kStatBlocks + min(i, nblk - 1)) * 8;
        lo[k] = *reinterpret_cast<const float4*>(e);
        hi[k] = *reinterpret_cast<const float2*>(e + 4);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (tid + 256 * k < nblk) {
            stat_merge(t, lo[k].x, lo[k].y, __float_as_int(lo[k].z));
            stat_merge(u, lo[k].w, hi[k].x, __float_as_int(hi[k].y));
        }
    }
    stat_wave_reduce(t);
    stat_wave_reduce(u);
    if (lane == 0) { sm[wave][0] = t.m; sm[wave][1] = t.s; si[wave][0] = t.i; sm[wave][2] = u.m; sm[wave][3] = u.s; si[wave][1] = u.i; si[wave][2] = last_ts; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { stat_merge(t, sm[w][0], sm[w][1], si[w][0]); stat_merge(u, sm[w][2], sm[w][3], si[w][1]); last_ts = max(last_ts, si[w][2]); }
        const SamplerCfg cfg = *cfgp;
        const bool ts_active = sq_l.f_rules[1] != 0;
        int tok; float lp;
        const bool cond = ts_active && timestamp_mass_wins(t, u, cfg.f16_logits != 0);
        if (cond || t.m == -INFINITY) {          // text ids masked: the candidates are the timestamp ids
            tok = u.i; lp = -logf(u.s);
        } else {
            SoftStat g = t;
            stat_merge(g, u.m, u.s, u.i);        // equal maxima: the text id (smaller index) wins, like a first-maximum argmax
            tok = g.i; lp = -logf(g.s);
        }
        advance_decode_state(cfg, &sq_l, tok, lp, sq_l.n_tokens, last_ts);
    }
    __syncthreads();
    for (int i = tid; i < kWords; i += 256) reinterpret_cast<int*>(sq)[i] = reinterpret_cast<const int*>(&sq_l)[i];
}

__global__ void rules_init_kernel(const SamplerCfg* __restrict__ cfgp, SeqState* __restrict__ seqs) {
    if (threadIdx.x != 0) return;
    SeqState* sq = seqs + blockIdx.x;
    const SamplerCfg cfg = *cfgp;
    compute_filter_rules(cfg, sq, sq->n_tokens, cfg.n_vocab, sq->f_rules);
}
void launch_rules_init(const SamplerCfg* cfg_dev, SeqState* seq, int batch, hipStream_t st) { rules_init_kernel<<<batch, 64, 0, st>>>(cfg_dev, seq); }

// ---------------------------------------------------------------------------------------------- launchers
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// WH_DBG=1 timeline probe buffer [KK_COUNT][4096][8], allocated once per process (thread-safe function-local static)
unsigned long long* debug_buffer() {
    static unsigned long long* const buf = [] {
        unsigned long long* p = nullptr;
        const char* e = getenv("WH_DBG");
        if (e && e[0] == '1') {
            const size_t bytes = (size_t)KK_COUNT * 4096 * 8 * 8;
            if (hipMalloc((void**)&p, bytes) != hipSuccess) p = nullptr;
            else (void)hipMemset(p, 0, bytes);
        }
        return p;
    }();
    return buf;
}

int cross_attn_splits(int batch, int n_head) {
    // Keys per workgroup 256 / 128 / 64, chosen from the head count ONLY: the number of splits fixes the order in which the
    // partial softmax sums are combined, so it must not depend on the batch - a slot decodes to the same bits alone or in a
    // batch of 32 (tests/test_gpu_dims.py).  With >= 12 heads even a batch of 8 gives >= 576 workgroups at 6 splits
    // (measured large-v3: 8 passes 53.0 us at 32 slots / 20.4 at 8; 12 passes 53.8 / 20.8; 16 passes 58.5 / 20.7).
    (void)batch;
    static const int forced = env_int("WH_XATT_PASSES", 0);     // tuning knob: 8 / 6 / 4 / 2
    // measured large-v3, 64 slots (profiles/r03a_*): 6 passes (8 splits, 70 registers, 7 waves per SIMD) 5.20 ms per decoder step against
    // 5.38 with 8 passes (6 splits, 87 registers); 1754 vs 1715 audio-s/s with three sessions in flight
    // round 5: the rows are 24-bit (24 bytes per lane and key: a pass carries 1.5 x the bytes of a Float16 pass): 4 passes at >= 12 heads
    // (48 KB in flight per workgroup; the round-3 choice of 6 Float16 passes held the same), 2 passes below
    int passes = forced ? forced : (n_head >= 12 ? 4 : 2);
    passes = passes > 6 ? 8 : passes > 4 ? 6 : passes > 2 ? 4 : 2;            // instantiated: 2 / 4 / 6 / 8 passes = 24 / 12 / 8 / 6 splits
    return (kCtx + passes * 32 - 1) / (passes * 32);
}

static void launch_self_attn(const AttnArgs& at, int H, int B, hipStream_t st) {
    ProfScope ps_(KK_DEC_SELF_ATTN, st);
    const dim3 grid(H, B);
    const int passes = at.self_rows > 0 ? (at.self_rows + 31) / 32 : 7;
    if (at.self_owner) {
        switch (passes) {
            case 1: dec_self_attn_owner_kernel<1><<<grid, 256, 0, st>>>(at); break;
            case 2: dec_self_attn_owner_kernel<2><<<grid, 256, 0, st>>>(at); break;
            case 3: dec_self_attn_owner_kernel<3><<<grid, 256, 0, st>>>(at); break;
            case 4: dec_self_attn_owner_kernel<4><<<grid, 256, 0, st>>>(at); break;
            case 5: dec_self_attn_owner_kernel<5><<<grid, 256, 0, st>>>(at); break;
            case 6: dec_self_attn_owner_kernel<6><<<grid, 256, 0, st>>>(at); break;
            default: dec_self_attn_owner_kernel<7><<<grid, 256, 0, st>>>(at);
        }
        return;
    }
    switch (passes) {
        case 1: dec_self_attn_kernel<1><<<grid, 256, 0, st>>>(at); break;
        case 2: dec_self_attn_kernel<2><<<grid, 256, 0, st>>>(at); break;
        case 3: dec_self_attn_kernel<3><<<grid, 256, 0, st>>>(at); break;
        case 4: dec_self_attn_kernel<4><<<grid, 256, 0, st>>>(at); break;
        case 5: dec_self_attn_kernel<5><<<grid, 256, 0, st>>>(at); break;
        case 6: dec_self_attn_kernel<6><<<grid, 256, 0, st>>>(at); break;
        default: dec_self_attn_kernel<7><<<grid, 256, 0, st>>>(at);
    }
}

static void launch_cross_attn(const AttnArgs& at_in, int S, int H, int B, hipStream_t st) {
    static const int nofence = env_int("WH_XATT_NOFENCE", 1);   // sc1 stores + sc1 loads need no acquire (MI355X_MICROARCH.md R1); 0 restores it (A/B)
    AttnArgs at = at_in;
    at.no_fence = nofence;
    ProfScope ps_(KK_DEC_CROSS_ATTN, st);
    const dim3 grid(S, H, B);
    // the gate goes back when workgroup (last - lead) is dispatched; WH_XATT_GATE_LEAD = workgroups before the end (tuning knob)
    static const int gate_lead = env_int("WH_XATT_GATE_LEAD", 0);
    at.gate_wg = std::max(0, S * H * B - 1 - gate_lead);
    static const int xlds = env_int("WH_XATT_LDS", 0);   // tuning knob: extra LDS per workgroup caps the residency
    // non-temporal K / V loads (each row is read once per step; measured large-v3, 32 slots: 51.9 -> 49.7 us per launch, 3 sessions in
    // flight 13.4 k -> 14.2 k sequence-steps/s, profiles/r02i_*); WH_XATT_NT=0 is the A/B side
    static const int nt = env_int("WH_XATT_NT", 1);
    const bool ntl = nt && at.cross_div <= 1;      // (cross_div > 1, beam search: cacheable loads - the L2 of the XCD serves the other beams of the audio)
#define XATT(P_) do { if (ntl) dec_cross_attn_kernel<P_, true><<<grid, 256, xlds, st>>>(at); else dec_cross_attn_kernel<P_, false><<<grid, 256, xlds, st>>>(at); } while (0)
    if (S == 6) XATT(8);
    else if (S == 8) XATT(6);
    else if (S == 12) XATT(4);
    else XATT(2);
#undef XATT
}

// One decoder step for all slots: embed -> per layer [QKV, self-attention, out projection, cross query, cross-attention, cross out
// projection, fc1, fc2] -> logits -> sampler.  Every activation hand-off is a plane pair in B-fragment order, every LayerNorm is
// folded into the consumer's epilogue (see decoder32.hip).
void launch_decoder_step(const DecodeBuffers& db, const SamplerCfg* cfg_dev, const int* suppress_dev, bool sample, hipStream_t st) {
    const Dec32& D = *db.d32;
    const int d = db.d, B = db.batch, H = db.n_head, L = db.n_layer, V = db.n_vocab;
    const int n_bt = (B + 31) / 32;
    const size_t self_stride = (size_t)db.max_batch * H * kMaxTok * kHeadDim;
    const size_t cross_stride = (size_t)db.max_batch * H * kCtx * kHeadDim;
    const int S = cross_attn_splits(B, H);
    launch_dec32_embed(db.emb, db.pos, db.seq, B, d, V, n_bt, D.x, db.layers_host[0].ln1_g, D.za_hi, D.za_lo, D.stat, st);
    P32Args base{};
    base.batch = B; base.d = d; base.n_head = H; base.n_vocab = V; base.seq = db.seq; base.part = D.part; base.ticket = D.ticket;
    base.x = D.x; base.stat_in = D.stat; base.n_stat = d / 32;
    for (int l = 0; l < L; ++l) {
        const DecLayerW& w = db.layers_host[l];
        const Dec32LayerW& t = D.layers_host[l];
        P32Args a = base;       // LN1 (folded) + QKV: q (f32), k / v into the self-attention cache at token_index
        a.N = 3 * d; a.K = d; a.Wt = t.qkv_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = t.qkv_g; a.fold_c = t.qkv_c; a.q = D.q;
        a.self_k = db.self_k + (size_t)l * self_stride; a.self_v = db.self_v + (size_t)l * self_stride; a.prof_kind = KK_DEC_QKV;
        launch_dec32_proj(P32_QKV, a, n_bt, st);
        AttnArgs at{};
        at.batch = B; at.d = d; at.n_head = H; at.layer = l; at.n_layer = L; at.n_split = S; at.q = D.q; at.cross_div = db.cross_div;
        at.self_k = a.self_k; at.self_v = a.self_v;
        at.cross_k_hi = db.cross_k_hi + (size_t)l * cross_stride; at.cross_k_lo = db.cross_k_lo + (size_t)l * cross_stride;
        at.cross_v_hi = db.cross_v_hi + (size_t)l * cross_stride; at.cross_v_lo = db.cross_v_lo + (size_t)l * cross_stride;
        at.att_hi = D.zb_hi; at.att_lo = D.zb_lo; at.part = db.part; at.ticket = db.ticket; at.seq = db.seq;
        at.align = db.align; at.align_slot = db.align_slot; at.n_align = db.n_align; at.gate = db.xattn_gate;
        at.self_rows = db.self_rows; at.self_owner = db.self_owner;
        launch_self_attn(at, H, B, st);
        a = base;               // x += W_o att + b_o; planes gamma_2 x, statistics for LN2
        a.N = d; a.K = d; a.Wt = t.o_t; a.zhi = D.zb_hi; a.zlo = D.zb_lo; a.bias = w.o_b; a.gamma_next = w.ln2_g;
        a.zhi_out = D.za_hi; a.zlo_out = D.za_lo; a.stat_out = D.stat; a.prof_kind = KK_DEC_OPROJ;
        launch_dec32_proj(P32_RESID, a, n_bt, st);
        a = base;               // LN2 (folded) + cross-attention query
        a.N = d; a.K = d; a.Wt = t.cq_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = t.cq_g; a.fold_c = t.cq_c; a.q = D.q; a.prof_kind = KK_DEC_CQ;
        a.gate = db.xabs ? nullptr : db.xattn_gate;
        launch_dec32_proj(P32_Q, a, n_bt, st);
        if (db.xabs) {          // weight-absorbed cross-attention over the encoder output (xabs.hip): no per-layer K / V rows
            const Xabs& X = *db.xabs;
            XabsArgs xa{};
            xa.batch = B; xa.max_batch = db.max_batch; xa.d = d; xa.n_head = H; xa.layer = l; xa.n_split = X.n_split; xa.cross_div = db.cross_div;
            xa.enc = X.enc; xa.q = D.q; xa.wkT = X.layers_host[l].wkT; xa.wv_t = X.layers_host[l].wv_t; xa.bv = X.layers_host[l].bv;
            xa.qf_hi = X.qf_hi; xa.qf_lo = X.qf_lo; xa.part = X.part; xa.ml = X.ml; xa.att_hi = D.zb_hi; xa.att_lo = D.zb_lo;
            xa.align = db.align; xa.align_slot = db.align_slot; xa.n_align = db.n_align; xa.seq = db.seq; xa.kpart = D.part; xa.ticket = D.ticket;
            xa.gate = db.xattn_gate;
            xa.spw = X.spw;
            xa.dbg = debug_buffer() ? debug_buffer() + (size_t)KK_DEC_CROSS_ATTN * 4096 * 8 : nullptr;
            launch_xabs_qk(xa, n_bt, st);
            // (Round 6, measured and rejected, profiles/r06x_*, r06y_*: xabs_attn on a CU-masked HIP stream of its own - the stream kernels of all sessions confined to the first
            // n CUs, the launch chain on the rest or everywhere - joined to the session stream by an event pair per layer, eager launches: 2796 -> 2314 - 2387 audio-s/s for
            // n = 128 .. 255, i.e. the two cross-queue hops per layer cost 15 % before any partition can pay; with the chain confined to the other 96 CUs 1830.)
            launch_xabs_attn(xa, st);
            launch_xabs_vup(xa, n_bt, st);
        } else {
            at.dbg = debug_buffer() ? debug_buffer() + (size_t)KK_DEC_CROSS_ATTN * 4096 * 8 : nullptr;
            launch_cross_attn(at, S, H, B, st);
        }
        a = base;               // x += W_co att + b_co; planes gamma_3 x, statistics for LN3
        a.N = d; a.K = d; a.Wt = t.co_t; a.zhi = D.zb_hi; a.zlo = D.zb_lo; a.bias = w.co_b; a.gamma_next = w.ln3_g;
        a.zhi_out = D.za_hi; a.zlo_out = D.za_lo; a.stat_out = D.stat; a.prof_kind = KK_DEC_COPROJ;
        launch_dec32_proj(P32_RESID, a, n_bt, st);
        a = base;               // LN3 (folded) + fc1 + GELU -> f16 plane
        a.N = 4 * d; a.K = d; a.Wt = t.fc1_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = t.fc1_g; a.fold_c = t.fc1_c; a.h_out = D.h; a.h_out_lo = D.h_lo; a.prof_kind = KK_DEC_FC1;
        launch_dec32_proj(P32_FC1, a, n_bt, st);
        a = base;               // x += W_2 h + b_2; planes of the next layer's LN1 (or the final LayerNorm)
        a.N = d; a.K = 4 * d; a.Wt = t.fc2_t; a.zhi = D.h; a.zlo = D.h_lo; a.bias = w.fc2_b;
        a.gamma_next = (l + 1 < L) ? db.layers_host[l + 1].ln1_g : db.lnf_g;
        a.zhi_out = D.za_hi; a.zlo_out = D.za_lo; a.stat_out = D.stat; a.prof_kind = KK_DEC_FC2;
        launch_dec32_proj(P32_RESID, a, n_bt, st);
    }
    const bool fused = sample && db.fused_greedy;
    P32Args a = base;           // final LayerNorm (folded) + tied-embedding logits (+ the fused greedy sampler statistics)
    a.N = V; a.K = d; a.Wt = D.emb_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = D.lg_g; a.fold_c = D.lg_c;
    a.logits = fused ? nullptr : db.logits; a.prof_kind = KK_DEC_LOGITS;
    if (sample) a.cfg = cfg_dev;                 // Float16-logits switch
    if (fused) { a.stats = db.stats; a.sup_mask = db.sup_mask; }
    launch_dec32_proj(P32_LOGITS, a, n_bt, st);
    if (fused) {
        ProfScope ps_(KK_SAMPLER, st);
        sampler_final_kernel<<<B, 256, 0, st>>>(cfg_dev, db.seq, db.stats, (V + 31) / 32);
    } else if (sample) {
        ProfScope ps_(KK_SAMPLER, st);
        sampler_kernel<1, 1, 1, 0><<<B, SAMP_T, 0, st>>>(cfg_dev, suppress_dev, db.seq, db.logits, 0, nullptr, nullptr);
    }
}

void launch_filter_only(const SamplerCfg* cfg_dev, const int* suppress_dev, SeqState* seq, float* logits, int n_vocab, hipStream_t st) {
    sampler_kernel<1, 0, 0, 1><<<1, SAMP_T, 0, st>>>(cfg_dev, suppress_dev, seq, logits, 0, nullptr, nullptr);
}
void launch_sample_only(const SamplerCfg* cfg_dev, SeqState* seq, float* logits, int n_vocab, int counter, int* token_out, float* logprob_out, hipStream_t st) {
    sampler_kernel<0, 1, 0, 0><<<1, SAMP_T, 0, st>>>(cfg_dev, nullptr, seq, logits, counter, token_out, logprob_out);
}


// ---------------------------------------------------------------------------------------------- beam search support
void launch_beam_filter_topk(const SamplerCfg* cfg_dev, const int* suppress_dev, SeqState* seq, float* logits, int batch, int K, float* lp_out,
                             int* tok_out, hipStream_t st) {
    sampler_kernel<1, 0, 0, 0, 1><<<batch, SAMP_T, 0, st>>>(cfg_dev, suppress_dev, seq, logits, K, tok_out, lp_out);
}

void launch_filter_sample(const SamplerCfg* cfg_dev, const int* suppress_dev, SeqState* seq, float* logits, int batch, int* token_out, float* logprob_out, hipStream_t st) {
    sampler_kernel<1, 1, 0, 0><<<batch, SAMP_T, 0, st>>>(cfg_dev, suppress_dev, seq, logits, 0, token_out, logprob_out);
}

// align [B][224][n_align][1500] raw cross-attention score rows of the alignment heads -> out [B][224][1500]:
// softmax over the 1500 positions per head, then the mean over heads (never-written rows are all-zero and stay zero).
// One workgroup per (slot, row); wave w takes heads w, w + 4, ...
__global__ __launch_bounds__(256) void alignment_mean_kernel(const float* __restrict__ align, int n_align, float* __restrict__ out) {
    __shared__ float acc[4][kCtx];
    const size_t row = blockIdx.x;  // b * 224 + pos
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* src = align + row * n_align * kCtx;
    for (int t = lane; t < kCtx; t += 64) acc[wave][t] = 0.0f;
    for (int j = wave; j < n_align; j += 4) {
        const float* sp = src + (size_t)j * kCtx;
        float v[24];
        float mx = -INFINITY, amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            int t = lane + 64 * i;
            v[i] = t < kCtx ? sp[t] : -INFINITY;
            mx = fmaxf(mx, v[i]);
            if (t < kCtx) amax = fmaxf(amax, fabsf(v[i]));
        }
        mx = wave_max(mx);
        amax = wave_max(amax);
        if (amax == 0.0f) continue;   // row never written by a decode step
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) { v[i] = __expf(v[i] - mx); sum += v[i]; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            int t = lane + 64 * i;
            if (t < kCtx) acc[wave][t] += v[i] * inv;
        }
    }
    __syncthreads();
    const float invn = 1.0f / (float)n_align;
    for (int t = threadIdx.x; t < kCtx; t += 256) out[row * kCtx + t] = ((acc[0][t] + acc[1][t]) + (acc[2][t] + acc[3][t])) * invn;
}
// ---- optional openai/whisper-style post-processing of the alignment heads before the head mean (timing.py find_alignment =
// transformers generation_whisper.py:341-349): softmax rows -> z-normalise every (head, frame) over the decoded token rows ->
// median filter of odd width along the frames (reflect padding) -> mean over heads.  SegmentSeeker applies none of it
// (Core/Text/SegmentSeeker.swift:195-237: whatever the CoreML model outputs is used as is), so this is an OPTION, default off.
__global__ __launch_bounds__(256) void align_softmax_kernel(const float* __restrict__ align, int n_align, float* __restrict__ prob, int* __restrict__ row_written) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x, j = blockIdx.y * 4 + wave;
    if (j >= n_align) return;
    const float* sp = align + ((size_t)row * n_align + j) * kCtx;
    float* dp = prob + ((size_t)row * n_align + j) * kCtx;
    float v[24];
    float mx = -INFINITY, amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int t = lane + 64 * i;
        v[i] = t < kCtx ? sp[t] : -INFINITY;
        mx = fmaxf(mx, v[i]);
        if (t < kCtx) amax = fmaxf(amax, fabsf(v[i]));
    }
    mx = wave_max(mx);
    amax = wave_max(amax);
    const bool written = amax != 0.0f;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 24; ++i) { v[i] = written ? __expf(v[i] - mx) : 0.0f; sum += v[i]; }
    sum = wave_sum(sum);
    const float inv = written ? 1.0f / sum : 0.0f;
#pragma unroll
    for (int i = 0; i < 24; ++i) { const int t = lane + 64 * i; if (t < kCtx) dp[t] = v[i] * inv; }
    if (j == 0 && lane == 0) row_written[row] = written ? 1 : 0;
}
__global__ __launch_bounds__(256) void align_stats_kernel(const float* __restrict__ prob, int n_align, const int* __restrict__ row_written,
                                                          float* __restrict__ mean, float* __restrict__ rstd) {
    const int j = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= kCtx) return;
    float s = 0.0f; int n = 0;
    for (int r = 0; r < kMaxTok; ++r) if (row_written[r]) { s += prob[((size_t)r * n_align + j) * kCtx + t]; ++n; }
    const float m = n ? s / (float)n : 0.0f;
    float q = 0.0f;
    for (int r = 0; r < kMaxTok; ++r) if (row_written[r]) { const float e = prob[((size_t)r * n_align + j) * kCtx + t] - m; q = fmaf(e, e, q); }
    const float sd = n ? sqrtf(q / (float)n) : 0.0f;        // torch.std(unbiased=False)
    mean[(size_t)j * kCtx + t] = m;
    rstd[(size_t)j * kCtx + t] = sd > 0.0f ? 1.0f / sd : 0.0f;
}
__global__ __launch_bounds__(256) void align_norm_mean_kernel(const float* __restrict__ prob, int n_align, const int* __restrict__ row_written,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, int znorm, int width,
                                                              float* __restrict__ out) {
    __shared__ float zr[kCtx];
    const int row = blockIdx.x, tid = threadIdx.x;
    constexpr int NT = (kCtx + 255) / 256;
    float acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = 0.0f;
    const bool written = row_written[row] != 0;
    const int p = width / 2;
    for (int j = 0; j < n_align && written; ++j) {
        __syncthreads();
        for (int t = tid; t < kCtx; t += 256) {
            float v = prob[((size_t)row * n_align + j) * kCtx + t];
            if (znorm) v = (v - mean[(size_t)j * kCtx + t]) * rstd[(size_t)j * kCtx + t];
            zr[t] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = tid + 256 * i;
            if (t >= kCtx) continue;
            float w[15];
            for (int k = 0; k < width; ++k) {
                int u = t - p + k;
                if (u < 0) u = -u;                          // reflect: x[1], x[2], ... left of the edge
                if (u >= kCtx) u = 2 * (kCtx - 1) - u;
                w[k] = zr[u];
            }
            for (int a_ = 1; a_ < width; ++a_) {           // insertion sort of <= 15 values
                const float key = w[a_];
                int b_ = a_ - 1;
                while (b_ >= 0 && w[b_] > key) { w[b_ + 1] = w[b_]; --b_; }
                w[b_ + 1] = key;
            }
            acc[i] += w[p];
        }
    }
    const float invn = 1.0f / (float)n_align;
#pragma unroll
    for (int i = 0; i < NT; ++i) { const int t = tid + 256 * i; if (t < kCtx) out[(size_t)row * kCtx + t] = acc[i] * invn; }
}
void launch_alignment_postprocess(const float* align, int n_align, float* prob_tmp, float* stat_tmp, int* row_written, int znorm, int median_width,
                                  float* out, hipStream_t st) {
    // one slot: align / prob_tmp [224][n_align][1500], stat_tmp [2][n_align][1500], out [224][1500]
    align_softmax_kernel<<<dim3(kMaxTok, (n_align + 3) / 4), 256, 0, st>>>(align, n_align, prob_tmp, row_written);
    if (znorm) align_stats_kernel<<<dim3((kCtx + 255) / 256, n_align), 256, 0, st>>>(prob_tmp, n_align, row_written, stat_tmp, stat_tmp + (size_t)n_align * kCtx);
    align_norm_mean_kernel<<<kMaxTok, 256, 0, st>>>(prob_tmp, n_align, row_written, stat_tmp, stat_tmp + (size_t)n_align * kCtx, znorm, median_width > 1 ? median_width : 1, out);
}

void launch_alignment_mean(const float* align, int batch, int n_align, float* out, hipStream_t st) {
    // `align` / `out` point at the first of `batch` consecutive slots
    alignment_mean_kernel<<<batch * kMaxTok, 256, 0, st>>>(align, n_align, out);
}

}  // namespace wh
