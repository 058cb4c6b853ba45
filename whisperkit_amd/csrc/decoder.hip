// Decoder token step for gfx950: replaces the per-token CoreML TextDecoder call of
// Sources/WhisperKit/Core/TextDecoder.swift:381-418 plus the host-side K4..K8 work around it
// (updateKVCache :218-270, updateAlignmentWeights :272-296, LogitsFilter.swift, TokenSampler.swift)
// and the loop bookkeeping of decodeText (:573-757).  Everything the loop needs lives in device
// memory (SeqState), so a step is a fixed kernel chain with no host round trip: the next input token,
// the cache position and the stop flag are read from / written to SeqState by the kernels themselves.
//
// This path is HBM-bandwidth and launch-latency bound (fp16 weight / KV streaming against <= 8 activation
// rows per batch tile): no MFMA.  Every grid-wide dependency of a decoder layer is one kernel boundary
// (cheaper on gfx950 than an in-kernel grid barrier, MI355X_MICROARCH.md "boundary" vs "barrier-xcd"):
//
//   gemv<QKV>     LN1(x) -> q (f32); k, v straight into the head-major self-attention cache at `pos`
//   self_attn     one workgroup per (head, slot): softmax(q K^T) V over <= 224 cached positions -> att
//   gemv<RESID>   x += W_o att + b_o
//   gemv<Q>       LN2(x) -> cross-attention query
//   cross_attn    workgroups per (key split, head, slot) over the 1500 cached cross K/V rows (flash-decoding
//                 split; the last-arriving split combines the partials in a fixed order -> att); alignment heads
//                 also store their raw score row (DecodingCache.alignmentWeights row tokenIndex + 1)
//   gemv<RESID>   x += W_co att + b_co
//   gemv<FC1>     LN3(x) -> GELU(fc1) (f16)
//   gemv<FC2>     x += W_2 h + b_2
//   ... per layer, then
//   gemv<LOGITS>  LN_f(x) . E^T  (tied embedding)
//   sampler       logits filters + greedy/top-k sample + decodeText state advance
//
// GEMV kernels: the LN prologue runs in registers (a row is spread over 256/BT lanes), the normalised
// activations of the batch tile sit in LDS as f32, weights stream as 16-byte loads with the next
// K-slice prefetched under the current one's FMAs (first slice issued before the prologue), a wave owns
// R rows x BT slots = 32 accumulators and finishes with a transposing butterfly (31 shuffles).
// All results are bit-deterministic (fixed summation orders, no float atomics).
#include <cstdlib>

#include "dec_shared.h"

namespace wh {

enum { MODE_QKV = 0, MODE_Q = 1, MODE_RESID = 2, MODE_FC1 = 3, MODE_FC2 = 4, MODE_LOGITS = 5 };

struct GemvArgs {
    int batch, d, n_head, N, K, rows_per_block, k_split, n_vocab;
    int layer;                 // MODE_QKV: layer index (0 -> token + position embedding prologue)
    const f16* W;              // [N][K]
    const float* bias;         // [N] or null
    const float *ln_g, *ln_b;  // prologue LayerNorm (QKV, Q, FC1, LOGITS)
    float* x;                  // residual stream [B][d]
    const float* ain;          // MODE_RESID input [B][d] f32 (attention output)
    // MODE_RESID, optional second problem in the same launch (workgroups >= nblk1): u = W2 [x ; att] + bias2, W2 = [N2][K2 = 4d]
    // fp16 hi|lo pairs of the folded cross-query matrices, out2 [B][N2] (finished to q by the cross-attention kernel)
    const f16* W2; const float* bias2; float* out2; int N2, K2, k_split2, rows_per_block2, nblk1;
    const f16* emb; const float* pos;   // layer-0 embedding
    float* q;                  // [B][d] f32 query out (QKV / Q)
    f16* self_k; f16* self_v;  // this layer's cache base [Bmax][H][224][64]
    f16* hbuf;                 // [B][4d]
    float* logits;             // [B][V]
    SeqState* seq;
    float* stats; const unsigned char* sup_mask; const SamplerCfg* cfg;   // MODE_LOGITS fused greedy sampler (stats != null)
    unsigned long long* dbg;   // optional timeline probe (WH_DBG=1): 8 timestamps per workgroup
};

#define DBG_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y == 0) a.dbg[(size_t)blockIdx.x * 8 + (i)] = ((i) == 6) ? (unsigned long long)clock64() : ((i) == 7 ? (unsigned long long)clock64() : (unsigned long long)wall_clock64()); } while (0)

// Sum 64 lanes of N values each; afterwards lane L holds the total of value index L >> (6 - log2 N)
// (N = 4, 8, 16, 32).  Costs N-1 + (6 - log2 N) shuffles instead of 6 N.  Template recursion keeps every
// register-array index a compile-time constant (a runtime-bounded loop here turns into v_cndmask chains).
template <int N, int OFF>
struct ReduceTranspose {
    static __device__ __forceinline__ float run(float* v, int lane) {
        const bool upper = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            float send = upper ? v[i] : v[i + N / 2];
            float keep = upper ? v[i + N / 2] : v[i];
            v[i] = keep + __shfl_xor(send, OFF, 64);
        }
        return ReduceTranspose<N / 2, OFF / 2>::run(v, lane);
    }
};
template <int OFF>
struct ReduceTranspose<1, OFF> {
    static __device__ __forceinline__ float run(float* v, int lane) {
        float r = v[0];
#pragma unroll
        for (int o = OFF; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        return r;
    }
};
template <int N>
__device__ __forceinline__ float reduce_transpose(float (&v)[N], int lane) { return ReduceTranspose<N, 32>::run(v, lane); }

// sum over the TPR consecutive threads that share one activation row (TPR a power of two; rows never straddle a
// wave when TPR <= 64, otherwise a row is TPR/64 whole waves and the wave totals meet in LDS)
template <int TPR, int NT>
__device__ __forceinline__ float row_sum(float v, float* red) {
    if constexpr (TPR <= 64) {
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    } else {
        constexpr int WPR = TPR / 64;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const int w0 = (wave / WPR) * WPR;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < WPR; ++i) s += red[w0 + i];
        return s;
    }
}

// ---------------------------------------------------------------------------------------------- fused greedy sampler, part 1
// Filter rules of one sampling step as scalars (restating LogitsFilter.swift): r[0] SuppressBlank active,
// r[1] TimestampRules active, [r2, r3) and [r4, r5) id ranges masked by the timestamp rules.
__device__ __forceinline__ void compute_filter_rules(const SamplerCfg& cfg, const SeqState* sq, int n_tok, int V, int* r) {
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
                for (int i = n_tok - 1; i >= sb; --i)
                    if (sq->tokens[i] >= tb) { lastTimestamp = sq->tokens[i]; break; }
                if (lastTimestamp >= 0) {
                    int tl = (lastTs && !penTs) ? lastTimestamp : lastTimestamp + 1;
                    r2lo = tb; r2hi = tl;
                }
            }
        }
    }
    r[0] = blank; r[1] = ts_active; r[2] = r1lo; r[3] = r1hi; r[4] = r2lo; r[5] = r2hi;
}

__device__ __forceinline__ void wave_argmax(float& v, int& idx) {   // ties -> smallest index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, 64);
        int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// What the statistics epilogue of the logits kernel needs from global memory, fetched at kernel ENTRY (scalar loads keyed
// by the wave-uniform slot index, one mask byte per lane) so that the epilogue itself is pure register / LDS work.
struct StatPre {
    int live[2];
    int rules[2][6];
    int masked;      // SuppressTokensFilter byte of row n_begin + lane (rows past n_end count as masked)
};
template <int BT, typename Args>
__device__ __forceinline__ void logits_stats_prefetch(const Args& a, int b0, int n_begin, int n_end, StatPre& pre) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int b = wave + 4 * s, gb = b0 + b;
        pre.live[s] = 0;
        if (b < BT && gb < a.batch) {
            const SeqState* sq = a.seq + gb;
            pre.live[s] = slot_live(sq);
#pragma unroll
            for (int i = 0; i < 6; ++i) pre.rules[s][i] = sq->f_rules[i];
        }
    }
    const int n = n_begin + lane;
    pre.masked = n < n_end ? (int)a.sup_mask[n] : 1;
}

// Epilogue of the logits kernel when every live slot samples greedily: apply the index-predicate filters to this
// workgroup's <= 64 logits per slot and reduce them to (max, argmax, sum exp) separately for text ids (< timeTokenBegin)
// and timestamp ids; sampler_final_kernel merges the per-workgroup records.  One wave per slot, lane = row.
template <int BT, typename Args>
__device__ __forceinline__ void logits_block_stats(const Args& a, const float* lt, const StatPre& pre, int b0, int n_begin, int n_end,
                                                   int tb, int ws_tok, int eot_tok, int no_ts_tok) {
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int b = wave + 4 * s;
        if (b >= BT || !pre.live[s]) continue;
        const int gb = b0 + b;
        const int n = n_begin + lane;
        float v = -INFINITY;
        if (n < n_end) {
            v = lt[b * 64 + lane];
            const int blank = pre.rules[s][0], ts_active = pre.rules[s][1];
            bool masked = pre.masked != 0;                                                       // SuppressTokensFilter
            masked |= blank && (n == ws_tok || n == eot_tok);                                    // SuppressBlankFilter
            masked |= ts_active && (n == no_ts_tok || (n >= pre.rules[s][2] && n < pre.rules[s][3]) ||
                                    (n >= pre.rules[s][4] && n < pre.rules[s][5]));             // TimestampRulesFilter
            if (masked) v = -INFINITY;
        }
        const bool is_ts = n >= tb;
        float mt = is_ts ? -INFINITY : v, ms = is_ts ? v : -INFINITY;
        int it = n, is = n;
        const float vt = mt, vs = ms;
        wave_argmax(mt, it);
        wave_argmax(ms, is);
        float st = (vt == -INFINITY) ? 0.0f : __expf(vt - mt);
        float ss = (vs == -INFINITY) ? 0.0f : __expf(vs - ms);
        st = wave_sum(st);
        ss = wave_sum(ss);
        if (lane == 0) {
            float* o = a.stats + ((size_t)gb * kStatBlocks + blockIdx.x) * 8;
            *reinterpret_cast<float4*>(o) = float4{mt, st, __int_as_float(it), ms};
            *reinterpret_cast<float2*>(o + 4) = float2{ss, __int_as_float(is)};
        }
    }
}

// Workgroup = 4 waves arranged as KS K-splits x RG = 4 / KS row groups.  A wave owns R weight rows and the K range
// [ks * K / KS, (ks + 1) * K / KS) of them (<= 1536 columns = 3 slices of 512 = 64 lanes x 8 halves), and issues ALL of its
// weight loads before touching them: these matrices are a few MB spread over 256 CUs, so the only way to reach the
// HBM rate is to have every byte of the matrix in flight at once (the first row group's loads are issued before the
// prologue; in multi-pass launches - the logits - the next pass is prefetched under the current one's FMAs).
template <int MODE, int BT, int R>
__global__ __launch_bounds__(256, 2) void dec_gemv_kernel(const GemvArgs a) {
    constexpr int NT = 256, NW = 4, KI = 3;
    constexpr int NV = R * BT;                              // accumulators per lane (<= 32)
    constexpr bool kPrefetchNextPass = (MODE == MODE_LOGITS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);   // [BT][K] f32 (+ gamma, beta in the LayerNorm modes); unused by MODE_FC2
    __shared__ float red[NW];
    __shared__ float kred[NW][32];
    __shared__ float lt[MODE == MODE_LOGITS ? BT * 64 : 1];   // this workgroup's logits (<= 64 rows) for the fused sampler statistics
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.y * BT;
    const int d = a.d;
    // MODE_RESID can carry a second, independent GEMV in the same launch (the folded cross-attention query): workgroups
    // >= nblk1 work on it.  The choice is workgroup-uniform; everything below reads the selected problem.
    const bool second = (MODE == MODE_RESID) && a.W2 != nullptr && (int)blockIdx.x >= a.nblk1;
    const int K = second ? a.K2 : a.K, N = second ? a.N2 : a.N;
    const f16* const Wp = second ? a.W2 : a.W;
    const float* const biasp = second ? a.bias2 : a.bias;
    const int rows_per_block = second ? a.rows_per_block2 : a.rows_per_block;
    const int bx = second ? (int)blockIdx.x - a.nblk1 : (int)blockIdx.x;
    const int KS = second ? a.k_split2 : a.k_split, RG = NW / KS;
    const int ks = wave % KS, rg = wave / KS;
    const int KC = K / KS, kbase = ks * KC;
    const int ldx = second ? 2 * d : K;                      // LDS row stride of the staged activations
    const int xoff = second ? (ks >> 1) * d : kbase;          // [x ; att]: K quarters 0,1 read x, quarters 2,3 read att
    // Slot liveness is LOADED here but only LOOKED AT after the activation and weight loads have been issued (LIVE_CHECK):
    // a test right away would put one more dependent L2 round trip in front of every kernel of the chain.
    int s_act[BT], s_done[BT];
#pragma unroll
    for (int i = 0; i < BT; ++i) {
        const bool in = b0 + i < a.batch;
        s_act[i] = in ? a.seq[b0 + i].active : 0;
        s_done[i] = in ? a.seq[b0 + i].done : 1;
    }
#define LIVE_CHECK()                                                          \
    do {                                                                      \
        bool any_live_ = false;                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < BT; ++i_) any_live_ |= (s_act[i_] && !s_done[i_]); \
        if (!any_live_) return;                                               \
    } while (0)
    DBG_STAMP(6);

    const int n_begin = bx * rows_per_block;
    const int n_end = min(N, n_begin + rows_per_block);
    const int rows_per_pass = RG * R;
    const int n_pass = (n_end - n_begin + rows_per_pass - 1) / rows_per_pass;

    uint4 cur[R][KI];
    auto load_group = [&](uint4 (&dst)[R][KI], int n0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f16* wr = Wp + (size_t)min(n0 + r, N - 1) * K + kbase + lane * 8;
#pragma unroll
            for (int i = 0; i < KI; ++i)
                dst[r][i] = (lane * 8 + 512 * i < KC) ? *reinterpret_cast<const uint4*>(wr + 512 * i) : uint4{0, 0, 0, 0};
        }
    };

    // ------------------------------------------------------------------ prologue -> LDS
    // Issue order matters: memory returns are in order per wave, so the (small, L2-resident) activation loads go out
    // first and the weight stream (HBM) second - the LayerNorm then runs while the weights are still in flight.
    // MODE_FC2: each wave needs only ITS K range of the hidden activations (f16, written by the fc1 kernel): they go straight
    // from L2 into registers - no LDS image (80 KB at d = 1280 would cap the residency at one workgroup per CU), no barrier
    uint4 hreg[MODE == MODE_FC2 ? BT : 1][MODE == MODE_FC2 ? KI : 1];
    if constexpr (MODE == MODE_FC2) {
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int i = 0; i < KI; ++i)
                hreg[b][i] = (b0 + b < a.batch && lane * 8 + 512 * i < KC)
                                 ? *reinterpret_cast<const uint4*>(a.hbuf + (size_t)(b0 + b) * K + kbase + lane * 8 + 512 * i) : uint4{0, 0, 0, 0};
        load_group(cur, n_begin + rg * R);
        DBG_STAMP(1);
        LIVE_CHECK();
    } else if constexpr (MODE == MODE_RESID) {
        constexpr int AV = (BT * 320 + NT - 1) / NT;    // float4 per thread, d <= 1280
        float4 areg[AV], xreg[AV];
        const int per_row = d / 4;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int idx = tid + i * NT;
            const int b = idx / per_row, c = idx - b * per_row;
            const bool ok = idx < BT * per_row && b0 + b < a.batch;
            areg[i] = ok ? reinterpret_cast<const float4*>(a.ain + (size_t)(b0 + b) * d)[c] : float4{0, 0, 0, 0};
            xreg[i] = (ok && second) ? reinterpret_cast<const float4*>(a.x + (size_t)(b0 + b) * d)[c] : float4{0, 0, 0, 0};
        }
        load_group(cur, n_begin + rg * R);
        DBG_STAMP(1);
        LIVE_CHECK();
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int idx = tid + i * NT;
            if (idx < BT * per_row) {
                const int b = idx / per_row, c = idx - b * per_row;
                if (second) {       // row layout [x | att]
                    reinterpret_cast<float4*>(xs + (size_t)b * ldx)[c] = xreg[i];
                    reinterpret_cast<float4*>(xs + (size_t)b * ldx + d)[c] = areg[i];
                } else {
                    reinterpret_cast<float4*>(xs + (size_t)b * ldx)[c] = areg[i];
                }
            }
        }
    } else {
        // LayerNorm in registers: row = tid / 32, the row's d/4 float4 are dealt round-robin to its 32 threads.  The split is
        // the same for every batch tile size (threads beyond BT rows idle) so that the statistics - hence every logit - of a
        // slot are bit-identical whether it is decoded alone or in a batch.
        constexpr int TPR = 32;
        constexpr int MAXV = (320 + TPR - 1) / TPR;     // d <= 1280
        const int row = tid / TPR, li = tid - row * TPR;
        const int gb = b0 + row;
        const bool rok = row < BT && gb < a.batch;
        const int nv4 = d / 4;
        float4 v[MAXV];
        // gamma / beta: issued first (they depend on nothing), parked in LDS behind the activations, read back after the
        // row statistics - a dependent global load after the statistics would put a second L2 round trip on the critical path
        float* gb_l = xs + (size_t)BT * K;      // [2][d]
        constexpr int GV = (2 * 320 + NT - 1) / NT;
        float4 g_reg[GV];
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int idx = tid + i * NT;       // float4 index into gamma (first d/4) then beta
            g_reg[i] = idx < 2 * nv4 ? *reinterpret_cast<const float4*>((idx < nv4 ? a.ln_g : a.ln_b - d) + (size_t)idx * 4) : float4{0, 0, 0, 0};
        }
        const bool embed = (MODE == MODE_QKV) && a.layer == 0;
        if (embed) {
            int tok = 0, pos = 0;
            if (rok) {
                tok = min(max(a.seq[gb].next_token, 0), a.n_vocab - 1);
                pos = min(max(a.seq[gb].token_index, 0), kMaxTok - 1);
            }
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c4 = li + i * TPR;
                float4 t = float4{0, 0, 0, 0};
                if (rok && c4 < nv4) {
                    f16x4 e = *reinterpret_cast<const f16x4*>(a.emb + (size_t)tok * d + c4 * 4);
                    float4 p = *reinterpret_cast<const float4*>(a.pos + (size_t)pos * d + c4 * 4);
                    t = float4{(float)e[0] + p.x, (float)e[1] + p.y, (float)e[2] + p.z, (float)e[3] + p.w};
                    if (blockIdx.x == 0) *reinterpret_cast<float4*>(a.x + (size_t)gb * d + c4 * 4) = t;
                }
                v[i] = t;
            }
        } else {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c4 = li + i * TPR;
                v[i] = (rok && c4 < nv4) ? *reinterpret_cast<const float4*>(a.x + (size_t)gb * d + c4 * 4) : float4{0, 0, 0, 0};
            }
        }
        load_group(cur, n_begin + rg * R);
        DBG_STAMP(1);
        LIVE_CHECK();
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int idx = tid + i * NT;
            if (idx < 2 * nv4) reinterpret_cast<float4*>(gb_l)[idx] = g_reg[i];
        }
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = row_sum<TPR, NT>(s, red) / (float)d;
        float qv = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (li + i * TPR < nv4) {
                float ax = v[i].x - mean, ay = v[i].y - mean, az = v[i].z - mean, aw = v[i].w - mean;
                qv += (ax * ax + ay * ay) + (az * az + aw * aw);
            }
        }
        const float rstd = rsqrtf(row_sum<TPR, NT>(qv, red) / (float)d + 1e-5f);
        __syncthreads();    // gamma / beta visible
        DBG_STAMP(0);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = li + i * TPR;
            if (c4 < nv4) {
                const float4 g = reinterpret_cast<const float4*>(gb_l)[c4];
                const float4 be = reinterpret_cast<const float4*>(gb_l)[nv4 + c4];
                float4 o;
                o.x = (v[i].x - mean) * rstd * g.x + be.x; o.y = (v[i].y - mean) * rstd * g.y + be.y;
                o.z = (v[i].z - mean) * rstd * g.z + be.z; o.w = (v[i].w - mean) * rstd * g.w + be.w;
                if (!rok) o = float4{0, 0, 0, 0};
                if (row < BT) *reinterpret_cast<float4*>(xs + (size_t)row * d + c4 * 4) = o;
            }
        }
    }
    __syncthreads();
    DBG_STAMP(2);

    StatPre pre;
    int st_tb = 0, st_ws = 0, st_eot = 0, st_nots = 0;
    if constexpr (MODE == MODE_LOGITS) {
        if (a.stats) {
            logits_stats_prefetch<BT>(a, b0, n_begin, n_end, pre);
            st_tb = a.cfg->time_token_begin; st_ws = a.cfg->whitespace_token; st_eot = a.cfg->end_token; st_nots = a.cfg->no_timestamps_token;
        }
    }
    // ------------------------------------------------------------------ GEMV passes
#pragma unroll 1
    for (int p = 0; p < n_pass; ++p) {
        const int n0 = n_begin + (p * RG + rg) * R;
        uint4 nxt[kPrefetchNextPass ? R : 1][kPrefetchNextPass ? KI : 1];
        if constexpr (kPrefetchNextPass) {
            if (p + 1 < n_pass) load_group(nxt, n_begin + ((p + 1) * RG + rg) * R);
        }
        // epilogue operands of this lane's output (bias, residual value, cache position): fetched now, used after the FMAs
        constexpr int SHp = (NV == 32) ? 1 : (NV == 16) ? 2 : (NV == 8) ? 3 : (NV == 4) ? 4 : (NV == 2) ? 5 : 6;
        const int idx_p = lane >> SHp;
        const int n_l = n0 + idx_p / BT, gb_l = b0 + idx_p % BT;
        const bool out_l = ks == 0 && (lane & ((1 << SHp) - 1)) == 0 && n_l < n_end && gb_l < a.batch;
        float bias_l = 0.0f, xold_l = 0.0f;
        int pos_l = 0;
        if (out_l) {
            if (biasp) bias_l = biasp[n_l];
            if constexpr (MODE == MODE_RESID || MODE == MODE_FC2) { if (!second) xold_l = a.x[(size_t)gb_l * d + n_l]; }
            if constexpr (MODE == MODE_QKV) pos_l = a.seq[gb_l].token_index;
        }
        float acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int kl = lane * 8 + 512 * i;     // column inside this wave's K range
            if (kl < KC) {
                const int k = kbase + kl;
                // activations in chunks of <= 4 slots: bounds the live x registers (accumulation order per output unchanged)
                constexpr int BC = BT < 4 ? BT : 4;
#pragma unroll
                for (int bc = 0; bc < BT; bc += BC) {
                    float xk[BC][8];
#pragma unroll
                    for (int b = 0; b < BC; ++b) {
                        if constexpr (MODE == MODE_FC2) {
                            f16x8 hv = *reinterpret_cast<f16x8*>(&hreg[bc + b][i]);
#pragma unroll
                            for (int j = 0; j < 8; ++j) xk[b][j] = (float)hv[j];
                        } else {
                            float4 x0 = *reinterpret_cast<const float4*>(xs + (bc + b) * ldx + xoff + kl);
                            float4 x1 = *reinterpret_cast<const float4*>(xs + (bc + b) * ldx + xoff + kl + 4);
                            xk[b][0] = x0.x; xk[b][1] = x0.y; xk[b][2] = x0.z; xk[b][3] = x0.w;
                            xk[b][4] = x1.x; xk[b][5] = x1.y; xk[b][6] = x1.z; xk[b][7] = x1.w;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f16x8 wh8 = *reinterpret_cast<f16x8*>(&cur[r][i]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float wf = (float)wh8[j];
#pragma unroll
                            for (int b = 0; b < BC; ++b) acc[r * BT + bc + b] = fmaf(wf, xk[b][j], acc[r * BT + bc + b]);
                        }
                    }
                }
            }
        }
        DBG_STAMP(3);
        float tot = reduce_transpose<NV>(acc, lane);
        constexpr int SH = (NV == 32) ? 1 : (NV == 16) ? 2 : (NV == 8) ? 3 : (NV == 4) ? 4 : (NV == 2) ? 5 : 6;   // 6 - log2(NV)
        const bool holder = (lane & ((1 << SH) - 1)) == 0;
        const int idx = lane >> SH;
        if (KS > 1) {   // K-split: the partial sums of a row group meet in LDS and are added in split order by split 0
            if (holder) kred[wave][idx] = tot;
            __syncthreads();
            if (ks == 0 && holder) {
                tot = kred[wave][idx];
                for (int j = 1; j < KS; ++j) tot += kred[wave + j][idx];
            }
        }
        if (ks == 0 && holder) {
            const int r = idx / BT, b = idx - r * BT;
            const int n = n0 + r, gb = b0 + b;
            if (n < n_end && gb < a.batch) {
                float v = tot + bias_l;
                if constexpr (MODE == MODE_QKV) {
                    int pos = min(max(pos_l, 0), kMaxTok - 1);
                    if (n < d) a.q[(size_t)gb * d + n] = v;
                    else {
                        int c = n - d;
                        f16* dst = a.self_k;
                        if (c >= d) { c -= d; dst = a.self_v; }
                        dst[(((size_t)gb * a.n_head + (c >> 6)) * kMaxTok + pos) * kHeadDim + (c & 63)] = (f16)v;
                    }
                } else if constexpr (MODE == MODE_Q) {
                    a.q[(size_t)gb * d + n] = v;
                } else if constexpr (MODE == MODE_FC1) {
                    a.hbuf[(size_t)gb * N + n] = (f16)gelu_erf(v);
                } else if constexpr (MODE == MODE_RESID || MODE == MODE_FC2) {
                    if (second) a.out2[(size_t)gb * N + n] = v;
                    else a.x[(size_t)gb * d + n] = xold_l + v;
                } else {
                    a.logits[(size_t)gb * N + n] = v;
                    if (a.stats) lt[b * 64 + (n - n_begin)] = v;
                }
            }
        }
        DBG_STAMP(4);
        if (KS > 1 && p + 1 < n_pass) __syncthreads();   // kred is reused by the next pass
        if constexpr (kPrefetchNextPass) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < KI; ++i) cur[r][i] = nxt[r][i];
        } else if (p + 1 < n_pass) {
            load_group(cur, n_begin + ((p + 1) * RG + rg) * R);
        }
    }
    if constexpr (MODE == MODE_LOGITS) {
        if (a.stats) logits_block_stats<BT>(a, lt, pre, b0, n_begin, n_end, st_tb, st_ws, st_eot, st_nots);
    }
    DBG_STAMP(5); DBG_STAMP(7);
}

// ---------------------------------------------------------------------------------------------- attention
struct AttnArgs {
    int batch, d, n_head, layer, n_layer, n_split;
    const float* q;          // [B][d]
    const f16* self_k; const f16* self_v;     // layer base [Bmax][H][224][64]
    const f16* cross_k; const f16* cross_v;   // layer base [Bmax][H][1500][64]
    float* att;              // [B][d] attention output (before the out projection); GEMV path
    f16 *att_hi, *att_lo;    // MFMA path (decoder32.hip): the same values as an f16 hi | lo pair in B-fragment plane order
    float* part;             // [B][H][n_split][kPartStride]: (m, l, o[64]) of every key split, one 128-byte-aligned slot each
    int* ticket;             // [B][H] arrival counters (zero between launches)
    float* align; const int* align_slot; int n_align;   // [B][224][n_align][1500] raw score rows of the alignment heads
    SeqState* seq;
    // folded cross query (xq != null): a.q holds u; q = (u - mean(x') r) * rstd(x') + c is finished here, x' = xq [B][d]
    const float* xq; const float* qr; const float* qc;
    int no_fence;
    unsigned long long* dbg;   // optional timeline probe (WH_DBG=1)
};
#define ATT_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 4096 * 8 + (i)] = (unsigned long long)wall_clock64(); } while (0)

__device__ __forceinline__ void store_att(const AttnArgs& a, int b, int n, float v) {
    if (a.att_hi) {
        f16 hi, lo;
        split_hilo(v, hi, lo);
        const size_t o = plane_index(b, n, a.d);
        a.att_hi[o] = hi;
        a.att_lo[o] = lo;
    } else a.att[(size_t)b * a.d + n] = v;
}

// One query against keys [t0, t0 + n) of a head-major K/V block (rows of 64 halves).  Thread layout: 8 lanes per
// key (16 bytes = 8 channels each), 32 keys per pass, PASSES passes; all K and V rows of the block are in flight
// before the first use.  Returns this block's softmax statistics (m, l) and leaves the unnormalised output
// o[64] = sum_t exp(s_t - m) V[t] in o_out (LDS, valid for tid < 64).  raw_scores (optional, global) gets s_t.
template <int PASSES, typename GetN, typename QFix>
__device__ __forceinline__ bool attend_block(const float* __restrict__ qg, const f16* __restrict__ kb, const f16* __restrict__ vb, int n_load,
                                             GetN get_n, QFix qfix, float* const* raw_pp, float* red /* [16] */, float* osum /* [4][64] */,
                                             float* o_out /* [64] */, float* m_out, float* l_out, unsigned long long* stamp = nullptr) {
    // n_load rows are FETCHED right away; how many of them count (n = get_n(), < 0: slot not live) is only looked at
    // afterwards, so the slot-state loads and the K/V stream share one memory round trip instead of two.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = tid & 7, kg = tid >> 3;
    uint4 kreg[PASSES], vreg[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = kg + 32 * i;
        kreg[i] = key < n_load ? *reinterpret_cast<const uint4*>(kb + (size_t)key * kHeadDim + part * 8) : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int key = kg + 32 * i;
        vreg[i] = key < n_load ? *reinterpret_cast<const uint4*>(vb + (size_t)key * kHeadDim + part * 8) : uint4{0, 0, 0, 0};
    }
    float qv[8];
    {
        float4 q0 = *reinterpret_cast<const float4*>(qg + part * 8);
        float4 q1 = *reinterpret_cast<const float4*>(qg + part * 8 + 4);
        qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    }
    const int n = get_n();
    if (n < 0) return false;            // workgroup-uniform
    qfix(qv, part);                     // identity, or the LayerNorm-folded cross query finish
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

__global__ __launch_bounds__(256) void dec_self_attn_kernel(const AttnArgs a) {
    __shared__ float red[16], osum[256], o_l[64];
    const int h = blockIdx.x, b = blockIdx.y;
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;     // looked at after the K/V loads are issued
    const int d = a.d;
    const size_t base = ((size_t)b * a.n_head + h) * kMaxTok * kHeadDim;
    float m, l;
    float* raw = nullptr;
    auto get_n = [&]() { return (s_act && !s_done) ? min(max(s_ti, 0), kMaxTok - 1) + 1 : -1; };
    auto qfix = [](float (&)[8], int) {};
    if (!attend_block<7>(a.q + (size_t)b * d + h * kHeadDim, a.self_k + base, a.self_v + base, kMaxTok, get_n, qfix, &raw, red, osum, o_l, &m, &l))
        return;
    if (threadIdx.x < 64) store_att(a, b, h * kHeadDim + threadIdx.x, o_l[threadIdx.x] / l);
}

template <int PASSES>
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const AttnArgs a) {
    constexpr int KPB = PASSES * 32;
    __shared__ float red[16], osum[256], o_l[64];
    __shared__ int last_flag;
    const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;     // looked at after the K/V loads are issued
    const int d = a.d, S = a.n_split;
    const int t0 = sp * KPB, n = min(KPB, kCtx - t0);
    const size_t base = (((size_t)b * a.n_head + h) * kCtx + t0) * kHeadDim;
    int slot = -1;
    if (a.align) slot = a.align_slot[a.layer * a.n_head + h];
    float m, l;
    ATT_STAMP(0);
    unsigned long long* stamp = a.dbg ? a.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 4096 * 8 : nullptr;
    // alignment-head row: DecodingCache.alignmentWeights row tokenIndex + 1 (TextDecoder.swift:272-296), raw scores here,
    // softmax + head mean in alignment_mean_kernel
    float* raw = nullptr;
    // folded cross query: this slot's residual row x' (for its LayerNorm statistics) and the r / c constants of the lane's 8
    // query channels are requested now, ahead of the K/V stream issued inside attend_block
    const bool fq = a.xq != nullptr;
    const int nv4 = d >> 2, tidq = threadIdx.x;
    float4 xa = float4{0, 0, 0, 0}, xb = float4{0, 0, 0, 0}, r0 = xa, r1 = xa, c0 = xa, c1 = xa;
    if (fq) {
        const float4* xr = reinterpret_cast<const float4*>(a.xq + (size_t)b * d);
        if (tidq < nv4) xa = xr[tidq];
        if (tidq + 256 < nv4) xb = xr[tidq + 256];
        const int qo = h * kHeadDim + (tidq & 7) * 8;
        r0 = *reinterpret_cast<const float4*>(a.qr + qo); r1 = *reinterpret_cast<const float4*>(a.qr + qo + 4);
        c0 = *reinterpret_cast<const float4*>(a.qc + qo); c1 = *reinterpret_cast<const float4*>(a.qc + qo + 4);
    }
    float q_mean = 0.0f, q_rstd = 1.0f;
    auto get_n = [&]() {
        if (!(s_act && !s_done)) return -1;
        const int pos = min(max(s_ti, 0), kMaxTok - 1);
        if (slot >= 0 && pos + 1 < kMaxTok) raw = a.align + (((size_t)b * kMaxTok + pos + 1) * a.n_align + slot) * kCtx + t0;
        if (fq) {       // LayerNorm statistics of x'[b] (two-pass, fixed order: the same in every workgroup of the slot)
            const int lane = tidq & 63, wave = tidq >> 6;
            float sm = (xa.x + xa.y) + (xa.z + xa.w) + ((xb.x + xb.y) + (xb.z + xb.w));
            sm = wave_sum(sm);
            if (lane == 0) red[8 + wave] = sm;
            __syncthreads();
            q_mean = ((red[8] + red[9]) + (red[10] + red[11])) / (float)d;
            float qv2 = 0.0f;
            if (tidq < nv4) { float e0 = xa.x - q_mean, e1 = xa.y - q_mean, e2 = xa.z - q_mean, e3 = xa.w - q_mean; qv2 += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3); }
            if (tidq + 256 < nv4) { float e0 = xb.x - q_mean, e1 = xb.y - q_mean, e2 = xb.z - q_mean, e3 = xb.w - q_mean; qv2 += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3); }
            qv2 = wave_sum(qv2);
            if (lane == 0) red[12 + wave] = qv2;
            __syncthreads();
            q_rstd = rsqrtf(((red[12] + red[13]) + (red[14] + red[15])) / (float)d + 1e-5f);
        }
        return n;
    };
    auto qfix = [&](float (&qv)[8], int) {
        if (!fq) return;
        const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = (qv[j] - q_mean * rr[j]) * q_rstd + cc[j];
    };
    if (!attend_block<PASSES>(a.q + (size_t)b * d + h * kHeadDim, a.cross_k + base, a.cross_v + base, n, get_n, qfix, &raw, red, osum, o_l, &m, &l, stamp))
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
    ATT_STAMP(5);
}

// ---------------------------------------------------------------------------------------------- sampler
constexpr int SAMP_T = 1024;
constexpr int SAMP_E = 51;   // ceil(51866 / 1024)

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
__device__ __forceinline__ void advance_decode_state(const SamplerCfg& cfg, SeqState* sq, int tok, float lp, int n_tok) {
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
    compute_filter_rules(cfg, sq, nt, cfg.n_vocab, sq->f_rules);
}

// MODE bit 0: apply filters; bit 1: sample; bit 2: advance decodeText state; bit 3: write filtered logits back
template <int DO_FILTER, int DO_SAMPLE, int DO_ADVANCE, int WRITE_BACK>
__global__ __launch_bounds__(SAMP_T) void sampler_kernel(const SamplerCfg* __restrict__ cfgp, const int* __restrict__ suppress,
                                                        SeqState* __restrict__ seqs, float* __restrict__ logits_all,
                                                        int counter_override, int* __restrict__ token_out, float* __restrict__ logprob_out) {
    __shared__ BlockRed br;
    __shared__ int sh_i[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    SeqState* sq = seqs + b;
    if (DO_ADVANCE && !slot_live(sq)) return;
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
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < SAMP_E; ++e) {
            int n = tid + SAMP_T * e;
            if (n < V && n >= tb && x[e] != -INFINITY) s += expf(x[e] - m_ts);
        }
        s = block_sum(s, &br);
        bool cond = (m_ts != -INFINITY) && (m_ts + logf(s) > m_text);
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
    __shared__ int si[4][2];
    __shared__ SeqState sq_l;     // the slot's whole decode state: thread 0's bookkeeping (token history scans, appends) runs on
                                  // this LDS copy instead of a chain of dependent global round trips
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SeqState* sq = seqs + b;
    if (!slot_live(sq)) return;
    constexpr int kWords = sizeof(SeqState) / 4;
    for (int i = tid; i < kWords; i += 256) reinterpret_cast<int*>(&sq_l)[i] = reinterpret_cast<const int*>(sq)[i];
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
    if (lane == 0) { sm[wave][0] = t.m; sm[wave][1] = t.s; si[wave][0] = t.i; sm[wave][2] = u.m; sm[wave][3] = u.s; si[wave][1] = u.i; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { stat_merge(t, sm[w][0], sm[w][1], si[w][0]); stat_merge(u, sm[w][2], sm[w][3], si[w][1]); }
        const SamplerCfg cfg = *cfgp;
        const bool ts_active = sq_l.f_rules[1] != 0;
        int tok; float lp;
        const bool cond = ts_active && u.m != -INFINITY && (u.m + logf(u.s) > t.m);
        if (cond || t.m == -INFINITY) {          // text ids masked: the candidates are the timestamp ids
            tok = u.i; lp = -logf(u.s);
        } else {
            SoftStat g = t;
            stat_merge(g, u.m, u.s, u.i);        // equal maxima: the text id (smaller index) wins, like a first-maximum argmax
            tok = g.i; lp = -logf(g.s);
        }
        advance_decode_state(cfg, &sq_l, tok, lp, sq_l.n_tokens);
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
static unsigned long long* g_dbg_buf = nullptr;   // [KK_COUNT][4096][8]
static thread_local int g_dbg_kind = 0;
unsigned long long* debug_buffer() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("WH_DBG"); on = (e && e[0] == '1'); if (on) { (void)hipMalloc((void**)&g_dbg_buf, (size_t)KK_COUNT * 4096 * 8 * 8); (void)hipMemset(g_dbg_buf, 0, (size_t)KK_COUNT * 4096 * 8 * 8); } }
    return g_dbg_buf;
}

template <int MODE, int BT, int R>
static void launch_gemv_r(GemvArgs a, int passes, hipStream_t st) {
    a.dbg = debug_buffer() ? debug_buffer() + (size_t)g_dbg_kind * 4096 * 8 : nullptr;
    a.rows_per_block = (4 / a.k_split) * R * passes;
    dim3 g((a.N + a.rows_per_block - 1) / a.rows_per_block, (a.batch + BT - 1) / BT);
    const bool ln_mode = MODE == MODE_QKV || MODE == MODE_Q || MODE == MODE_FC1 || MODE == MODE_LOGITS;
    const size_t smem = (MODE == MODE_FC2) ? 0 : (size_t)(BT * a.K + (ln_mode ? 2 * a.d : 0)) * sizeof(float);
    if (smem > 64 * 1024) {   // above the default dynamic-LDS limit: raise it once per instantiation (160 KB per CU on gfx950)
        static PerDeviceOnce raised;
        raised.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_gemv_kernel<MODE, BT, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); });   // + static LDS <= 160 KB
    }
    dec_gemv_kernel<MODE, BT, R><<<g, 256, smem, st>>>(a);
}

template <int MODE, int BT>
static void launch_gemv_bt(GemvArgs a, hipStream_t st) {
    // K-split so that a wave's K range fits its 3 x 512-column register slices; rows per wave: the largest of 4 / 2 / 1 that
    // still yields >= 256 workgroups (one per CU); the logits matrix runs 4 row passes per workgroup with prefetch
    a.k_split = a.K <= 1536 ? 1 : a.K <= 3072 ? 2 : 4;
    const int rg = 4 / a.k_split;
    const int passes = (MODE == MODE_LOGITS) ? 4 : 1;
    static const int minblk = env_int("WH_GEMV_MINBLK", 256);   // tuning knob
    if ((a.N + rg * 4 - 1) / (rg * 4) >= minblk) launch_gemv_r<MODE, BT, 4>(a, passes, st);
    else if ((a.N + rg * 2 - 1) / (rg * 2) >= minblk) launch_gemv_r<MODE, BT, 2>(a, passes, st);
    else launch_gemv_r<MODE, BT, 1>(a, passes, st);
}

// out projection + folded cross query in one launch (MODE_RESID with a second problem): batch tiles of 4 slots so that the
// [x ; att] image of the second problem is 2 * 4 * d floats of LDS, R = 4 rows per wave for both problems
static void launch_gemv_resid_cq(GemvArgs a, hipStream_t st) {
    a.k_split = 1;
    a.rows_per_block = 4 * 4;                    // 4 row groups x R
    a.nblk1 = (a.N + a.rows_per_block - 1) / a.rows_per_block;
    a.k_split2 = 4;                              // the four K quarters Wq'_hi | Wq'_lo | M_hi | M_lo, one wave each
    a.rows_per_block2 = 4;                       // 1 row group x R
    const int nblk2 = (a.N2 + a.rows_per_block2 - 1) / a.rows_per_block2;
    a.dbg = debug_buffer() ? debug_buffer() + (size_t)g_dbg_kind * 4096 * 8 : nullptr;
    const size_t smem4 = (size_t)4 * 2 * a.d * sizeof(float), smem1 = (size_t)2 * a.d * sizeof(float);
    if (a.batch >= 2) dec_gemv_kernel<MODE_RESID, 4, 4><<<dim3(a.nblk1 + nblk2, (a.batch + 3) / 4), 256, smem4, st>>>(a);
    else dec_gemv_kernel<MODE_RESID, 1, 4><<<dim3(a.nblk1 + nblk2, 1), 256, smem1, st>>>(a);
}

template <int MODE>
static void launch_gemv(const GemvArgs& a, hipStream_t st) {
    if constexpr (MODE == MODE_FC2) {
        // register-resident hidden activations: 4 slots per batch tile keep the kernel inside the VGPR budget (the second
        // tile of a batch of 8 re-reads the weights through L2 while the first one streams them from HBM)
        if (a.batch >= 2) launch_gemv_bt<MODE, 4>(a, st);
        else launch_gemv_bt<MODE, 1>(a, st);
        return;
    }
    static const int bt4 = env_int("WH_GEMV_BT4", 0);   // tuning knob: batch tiles of 4 slots for every GEMV
    if (a.batch >= 5 && !bt4) launch_gemv_bt<MODE, 8>(a, st);
    else if (a.batch >= 2) launch_gemv_bt<MODE, 4>(a, st);
    else launch_gemv_bt<MODE, 1>(a, st);
}

int cross_attn_splits(int batch, int n_head) {
    // Keys per workgroup 384 / 128 / 64, chosen from the head count ONLY: the number of splits fixes the order in which the
    // partial softmax sums are combined, so it must not depend on the batch - a slot decodes to the same bits alone or in a
    // batch of 32 (tests/test_gpu_dims.py).  With >= 12 heads even a batch of 8 gives >= 384 workgroups at 4 splits.
    (void)batch;
    static const int forced = env_int("WH_XATT_PASSES", 0);     // tuning knob: 16 / 12 / 8 / 4 / 2
    const int passes = forced ? forced : (n_head >= 12 ? 12 : n_head >= 4 ? 4 : 2);
    return (kCtx + passes * 32 - 1) / (passes * 32);
}

static void launch_cross_attn(const AttnArgs& at_in, int S, int H, int B, hipStream_t st) {
    static const int nofence = env_int("WH_XATT_NOFENCE", 0);
    AttnArgs at = at_in;
    at.no_fence = nofence;
    ProfScope ps_(KK_DEC_CROSS_ATTN, st);
    const dim3 grid(S, H, B);
    static const int xlds = env_int("WH_XATT_LDS", 0);   // tuning knob: extra LDS per workgroup caps the residency
    if (S == 3) dec_cross_attn_kernel<16><<<grid, 256, xlds, st>>>(at);
    else if (S == 4) dec_cross_attn_kernel<12><<<grid, 256, xlds, st>>>(at);
    else if (S == 6) dec_cross_attn_kernel<8><<<grid, 256, xlds, st>>>(at);
    else if (S == 12) dec_cross_attn_kernel<4><<<grid, 256, xlds, st>>>(at);
    else dec_cross_attn_kernel<2><<<grid, 256, xlds, st>>>(at);
}

// The MFMA batch-tile path (decoder32.hip): embed -> per layer [QKV, self-attention, out projection, cross query, cross-attention,
// cross out projection, fc1, fc2] -> logits -> sampler.  Every activation hand-off is a plane pair in B-fragment order, every
// LayerNorm is folded into the consumer's epilogue (see decoder32.hip).
static void launch_decoder_step32(const DecodeBuffers& db, const SamplerCfg* cfg_dev, const int* suppress_dev, bool sample, hipStream_t st) {
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
        at.batch = B; at.d = d; at.n_head = H; at.layer = l; at.n_layer = L; at.n_split = S; at.q = D.q;
        at.self_k = a.self_k; at.self_v = a.self_v;
        at.cross_k = db.cross_k + (size_t)l * cross_stride; at.cross_v = db.cross_v + (size_t)l * cross_stride;
        at.att_hi = D.zb_hi; at.att_lo = D.zb_lo; at.part = db.part; at.ticket = db.ticket; at.seq = db.seq;
        at.align = db.align; at.align_slot = db.align_slot; at.n_align = db.n_align;
        { ProfScope ps_(KK_DEC_SELF_ATTN, st); dec_self_attn_kernel<<<dim3(H, B), 256, 0, st>>>(at); }
        a = base;               // x += W_o att + b_o; planes gamma_2 x, statistics for LN2
        a.N = d; a.K = d; a.Wt = t.o_t; a.zhi = D.zb_hi; a.zlo = D.zb_lo; a.bias = w.o_b; a.gamma_next = w.ln2_g;
        a.zhi_out = D.za_hi; a.zlo_out = D.za_lo; a.stat_out = D.stat; a.prof_kind = KK_DEC_OPROJ;
        launch_dec32_proj(P32_RESID, a, n_bt, st);
        a = base;               // LN2 (folded) + cross-attention query
        a.N = d; a.K = d; a.Wt = t.cq_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = t.cq_g; a.fold_c = t.cq_c; a.q = D.q; a.prof_kind = KK_DEC_CQ;
        launch_dec32_proj(P32_Q, a, n_bt, st);
        at.dbg = debug_buffer() ? debug_buffer() + (size_t)KK_DEC_CROSS_ATTN * 4096 * 8 : nullptr;
        launch_cross_attn(at, S, H, B, st);
        a = base;               // x += W_co att + b_co; planes gamma_3 x, statistics for LN3
        a.N = d; a.K = d; a.Wt = t.co_t; a.zhi = D.zb_hi; a.zlo = D.zb_lo; a.bias = w.co_b; a.gamma_next = w.ln3_g;
        a.zhi_out = D.za_hi; a.zlo_out = D.za_lo; a.stat_out = D.stat; a.prof_kind = KK_DEC_COPROJ;
        launch_dec32_proj(P32_RESID, a, n_bt, st);
        a = base;               // LN3 (folded) + fc1 + GELU -> f16 plane
        a.N = 4 * d; a.K = d; a.Wt = t.fc1_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = t.fc1_g; a.fold_c = t.fc1_c; a.h_out = D.h; a.prof_kind = KK_DEC_FC1;
        launch_dec32_proj(P32_FC1, a, n_bt, st);
        a = base;               // x += W_2 h + b_2; planes of the next layer's LN1 (or the final LayerNorm)
        a.N = d; a.K = 4 * d; a.Wt = t.fc2_t; a.zhi = D.h; a.zlo = nullptr; a.bias = w.fc2_b;
        a.gamma_next = (l + 1 < L) ? db.layers_host[l + 1].ln1_g : db.lnf_g;
        a.zhi_out = D.za_hi; a.zlo_out = D.za_lo; a.stat_out = D.stat; a.prof_kind = KK_DEC_FC2;
        launch_dec32_proj(P32_RESID, a, n_bt, st);
    }
    const bool fused = sample && db.fused_greedy;
    P32Args a = base;           // final LayerNorm (folded) + tied-embedding logits (+ the fused greedy sampler statistics)
    a.N = V; a.K = d; a.Wt = D.emb_t; a.zhi = D.za_hi; a.zlo = D.za_lo; a.fold_g = D.lg_g; a.fold_c = D.lg_c;
    a.logits = fused ? nullptr : db.logits; a.prof_kind = KK_DEC_LOGITS;
    if (fused) { a.stats = db.stats; a.sup_mask = db.sup_mask; a.cfg = cfg_dev; }
    launch_dec32_proj(P32_LOGITS, a, n_bt, st);
    if (fused) {
        ProfScope ps_(KK_SAMPLER, st);
        sampler_final_kernel<<<B, 256, 0, st>>>(cfg_dev, db.seq, db.stats, (V + 31) / 32);
    } else if (sample) {
        ProfScope ps_(KK_SAMPLER, st);
        sampler_kernel<1, 1, 1, 0><<<B, SAMP_T, 0, st>>>(cfg_dev, suppress_dev, db.seq, db.logits, 0, nullptr, nullptr);
    }
}

void launch_decoder_step(const DecodeBuffers& db, const SamplerCfg* cfg_dev, const int* suppress_dev, bool sample, hipStream_t st) {
    if (db.d32) { launch_decoder_step32(db, cfg_dev, suppress_dev, sample, st); return; }
    const int d = db.d, B = db.batch, H = db.n_head, L = db.n_layer;
    const size_t self_stride = (size_t)db.max_batch * H * kMaxTok * kHeadDim;
    const size_t cross_stride = (size_t)db.max_batch * H * kCtx * kHeadDim;
    const int S = cross_attn_splits(B, H);
    for (int l = 0; l < L; ++l) {
        const DecLayerW& w = db.layers_host[l];
        GemvArgs g{};
        g.batch = B; g.d = d; g.n_head = H; g.seq = db.seq; g.layer = l; g.n_vocab = db.n_vocab; g.x = db.x;
        // LN1 + QKV
        g.N = 3 * d; g.K = d; g.W = w.qkv_w; g.bias = w.qkv_b; g.ln_g = w.ln1_g; g.ln_b = w.ln1_b;
        g.emb = db.emb; g.pos = db.pos; g.q = db.q;
        g.self_k = db.self_k + (size_t)l * self_stride; g.self_v = db.self_v + (size_t)l * self_stride;
        { ProfScope ps_(KK_DEC_QKV, st); g_dbg_kind = KK_DEC_QKV; launch_gemv<MODE_QKV>(g, st); }
        AttnArgs at{};
        at.batch = B; at.d = d; at.n_head = H; at.layer = l; at.n_layer = L; at.n_split = S; at.q = db.q;
        at.self_k = g.self_k; at.self_v = g.self_v;
        at.cross_k = db.cross_k + (size_t)l * cross_stride; at.cross_v = db.cross_v + (size_t)l * cross_stride;
        at.att = db.att; at.part = db.part; at.ticket = db.ticket; at.seq = db.seq;
        at.align = db.align; at.align_slot = db.align_slot; at.n_align = db.n_align;
        { ProfScope ps_(KK_DEC_SELF_ATTN, st); dec_self_attn_kernel<<<dim3(H, B), 256, 0, st>>>(at); }
        // x += W_o att + b_o   (+ in the same launch, when fused: u = Wq' x + M att + c0, the LayerNorm-folded cross query)
        g.N = d; g.K = d; g.W = w.o_w; g.bias = w.o_b; g.ain = db.att;
        const bool fcq = db.fused_cq != 0;
        if (fcq) {
            GemvArgs g2 = g;
            g2.W2 = w.cqf_w; g2.bias2 = w.cqf_c0; g2.out2 = db.q; g2.N2 = d; g2.K2 = 4 * d;
            ProfScope ps_(KK_DEC_OPROJ, st); g_dbg_kind = KK_DEC_OPROJ; launch_gemv_resid_cq(g2, st);
            at.xq = db.x; at.qr = w.cqf_r; at.qc = w.cqf_c;
        } else {
            { ProfScope ps_(KK_DEC_OPROJ, st); g_dbg_kind = KK_DEC_OPROJ; launch_gemv<MODE_RESID>(g, st); }
            // LN2 + cross query
            g.W = w.cq_w; g.bias = w.cq_b; g.ln_g = w.ln2_g; g.ln_b = w.ln2_b;
            { ProfScope ps_(KK_DEC_CQ, st); g_dbg_kind = KK_DEC_CQ; launch_gemv<MODE_Q>(g, st); }
        }
        at.dbg = debug_buffer() ? debug_buffer() + (size_t)KK_DEC_CROSS_ATTN * 4096 * 8 : nullptr;
        launch_cross_attn(at, S, H, B, st);
        // x += W_co att + b_co
        g.W = w.co_w; g.bias = w.co_b;
        { ProfScope ps_(KK_DEC_COPROJ, st); g_dbg_kind = KK_DEC_COPROJ; launch_gemv<MODE_RESID>(g, st); }
        // LN3 + fc1 + GELU
        g.N = 4 * d; g.K = d; g.W = w.fc1_w; g.bias = w.fc1_b; g.ln_g = w.ln3_g; g.ln_b = w.ln3_b; g.hbuf = db.hbuf;
        { ProfScope ps_(KK_DEC_FC1, st); g_dbg_kind = KK_DEC_FC1; launch_gemv<MODE_FC1>(g, st); }
        // x += W_2 h + b_2
        g.N = d; g.K = 4 * d; g.W = w.fc2_w; g.bias = w.fc2_b;
        { ProfScope ps_(KK_DEC_FC2, st); g_dbg_kind = KK_DEC_FC2; launch_gemv<MODE_FC2>(g, st); }
    }
    GemvArgs g{};
    g.batch = B; g.d = d; g.n_head = H; g.seq = db.seq; g.layer = -1; g.n_vocab = db.n_vocab; g.x = db.x;
    g.N = db.n_vocab; g.K = d; g.W = db.emb; g.bias = nullptr; g.ln_g = db.lnf_g; g.ln_b = db.lnf_b; g.logits = db.logits;
    const bool fused = sample && db.fused_greedy;
    if (fused) { g.stats = db.stats; g.sup_mask = db.sup_mask; g.cfg = cfg_dev; }
    { ProfScope ps_(KK_DEC_LOGITS, st); g_dbg_kind = KK_DEC_LOGITS; launch_gemv<MODE_LOGITS>(g, st); }
    if (fused) {
        ProfScope ps_(KK_SAMPLER, st);
        const int nblk = (db.n_vocab + 63) / 64;      // launch_gemv_bt: 4 row groups x R = 4 x 4 passes = 64 rows per workgroup
        sampler_final_kernel<<<B, 256, 0, st>>>(cfg_dev, db.seq, db.stats, nblk);
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
void launch_alignment_mean(const float* align, int batch, int n_align, float* out, hipStream_t st) {
    // `align` / `out` point at the first of `batch` consecutive slots
    alignment_mean_kernel<<<batch * kMaxTok, 256, 0, st>>>(align, n_align, out);
}

}  // namespace wh
