// Decoder token step for gfx950: replaces the per-token CoreML TextDecoder call of
// Sources/WhisperKit/Core/TextDecoder.swift:381-418 plus the host-side K4..K8 work around it
// (updateKVCache :218-270, updateAlignmentWeights :272-296, LogitsFilter.swift, TokenSampler.swift)
// and the loop bookkeeping of decodeText (:573-757).  Everything the loop needs lives in device
// memory (SeqState), so a step is a fixed kernel chain with no host round trip: the next input token,
// the cache position and the stop flag are read from / written to SeqState by the kernels themselves.
//
// This path is HBM/L2-bandwidth and launch-latency bound (GEMV over fp16 weights, M = batch <= 8 rows):
// no MFMA.  Per layer the chain is 6 kernels; the two attention out-projections are folded into the
// attention kernels as per-head slabs (y = sum_h W_o[:, h] a_h) whose partial sums are combined, in a
// fixed order, in the prologue of the next kernel - that removes two launches per layer and keeps the
// result bit-deterministic (no float atomics).
//
//   gemv<QKV>    LN1(x) -> q (f32), k/v straight into the self-attention KV cache at position `pos`
//   self_attn    one workgroup per (head, slot): softmax(q K^T) V over <= 224 cached positions, W_o slab
//   gemv<CQ>     x += b_o + sum_h partial_h ; LN2 -> cross-attention query
//   cross_attn   one workgroup per (head, slot): 1500 cached cross K/V rows, alignment-head row, W_o slab
//   gemv<FC1>    x += b_co + sum_h partial_h ; LN3 -> GELU(fc1)
//   gemv<FC2>    x += b_2 + W_2 h
//   gemv<LOGITS> LN_f(x) . E^T  (tied embedding)
//   sampler      logits filters + greedy/top-k sample + decodeText state advance
#include "kernels.h"

namespace wh {

enum { MODE_QKV = 0, MODE_CQ = 1, MODE_FC1 = 2, MODE_FC2 = 3, MODE_LOGITS = 4 };

struct GemvArgs {
    int batch, d, n_head, N, K, rows_per_block, n_vocab;
    int layer;                 // MODE_QKV: layer index (0 -> embed)
    const f16* W;              // [N][K]
    const float* bias;         // [N] or null
    const float *ln_g, *ln_b;  // prologue LayerNorm
    const float* xin;          // residual in  [B][d]
    float* xout;               // residual out [B][d] (written by block 0 when the prologue changes x)
    const float* comb_bias;    // out-proj bias added in the combine prologue
    const float* partial;      // [B][H][d]
    const f16* emb; const float* pos;   // layer-0 embedding
    float* q;                  // [B][d] f32 query out (QKV / CQ)
    f16* self_k; f16* self_v;  // this layer's cache base [B][224][d]
    f16* hbuf;                 // [B][4d]
    float* logits;             // [B][V]
    SeqState* seq;
};

__device__ __forceinline__ bool slot_live(const SeqState* s) { return s->active && !s->done; }

// Sum 64 lanes of N values each; afterwards lane L holds the total of value index L >> (6 - log2 N)
// (N = 4, 8, 16, 32).  Costs N-1 + (6 - log2 N) shuffles instead of 6 N.
template <int N>
__device__ __forceinline__ float reduce_transpose(float (&v)[N], int lane) {
    int off = 32;
#pragma unroll
    for (int n = N; n > 1; n >>= 1, off >>= 1) {
        const bool upper = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            float send = upper ? v[i] : v[i + n / 2];
            float keep = upper ? v[i + n / 2] : v[i];
            v[i] = keep + __shfl_xor(send, off, 64);
        }
    }
    float r = v[0];
    for (; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
    return r;
}

template <int MODE, int BT>
__global__ __launch_bounds__(256) void dec_gemv_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);   // [BT][K] f32   (MODE_FC2: f16 [BT][K])
    f16* xh = reinterpret_cast<f16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.y * BT;
    const int K = a.K, d = a.d;

    bool any_live = false;
#pragma unroll
    for (int i = 0; i < BT; ++i)
        if (b0 + i < a.batch) any_live |= slot_live(&a.seq[b0 + i]);
    if (!any_live) return;

    // ------------------------------------------------------------------ prologue -> xs
    if constexpr (MODE == MODE_FC2) {
        for (int idx = tid; idx < BT * K / 8; idx += 256) {
            int b = idx / (K / 8), c = idx - b * (K / 8);
            uint4 v = (b0 + b < a.batch) ? reinterpret_cast<const uint4*>(a.hbuf + (size_t)(b0 + b) * K)[c] : uint4{0, 0, 0, 0};
            reinterpret_cast<uint4*>(xh + (size_t)b * K)[c] = v;
        }
        __syncthreads();
    } else {
        for (int idx = tid; idx < BT * d; idx += 256) {
            int b = idx / d, c = idx - b * d;
            int gb = b0 + b;
            float x = 0.0f;
            if (gb < a.batch) {
                if (MODE == MODE_QKV && a.layer == 0) {
                    int tok = min(max(a.seq[gb].next_token, 0), a.n_vocab - 1);
                    int pos = min(max(a.seq[gb].token_index, 0), kMaxTok - 1);
                    x = (float)a.emb[(size_t)tok * d + c] + a.pos[(size_t)pos * d + c];
                } else if (MODE == MODE_CQ || MODE == MODE_FC1) {
                    x = a.xin[(size_t)gb * d + c] + a.comb_bias[c];
                    const float* pp = a.partial + (size_t)gb * a.n_head * d + c;
                    for (int h = 0; h < a.n_head; ++h) x += pp[(size_t)h * d];
                } else {
                    x = a.xin[(size_t)gb * d + c];
                }
                if (blockIdx.x == 0 && (MODE == MODE_CQ || MODE == MODE_FC1 || (MODE == MODE_QKV && a.layer == 0)))
                    a.xout[(size_t)gb * d + c] = x;
            }
            xs[b * d + c] = x;
        }
        __syncthreads();
        // LayerNorm in place, one wave per slot (two-pass, same arithmetic as layernorm_kernel)
        for (int b = wave; b < BT; b += 4) {
            float s = 0.0f;
            for (int c = lane; c < d; c += 64) s += xs[b * d + c];
            const float mean = wave_sum(s) / (float)d;
            float qv = 0.0f;
            for (int c = lane; c < d; c += 64) { float t = xs[b * d + c] - mean; qv += t * t; }
            const float rstd = rsqrtf(wave_sum(qv) / (float)d + 1e-5f);
            for (int c = lane; c < d; c += 64) xs[b * d + c] = (xs[b * d + c] - mean) * rstd * a.ln_g[c] + a.ln_b[c];
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ GEMV: 4 rows per wave pass
    const int n_begin = blockIdx.x * a.rows_per_block;
    const int n_end = min(a.N, n_begin + a.rows_per_block);
    for (int n4 = n_begin + wave * 4; n4 < n_end; n4 += 16) {
        float acc[4 * BT];
#pragma unroll
        for (int i = 0; i < 4 * BT; ++i) acc[i] = 0.0f;
        const f16* wrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wrow[r] = a.W + (size_t)min(n4 + r, a.N - 1) * K;
        for (int k = lane * 8; k < K; k += 512) {
            uint4 wv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = *reinterpret_cast<const uint4*>(wrow[r] + k);
            float xk[BT][8];
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                if constexpr (MODE == MODE_FC2) {
                    f16x8 hv = *reinterpret_cast<const f16x8*>(xh + (size_t)b * K + k);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xk[b][j] = (float)hv[j];
                } else {
                    float4 x0 = *reinterpret_cast<const float4*>(xs + b * K + k);
                    float4 x1 = *reinterpret_cast<const float4*>(xs + b * K + k + 4);
                    xk[b][0] = x0.x; xk[b][1] = x0.y; xk[b][2] = x0.z; xk[b][3] = x0.w;
                    xk[b][4] = x1.x; xk[b][5] = x1.y; xk[b][6] = x1.z; xk[b][7] = x1.w;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16x8 wh8 = *reinterpret_cast<f16x8*>(&wv[r]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float wf = (float)wh8[j];
#pragma unroll
                    for (int b = 0; b < BT; ++b) acc[r * BT + b] = fmaf(wf, xk[b][j], acc[r * BT + b]);
                }
            }
        }
        float tot = reduce_transpose<4 * BT>(acc, lane);
        constexpr int SH = (BT == 1) ? 4 : (BT == 2) ? 3 : (BT == 4) ? 2 : 1;   // 6 - log2(4*BT)
        if ((lane & ((1 << SH) - 1)) == 0) {
            int idx = lane >> SH, r = idx / BT, b = idx - r * BT;
            int n = n4 + r, gb = b0 + b;
            if (n < n_end && gb < a.batch) {
                float v = tot + (a.bias ? a.bias[n] : 0.0f);
                if constexpr (MODE == MODE_QKV) {
                    int pos = min(max(a.seq[gb].token_index, 0), kMaxTok - 1);
                    if (n < d) a.q[(size_t)gb * d + n] = v;
                    else if (n < 2 * d) a.self_k[((size_t)gb * kMaxTok + pos) * d + (n - d)] = (f16)v;
                    else a.self_v[((size_t)gb * kMaxTok + pos) * d + (n - 2 * d)] = (f16)v;
                } else if constexpr (MODE == MODE_CQ) {
                    a.q[(size_t)gb * d + n] = v;
                } else if constexpr (MODE == MODE_FC1) {
                    a.hbuf[(size_t)gb * a.N + n] = (f16)gelu_erf(v);
                } else if constexpr (MODE == MODE_FC2) {
                    a.xout[(size_t)gb * d + n] = a.xin[(size_t)gb * d + n] + v;
                } else {
                    a.logits[(size_t)gb * a.N + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- attention
struct AttnArgs {
    int batch, d, n_head, layer, n_layer;
    const float* q;          // [B][d]
    const f16* self_k; const f16* self_v;   // layer base [B][224][d]
    const f16* cross_kv;     // [B*1500][L*2d]
    const f16* o_w;          // [d][d] out projection
    float* partial;          // [B][H][d]
    float* align; const int* align_slot; int n_align;
    SeqState* seq;
};

// W_o slab: partial[n] = sum_c W_o[n][h*64 + c] * a[c]   (a in LDS)
__device__ __forceinline__ void out_proj_slab(const f16* __restrict__ o_w, int d, int h, const float* a_lds,
                                              float* __restrict__ partial_row) {
    for (int n = threadIdx.x; n < d; n += blockDim.x) {
        const uint4* wp = reinterpret_cast<const uint4*>(o_w + (size_t)n * d + h * kHeadDim);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint4 wv = wp[i];
            f16x8 w8 = *reinterpret_cast<f16x8*>(&wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf((float)w8[j], a_lds[i * 8 + j], acc);
        }
        partial_row[n] = acc;
    }
}

// attention of one query against `len` rows of K/V (row stride `ld` halves); result a[64] in LDS
template <int MAXLEN>
__device__ __forceinline__ void attend(const float* q_lds, const f16* __restrict__ kb, const f16* __restrict__ vb, size_t ld,
                                       int len, float* sc /* [MAXLEN] */, float* red /* [>=16] */, float* a_out /* [64] */,
                                       float* acc4 /* [4][64] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float lmax = -INFINITY;
    for (int t = tid; t < len; t += 256) {
        const uint4* kp = reinterpret_cast<const uint4*>(kb + (size_t)t * ld);
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint4 kv = kp[i];
            f16x8 k8 = *reinterpret_cast<f16x8*>(&kv);
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf((float)k8[j], q_lds[i * 8 + j], s);
        }
        sc[t] = s;
        lmax = fmaxf(lmax, s);
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.0f;
    for (int t = tid; t < len; t += 256) {
        float p = expf(sc[t] - m);
        sc[t] = p;
        lsum += p;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int t = tid; t < len; t += 256) sc[t] *= inv;
    __syncthreads();
    // O[c] = sum_t p[t] V[t][c]: lane = (t_sub = lane >> 3, 8 channels at (lane & 7) * 8); a wave covers 8 rows per load
    const int ts = lane >> 3, c8 = (lane & 7) * 8;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
    for (int t0 = wave * 8; t0 < len; t0 += 32) {
        int t = t0 + ts;
        if (t < len) {
            uint4 vv = *reinterpret_cast<const uint4*>(vb + (size_t)t * ld + c8);
            f16x8 v8 = *reinterpret_cast<f16x8*>(&vv);
            float p = sc[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf(p, (float)v8[j], o[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = o[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        o[j] = v;
    }
    if (ts == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc4[wave * 64 + c8 + j] = o[j];
    }
    __syncthreads();
    if (tid < 64) a_out[tid] = acc4[tid] + acc4[64 + tid] + acc4[128 + tid] + acc4[192 + tid];
    __syncthreads();
}

__global__ __launch_bounds__(256) void dec_self_attn_kernel(const AttnArgs a) {
    __shared__ float q_l[64], a_l[64], sc[kMaxTok], red[16], acc4[256];
    const int h = blockIdx.x, b = blockIdx.y;
    if (!slot_live(&a.seq[b])) return;
    const int pos = min(max(a.seq[b].token_index, 0), kMaxTok - 1);
    const int d = a.d;
    if (threadIdx.x < 64) q_l[threadIdx.x] = a.q[(size_t)b * d + h * kHeadDim + threadIdx.x];
    __syncthreads();
    const f16* kb = a.self_k + (size_t)b * kMaxTok * d + h * kHeadDim;
    const f16* vb = a.self_v + (size_t)b * kMaxTok * d + h * kHeadDim;
    attend<kMaxTok>(q_l, kb, vb, (size_t)d, pos + 1, sc, red, a_l, acc4);
    out_proj_slab(a.o_w, d, h, a_l, a.partial + ((size_t)b * a.n_head + h) * d);
}

__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const AttnArgs a) {
    __shared__ float q_l[64], a_l[64], sc[kCtx + 4], red[16], acc4[256];
    const int h = blockIdx.x, b = blockIdx.y;
    if (!slot_live(&a.seq[b])) return;
    const int pos = min(max(a.seq[b].token_index, 0), kMaxTok - 1);
    const int d = a.d;
    if (threadIdx.x < 64) q_l[threadIdx.x] = a.q[(size_t)b * d + h * kHeadDim + threadIdx.x];
    __syncthreads();
    const size_t ld = (size_t)a.n_layer * 2 * d;
    const f16* kb = a.cross_kv + (size_t)b * kCtx * ld + (size_t)a.layer * 2 * d + h * kHeadDim;
    const f16* vb = kb + d;
    attend<kCtx>(q_l, kb, vb, ld, kCtx, sc, red, a_l, acc4);
    // alignment-head row: DecodingCache.alignmentWeights row tokenIndex + 1 (TextDecoder.swift:272-296)
    if (a.align) {
        int slot = a.align_slot[a.layer * a.n_head + h];
        if (slot >= 0 && pos + 1 < kMaxTok) {
            float* dst = a.align + (((size_t)b * kMaxTok + pos + 1) * a.n_align + slot) * kCtx;
            for (int t = threadIdx.x; t < kCtx; t += 256) dst[t] = sc[t];
        }
    }
    out_proj_slab(a.o_w, d, h, a_l, a.partial + ((size_t)b * a.n_head + h) * d);
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
        int blank = 0, ts_active = 0, r1lo = 0, r1hi = 0, r2lo = 0, r2hi = 0;
        if (DO_FILTER) {
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
        }
        sh_i[0] = blank; sh_i[1] = ts_active; sh_i[2] = r1lo; sh_i[3] = r1hi; sh_i[4] = r2lo; sh_i[5] = r2hi;
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
        if (DO_ADVANCE) {
            // decodeText bookkeeping, TextDecoder.swift:573-757
            const int ti_cur = sq->token_index;
            const bool isFirstToken = ti_cur == cfg.prefilled_index;
            const bool tooLow = isFirstToken && cfg.has_first_token_threshold && lp < cfg.first_token_log_prob_threshold;
            const bool completed = tok == cfg.end_token;
            const bool segDone = completed || n_tok >= kMaxTok - 1 || tooLow;
            sq->steps += 1;
            sq->first_token_too_low = tooLow ? 1 : 0;
            if (segDone) {
                sq->done = 1;
            } else {
                const bool isPrefill = ti_cur < sq->prompt_len - 1;
                int nt = n_tok;
                if (!isPrefill) { sq->tokens[nt] = tok; sq->logprobs[nt] = lp; nt += 1; sq->n_tokens = nt; }
                const int ti_next = ti_cur + 1;
                if (ti_next >= cfg.loop_count) {
                    sq->done = 1;
                } else {
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
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- launchers
template <int MODE>
static void launch_gemv(const GemvArgs& a, hipStream_t st) {
    int bt = a.batch >= 8 ? 8 : a.batch >= 3 ? 4 : a.batch;   // 1, 2, 4, 8
    if (bt == 3) bt = 4;
    dim3 g((a.N + a.rows_per_block - 1) / a.rows_per_block, (a.batch + bt - 1) / bt);
    size_t smem = (MODE == MODE_FC2) ? (size_t)bt * a.K * sizeof(f16) : (size_t)bt * a.K * sizeof(float);
    if (smem > 64 * 1024) {   // above the default dynamic-LDS limit: raise it once per instantiation (160 KB per CU on gfx950)
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_gemv_kernel<MODE, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
    }
    switch (bt) {
        case 1: dec_gemv_kernel<MODE, 1><<<g, 256, smem, st>>>(a); break;
        case 2: dec_gemv_kernel<MODE, 2><<<g, 256, smem, st>>>(a); break;
        case 4: dec_gemv_kernel<MODE, 4><<<g, 256, smem, st>>>(a); break;
        default: dec_gemv_kernel<MODE, 8><<<g, 256, smem, st>>>(a); break;
    }
}


void launch_decoder_step(const DecodeBuffers& db, const SamplerCfg* cfg_dev, const int* suppress_dev, bool sample, hipStream_t st) {
    const int d = db.d, B = db.batch, H = db.n_head, L = db.n_layer;
    float* xcur = db.xa;
    float* xalt = db.xb;
    for (int l = 0; l < L; ++l) {
        const DecLayerW& w = db.layers_host[l];
        GemvArgs g{};
        g.batch = B; g.d = d; g.n_head = H; g.seq = db.seq; g.layer = l; g.n_vocab = db.n_vocab;
        // QKV
        g.N = 3 * d; g.K = d; g.rows_per_block = 16; g.W = w.qkv_w; g.bias = w.qkv_b; g.ln_g = w.ln1_g; g.ln_b = w.ln1_b;
        g.xin = xcur; g.xout = xcur; g.emb = db.emb; g.pos = db.pos; g.q = db.q;
        g.self_k = db.self_k + (size_t)l * B * kMaxTok * d; g.self_v = db.self_v + (size_t)l * B * kMaxTok * d;
        { ProfScope ps_(KK_DEC_QKV, st); launch_gemv<MODE_QKV>(g, st); }
        AttnArgs at{};
        at.batch = B; at.d = d; at.n_head = H; at.layer = l; at.n_layer = L; at.q = db.q; at.self_k = g.self_k; at.self_v = g.self_v;
        at.cross_kv = db.cross_kv; at.o_w = w.o_w; at.partial = db.partial; at.seq = db.seq;
        at.align = db.align; at.align_slot = db.align_slot; at.n_align = db.n_align;
        { ProfScope ps_(KK_DEC_SELF_ATTN, st); dec_self_attn_kernel<<<dim3(H, B), 256, 0, st>>>(at); }
        // cross query: x' = x + b_o + sum partial
        g.N = d; g.K = d; g.rows_per_block = 16; g.W = w.cq_w; g.bias = w.cq_b; g.ln_g = w.ln2_g; g.ln_b = w.ln2_b;
        g.xin = xcur; g.xout = xalt; g.comb_bias = w.o_b; g.partial = db.partial;
        { ProfScope ps_(KK_DEC_CQ, st); launch_gemv<MODE_CQ>(g, st); }
        at.o_w = w.co_w;
        { ProfScope ps_(KK_DEC_CROSS_ATTN, st); dec_cross_attn_kernel<<<dim3(H, B), 256, 0, st>>>(at); }
        // fc1: x'' = x' + b_co + sum partial
        g.N = 4 * d; g.K = d; g.rows_per_block = 16; g.W = w.fc1_w; g.bias = w.fc1_b; g.ln_g = w.ln3_g; g.ln_b = w.ln3_b;
        g.xin = xalt; g.xout = xcur; g.comb_bias = w.co_b; g.hbuf = db.hbuf;
        { ProfScope ps_(KK_DEC_FC1, st); launch_gemv<MODE_FC1>(g, st); }
        // fc2: x''' = x'' + b_2 + W_2 h   (in place on xcur)
        g.N = d; g.K = 4 * d; g.rows_per_block = 16; g.W = w.fc2_w; g.bias = w.fc2_b; g.xin = xcur; g.xout = xcur;
        { ProfScope ps_(KK_DEC_FC2, st); launch_gemv<MODE_FC2>(g, st); }
    }
    GemvArgs g{};
    g.batch = B; g.d = d; g.n_head = H; g.seq = db.seq; g.layer = -1; g.n_vocab = db.n_vocab;
    g.N = db.n_vocab; g.K = d; g.rows_per_block = 64; g.W = db.emb; g.bias = nullptr; g.ln_g = db.lnf_g; g.ln_b = db.lnf_b;
    g.xin = xcur; g.logits = db.logits;
    { ProfScope ps_(KK_DEC_LOGITS, st); launch_gemv<MODE_LOGITS>(g, st); }
    if (sample) { { ProfScope ps_(KK_SAMPLER, st); sampler_kernel<1, 1, 1, 0><<<B, SAMP_T, 0, st>>>(cfg_dev, suppress_dev, db.seq, db.logits, 0, nullptr, nullptr); } }
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

__global__ void alignment_mean_kernel(const float* __restrict__ align, int n_align, float* __restrict__ out) {
    // align [B][224][n_align][1500] -> out [B][224][1500]
    size_t row = blockIdx.x;  // b * 224 + pos
    const float* src = align + row * n_align * kCtx;
    const float inv = 1.0f / (float)n_align;
    for (int t = threadIdx.x; t < kCtx; t += blockDim.x) {
        float s = 0.0f;
        for (int j = 0; j < n_align; ++j) s += src[(size_t)j * kCtx + t];
        out[row * kCtx + t] = s * inv;
    }
}
void launch_alignment_mean(const float* align, int batch, int n_align, float* out, hipStream_t st) {
    alignment_mean_kernel<<<batch * kMaxTok, 256, 0, st>>>(align, n_align, out);
}

}  // namespace wh
