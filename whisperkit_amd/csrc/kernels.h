// Launcher declarations shared between the .hip kernel files and the C-ABI implementation.
#pragma once
#include "common.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace wh {

// Run `f` once per HIP device of the process (hipFuncSetAttribute is per device): sessions of different GPUs may be driven from
// different host threads of one process.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    template <class F> void run(F f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lk(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        f();
        done.fetch_or(bit, std::memory_order_release);
    }
};

// ---------------------------------------------------------------------------------------------- 24-bit rows (hr24)
// The cross-attention key / value rows of the K / V-row mode: x ~ hi + lo * ulp(hi) / 256 with hi = Float16(x) and lo a signed byte - 19 mantissa
// bits in 3 bytes.  Float16 rows cost 7.1e-3 sigma of the logits under a sharp softmax (keys) and 1.9e-3 sigma (values); fp32 rows (the first
// round-5 form) cost twice the stream; this keeps the error below 1e-5 sigma at 1.5 x the Float16 bytes.
struct hr24 {};
__host__ __device__ __forceinline__ float hr24_unit(unsigned hi_bits) {      // ulp(hi) / 256 = 2^(e - 33), e = the biased exponent (subnormals: e = 1)
    unsigned e = (hi_bits >> 10) & 31u;
    e = e < 1u ? 1u : e;
    const unsigned bits = (e + 94u) << 23;
    float f;
    __builtin_memcpy(&f, &bits, 4);
    return f;
}
__device__ __forceinline__ void hr24_encode(float x, f16& hi, signed char& lo) {
    hi = (f16)x;
    unsigned short hb;
    __builtin_memcpy(&hb, &hi, 2);
    unsigned e = ((unsigned)hb >> 10) & 31u;
    e = e < 1u ? 1u : e;
    const unsigned ibits = (160u - e) << 23;          // 2^(33 - e) = 1 / hr24_unit
    float inv;
    __builtin_memcpy(&inv, &ibits, 4);
    const float r = rintf((x - (float)hi) * inv);     // |x - hi| <= ulp / 2  ->  |r| <= 128
    lo = (signed char)(int)fminf(fmaxf(r, -128.0f), 127.0f);
}

// ---------------------------------------------------------------------------------------------- measurement
// Every kernel launch of the hot path is tagged with a kind; when a KernelProfiler is armed on the calling thread
// (wh_measure_kernels) each launch is bracketed by a HIP event pair on the launch stream.
enum KernelKind {
    KK_MEL_POWER = 0, KK_MEL_FINALIZE, KK_CONV1, KK_CONV2, KK_LAYERNORM, KK_ENC_QKV, KK_ENC_ATTN, KK_ENC_O, KK_ENC_FC1, KK_ENC_FC2,
    KK_CROSS_KV, KK_DEC_QKV, KK_DEC_SELF_ATTN, KK_DEC_OPROJ, KK_DEC_CQ, KK_DEC_CROSS_ATTN, KK_DEC_COPROJ, KK_DEC_FC1, KK_DEC_FC2, KK_DEC_LOGITS,
    KK_SAMPLER, KK_DEC_EMBED, KK_DEC_XQK, KK_DEC_XVUP,
    KK_COUNT
};
struct KernelProfiler {
    std::vector<hipEvent_t> ev;   // 2 per recorded launch
    std::vector<int> kind;
    size_t n = 0, capacity = 0;
};
extern thread_local KernelProfiler* g_prof;
struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(int kind, hipStream_t s) : st(s), on(false) {
        KernelProfiler* p = g_prof;
        if (p && kind >= 0 && p->n < p->capacity) { on = true; p->kind[p->n] = kind; (void)hipEventRecord(p->ev[2 * p->n], st); }
    }
    ~ProfScope() { if (on) { KernelProfiler* p = g_prof; (void)hipEventRecord(p->ev[2 * p->n + 1], st); p->n++; } }
};

struct MelTables {
    const float* basis_c;   // [200][208] w[n] cos(2 pi n k / 400), n = 1..200
    const float* basis_s;   // [200][208] w[n] sin(2 pi n k / 400)
    const float* filt;      // [201][n_mels] slaney mel filterbank
    const int2* filt_range; // [n_mels] first / last non-zero bin
    const float* filt_c;    // compact non-zero weights, filter m at [filt_off[m], filt_off[m] + last - first]
    const int* filt_off;    // [n_mels]
    int filt_nnz;
    int n_mels;
};

void launch_log_mel(const MelTables& t, const float* pcm, const int* n_valid, int batch, float* logspec, unsigned* maxkey,
                    f16* mel_t, float* mel_f32, hipStream_t st);
void launch_mel_import(const float* mel_f32, int n_mels, int batch, f16* mel_t, hipStream_t st);

// ---------------------------------------------------------------------------------------------- GEMM
// C[M][N] = A[M][K] * W[N][K]^T (+ bias[N]) with fused epilogues.  A rows may be an overlapping-row view:
// address(m, k) = A + (m / a_rows_per_batch) * a_batch_stride + (m % a_rows_per_batch) * lda + k.
enum GemmEpi {
    EPI_F16 = 0,        // out16[m*ldc + n] = v
    EPI_GELU_F16 = 1,   // out16 = gelu(v)
    EPI_RESID_F32 = 2,  // x32[m*ldc + n] += v                        (fp32 residual stream)
    EPI_QKV_ENC = 3,    // n<d: q16[m*d+n]; n<2d: k16[m*d+n-d]; else V^T[(b*H+h)*64+c][t]   (encoder attention operands)
    EPI_CONV1 = 4,      // out16[(b*3002 + t + 1)*ldc + n] = gelu(v)  (padded time-major input of conv2)
    EPI_CONV2 = 5,      // x32[m*ldc + n] = gelu(v) + pos[t*ldc + n]
    EPI_F32 = 6,        // out32[m*ldc + n] = v
    EPI_CROSS_KV = 7,   // n = l*2d + kv*d + h*64 + c, m = b*1500 + t -> (kv ? V : K) hi / lo [(((l*Bmax + b)*H + h)*1500 + t)*64 + c]  (24-bit rows, hr24)
};

struct GemmArgs {
    const f16* A;
    const f16* W;
    const float* bias;   // may be null
    int M, N, K;
    int lda;
    int a_rows_per_batch;      // = M for a plain matrix
    long long a_batch_stride;  // elements
    int ldc;
    f16* out16;
    float* out32;
    // EPI_QKV_ENC
    f16* k16;
    f16* vt16;
    int d_model;
    // EPI_CONV2
    const float* pos;
    int rows_per_batch_out;    // 3000 (conv1) / 1500 (conv2, qkv)
    int max_batch = 0;         // EPI_CROSS_KV: slot stride of the head-major cross K/V layout
    f16 *kv_k_hi = nullptr, *kv_v_hi = nullptr;              // EPI_CROSS_KV: the cross-attention key / value rows as hr24 (below): Float16 part ...
    signed char *kv_k_lo = nullptr, *kv_v_lo = nullptr;      // ... and the 8-bit residuals
    int prof_kind = -1;        // KernelKind of this launch (measurement only)
};

void launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------------------------- LayerNorm
// y = LN(x) * g + b over rows of length d; x fp32 [rows][d]; writes f16 and/or f32
void launch_layernorm(const float* x, const float* g, const float* b, int rows, int d, f16* y16, float* y32, hipStream_t st);

// ---------------------------------------------------------------------------------------------- encoder attention
// q16,k16: [B*1500][d] (q pre-scaled), vt16: [B][H][64][1536]; out16: [B*1500][d]
void launch_encoder_attention(const f16* q16, const f16* k16, const f16* vt16, f16* out16, int batch, int n_head, int d, hipStream_t st);

// ---------------------------------------------------------------------------------------------- decoder
struct DecLayerW {
    const float *ln1_g, *ln1_b; const f16* qkv_w; const float* qkv_b;
    const f16* o_w; const float* o_b;
    const float *ln2_g, *ln2_b; const f16* cq_w; const float* cq_b;
    const f16* co_w; const float* co_b;
    const float *ln3_g, *ln3_b; const f16* fc1_w; const float* fc1_b; const f16* fc2_w; const float* fc2_b;
};

// filter / sampler configuration resident on the device for a decode_text call
struct SamplerCfg {
    int n_vocab;
    int end_token, no_timestamps_token, time_token_begin, transcribe_token, translate_token, whitespace_token;
    int is_multilingual;
    int suppress_blank, prefilled_index;     // SuppressBlankFilter(sampleBegin = prefilledIndex)
    int timestamp_rules, initial_prompt_index;
    int language_filter, language_token_begin, n_language_tokens;
    int n_suppress;                          // ids in SeqState-independent list `suppress`
    int top_k;
    int loop_count;                          // min(sampleLength, 223)
    int has_first_token_threshold; float first_token_log_prob_threshold;
    int f16_logits;                          // reference-numerics switch: Float16 logits + Float16 timestamp-mass comparison
    unsigned long long seed;
};

constexpr int kMaxSuppress = 256;
constexpr int kMaxPrompt = 232;

// per-slot decode state (device memory)
struct SeqState {
    int tokens[kMaxTok + 8];      // currentTokens
    float logprobs[kMaxTok + 8];
    int n_tokens;                 // currentTokens.count
    int token_index;              // tokenIndex = cacheLength of the next decoder call
    int next_token;               // input id of the next decoder call
    int done;                     // loop left (EOT / length / first-token threshold / loopCount)
    int first_token_too_low;
    int steps;
    int active;
    float temperature;
    int prompt_len;               // initialPromptIndex
    int f_rules[6];               // filter rules of the NEXT sampling step: blank, ts_active, r1_lo, r1_hi, r2_lo, r2_hi
    int pad;
};

struct DecodeBuffers {
    int batch, max_batch, d, n_head, n_layer, n_vocab;
    const f16* emb;          // [V][d]
    const float* pos;        // [448][d]
    const DecLayerW* layers_host; // host array [L] of device pointers
    const float *lnf_g, *lnf_b;
    f16* self_k;             // [L][Bmax][H][224][64]  head-major self-attention cache
    f16* self_v;
    const f16 *cross_k_hi, *cross_v_hi;           // [L][Bmax][H][1500][64] head-major cross-attention K / V rows in 24 bits per element (hr24: Float16 +
    const signed char *cross_k_lo, *cross_v_lo;   //  8-bit residual), written by the cross-K/V GEMM epilogue; K / V-row mode only
    float* x;                // [n_bt*32][d] residual stream (= d32->x)
    float* q;                // [n_bt*32][d] f32 query (= d32->q)
    float* part;             // [B][H][kMaxSplit][kPartStride] cross-attention split partials
    int* ticket;             // [B][H]
    float* logits;           // [B][V]
    float* stats;            // [B][kStatBlocks][8] per-workgroup softmax statistics of the logits kernel (fused greedy sampler)
    const unsigned char* sup_mask;   // [V] SuppressTokensFilter as a byte mask
    int fused_greedy;        // every active slot samples at T = 0: filters + statistics in the logits epilogue, tiny final kernel
    float* align;            // [B][224][n_align][1500] raw score rows of the alignment heads (or null)
    const int* align_slot;   // [L*H] -> slot index or -1
    int n_align;
    SeqState* seq;           // [B]
    int cross_div;           // > 1: slot b reads the cross K / V of slot b / cross_div (beam search: the beams of an audio share ONE copy)
    int* xattn_gate;         // null, or the model's cross-attention gate word (dec_shared.h): concurrent sessions take turns at the HBM
    int self_rows;           // self-attention fetch bound: 1 + the largest token_index a live slot can have in these launches (1..224);
                             // the kernel instantiation covers ceil(self_rows / 32) passes of 32 rows
    const int* self_owner;   // null, or [Bmax][224]: the slot whose cache holds row r of slot b's history (beam search: a beam that
                             // continues another beam's sequence reads that beam's rows in place - no cache rearrangement copies)
    const struct Dec32* d32; // activation planes / split-K scratch / tiled weights of the projection kernels (decoder32.hip)
    const struct Xabs* xabs; // non-null: weight-absorbed cross-attention over the encoder output (xabs.hip) instead of the cross K / V stream
};
constexpr int kStatBlocks = 1792; // >= workgroups of the logits kernel (V / 64 rows: GEMV path, V / 32 rows: MFMA path), multiple of 256
// Largest vocabulary the sampling kernels cover: sampler_kernel holds SAMP_T x SAMP_E = 1024 x 51 ids in registers, the fused greedy
// path writes one statistics record per 32 logits rows (kStatBlocks of them).  wh_model_create rejects anything larger (Whisper
// vocabularies are 51864 / 51865 / 51866, Utilities/ModelUtilities.swift:124-170).
constexpr int kMaxVocab = 52224;
static_assert(kMaxVocab <= kStatBlocks * 32, "fused greedy sampler: one statistics record per 32-row logits tile");
constexpr int kMaxSplit = 24;   // cross-attention key splits (64 keys per workgroup at the finest)
constexpr int kPartStride = 96; // floats per split partial (m, l, o[64]) padded to 3 x 128 bytes: no cache line is shared between splits
int cross_attn_splits(int batch, int n_head);

// ---------------------------------------------------------------------------------------------- MFMA decode path (decoder32.hip)
// Batch tiles of 32 slots are the N side of v_mfma_f32_32x32x16_f16; weights are re-tiled at model load so that one
// 1 KB coalesced load IS one A fragment: Wt[row tile][k tile][lane 64][8 halves], lane l = (row & 31) | (k half << 5).
// Activations travel between kernels in the matching B-fragment order ("planes"): Z[batch tile][k tile][k half][slot 32][8],
// f16 hi plane + f16 lo plane (lo = (z - hi) * 2048: 22 mantissa bits, no subnormals), written by the producing kernel.
struct Dec32LayerW {
    const f16 *qkv_t, *o_t, *cq_t, *co_t, *fc1_t, *fc2_t;      // tiled weights
    const float *qkv_g, *qkv_c, *cq_g, *cq_c, *fc1_g, *fc1_c;  // LayerNorm folds: g = W gamma, c = W beta + bias
};
struct Dec32 {
    const Dec32LayerW* layers_host;   // [L]
    const f16* emb_t; const float *lg_g, *lg_c;   // tied-embedding logits: tiled [ceil(V/32)*32][d], folds of the final LayerNorm
    int n_bt;                // batch tiles allocated (ceil(max_batch / 32))
    float* x;                // [n_bt*32][d] residual stream
    float* q;                // [n_bt*32][d] query of the attention kernels
    f16 *za_hi, *za_lo;      // [n_bt][d/16][2][32][8] gamma * x of the next LayerNorm consumer
    f16 *zb_hi, *zb_lo;      // attention output (input of the out projections)
    f16 *h, *h_lo;           // [n_bt][4d/16][2][32][8] GELU(fc1) as an f16 hi | lo plane pair (a single f16 plane moved the large-v3 logits by 1e-3)
    float2* stat;            // [n_bt][d/32][32] per-row-tile (mean, M2) of the residual stream: LayerNorm statistics, Chan-combined
    float* part; int* ticket; // split-K partial tiles + arrival counters
    size_t part_floats;
};
constexpr int kD32PartFloats = 2 * 1024 * 1024;   // per batch tile: row tiles x K splits x 1024 <= 2 M floats
enum { P32_QKV = 0, P32_Q = 1, P32_RESID = 2, P32_FC1 = 3, P32_LOGITS = 4 };
struct P32Args {
    int batch, N, K, d, n_head, n_vocab;
    int ks, tw;              // K splits across workgroups; 16-wide k tiles per wave = K / (64 ks)
    int n_bt;                // batch tiles of this launch (set by the launcher)
    int rt;                  // weight-row tiles per workgroup, 1 or 2 (set by the launcher)
    const f16* Wt;
    const f16 *zhi, *zlo;    // input planes (zlo == null: single f16 plane)
    const float2* stat_in; int n_stat; const float *fold_g, *fold_c;           // LayerNorm fold (QKV, Q, FC1, LOGITS)
    const float* bias; float* x; const float* gamma_next; f16 *zhi_out, *zlo_out; float2* stat_out;   // RESID
    float* q; f16 *self_k, *self_v;                                            // QKV / Q
    f16 *h_out, *h_out_lo;                                                     // FC1
    float* logits; float* stats; const unsigned char* sup_mask; const SamplerCfg* cfg;   // LOGITS (+ fused greedy statistics)
    float* part; int* ticket;
    const SeqState* seq;
    int* gate;                   // P32_Q only: workgroup 0 takes the cross-attention gate before it exits (or null)
    int prof_kind;
    unsigned long long* dbg;     // WH_DBG=1: 8 wall-clock stamps per workgroup (tools/probe_dec32.py)
};
void launch_dec32_proj(int mode, const P32Args& a, int n_bt, hipStream_t st);
void launch_dec32_embed(const f16* emb, const float* pos, const SeqState* seq, int batch, int d, int n_vocab, int n_bt, float* x,
                        const float* gamma_next, f16* zhi, f16* zlo, float2* stat, hipStream_t st);
// model-load helpers: re-tile W[N][K] -> out[ceil(N/32)][K/16][64][8]; g[n] = sum_k W[n][k] gamma[k], c[n] = sum_k W[n][k] beta[k] + bias[n] (f64 sums)
void dec32_tile_weights(const f16* W, int N, int K, f16* out, hipStream_t st);
void dec32_fold_vectors(const f16* W, int N, int K, const float* gamma, const float* beta, const float* bias, float* g, float* c, hipStream_t st);
int dec32_ksplit(int mode, int N, int K, bool f16_input);

// ---------------------------------------------------------------------------------------------- absorbed cross-attention (xabs.hip)
constexpr int kXabsMaxSlotsPerWorkgroup = 16;   // slots one xabs_attn workgroup streams one after the other (wh_session_options)
constexpr int kMaxSessionSlots = 256;    // windows one session decodes in lock-step (eight 32-slot batch tiles of the decoder projections)
constexpr int kXabsAutoMinSlots = 28;   // wh_session_create picks the absorbed path from this many slots (WH_XABS_MIN_SLOTS overrides): measured large-v3,
                                        // one stream, ms per decoder step with 24-bit K / V rows vs absorbed (4 splits): 16 slots 3.27 / 3.91, 24 slots 3.88 / 4.00, 28 slots
                                        // 4.17 / 4.05, 32 slots 4.40 / 4.09 (profiles/r05h_*, r05a_*)
constexpr int kXabsSplits = 4;      // most key splits per slot (buffer sizes); a session uses Xabs::n_split of them, fixed at creation
// key splits of a session: one workgroup per (slot, split) owns a whole CU (LDS, registers), so slots x splits is the number of CUs the
// kernel takes.  WH_XABS_SPLITS overrides (A/B).
int xabs_splits(int max_batch);
int xabs_auto_splits(int max_batch);     // the automatic choice without the WH_XABS_SPLITS override (wh_xabs_auto_splits)
struct XabsLayerW {
    const f16* wkT;      // W_k^T tiles [H][d / 32][4][64][8] (A fragments of the Q' projection)
    const f16* wv_t;     // W_v in the decoder projection tiling [d / 32][d / 16][64][8]
    const float* bv;     // [d]
};
struct Xabs {
    const XabsLayerW* layers_host;   // [L]
    const f16* enc;      // [Bmax][1500][d] encoder output (f16), the session's enc16
    f16 *qf_hi, *qf_lo;  // [Bmax][heads padded to 16 / 32][d] absorbed queries Q' = W_k^T q, f16 hi | lo
    float* part;         // [splits][H][d / 8][Bmax][8] unnormalised O' of every key split
    float2* ml;          // [splits][H][Bmax] (running maximum, sum)
    int n_split;         // key splits per slot (1 .. kXabsSplits), a constant of the session (the combine order fixes the bits)
    int spw;             // slots per xabs_attn workgroup (>= 1), a constant of the session: workgroups per launch = ceil(batch / spw) x n_split
};
struct XabsArgs {
    int batch, max_batch, d, n_head, layer, n_split, cross_div;
    const f16* enc; const float* q;
    const f16 *wkT, *wv_t; const float* bv;
    f16 *qf_hi, *qf_lo; float* part; float2* ml;
    f16 *att_hi, *att_lo;
    float* align; const int* align_slot; int n_align;
    const SeqState* seq;
    float* kpart; int* ticket;       // K-slice scratch of xabs_vup (the projection kernels' part / ticket buffers)
    unsigned long long* dbg;         // WH_DBG=1 timeline stamps of xabs_attn
    int ablate;                      // WH_XABS_ABLATE (timing probe, results are garbage): bit 0 no LDS-DMA, bit 1 no S / softmax / P V work
    int* gate;                       // cross-attention gate (dec_shared.h, WH_XATT_GATE=1): xabs_qk takes it, xabs_attn's last workgroup returns it
    int spw;                         // xabs_attn: slots per workgroup (round 6): the launch has ceil(batch / spw) x n_split workgroups, each streams its slots one after the other
};
bool xabs_supported(int d, int n_head);
void xabs_tile_wk(const f16* Wk, int d, int H, f16* out, hipStream_t st);
void launch_xabs_qk(const XabsArgs& a, int n_bt, hipStream_t st);
void launch_xabs_attn(const XabsArgs& a, hipStream_t st);
void launch_xabs_vup(const XabsArgs& a, int n_bt, hipStream_t st);

// one decoder forward + (optionally) fused filter/sample/state-advance for all slots
void launch_decoder_step(const DecodeBuffers& db, const SamplerCfg* cfg_dev, const int* suppress_dev, bool sample, hipStream_t st);
// filter rules of the first sampling step of a decodeText call (later steps: computed by the sampler itself)
void launch_rules_init(const SamplerCfg* cfg_dev, SeqState* seq, int batch, hipStream_t st);
// standalone filter / sampler entry points (KAT surface of the C ABI)
void launch_filter_only(const SamplerCfg* cfg_dev, const int* suppress_dev, SeqState* seq, float* logits, int n_vocab, hipStream_t st);
void launch_sample_only(const SamplerCfg* cfg_dev, SeqState* seq, float* logits, int n_vocab, int counter, int* token_out, float* logprob_out, hipStream_t st);
// filter + sample without advancing the decode state (detectLanguage)
void launch_filter_sample(const SamplerCfg* cfg_dev, const int* suppress_dev, SeqState* seq, float* logits, int batch, int* token_out, float* logprob_out, hipStream_t st);
constexpr int kBeamTopK = 16;   // row stride of the top-k outputs: beam sizes up to 15 (topk(beam_size + 1))
// beam search: the LogitsFiltering rules, log-softmax and the K best entries of every live slot's row in one pass (row in registers)
void launch_beam_filter_topk(const SamplerCfg* cfg_dev, const int* suppress_dev, SeqState* seq, float* logits, int batch, int K, float* lp_out,
                             int* tok_out, hipStream_t st);
// mean over alignment heads -> [B][224][1500]
void launch_alignment_mean(const float* align, int batch, int n_align, float* out, hipStream_t st);
// openai/whisper-style alignment post-processing of one slot (z-normalise over the token rows, median filter, head mean)
void launch_alignment_postprocess(const float* align, int n_align, float* prob_tmp, float* stat_tmp, int* row_written, int znorm, int median_width,
                                  float* out, hipStream_t st);
void launch_f32_to_f16(const float* in, f16* out, size_t n, hipStream_t st);
void launch_f16_to_f32(const f16* in, float* out, size_t n, hipStream_t st);

}  // namespace wh
