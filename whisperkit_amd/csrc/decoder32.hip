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
#include "dec32_body.h"

namespace wh {

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

// ---------------------------------------------------------------------------------------------- the projection kernel
// Body: dec32_body.h (shared with the fused projection + attention launches of decoder_fused.hip).
// __launch_bounds__(256, 2): capping the wave at 256 unified registers keeps the accumulators in arch VGPRs (with 512 allowed the
// compiler parked them in AccVGPRs and copied all 32 of them out and back in every loop iteration: 160 v_accvgpr moves per launch);
// every instantiation fits (88 - 204 VGPRs, no scratch), two workgroups can share a CU.
template <int MODE, bool HILO, int TC>
__global__ __launch_bounds__(256, 2) void dec32_proj_kernel(const P32Args a) { dec32_proj_body<MODE, HILO, TC>(a, (int)blockIdx.x); }

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

template <int MODE, bool HILO>
static void launch_tc(const P32Args& a, dim3 grid, hipStream_t st) {
    const int tw = a.tw;
    // chunks of <= 5 k-tiles: 6 would put the LOGITS instantiation at 226 VGPRs + accumulators = one wave per SIMD (tiny.en: 22 -> 47 us)
    if (tw % 5 == 0) dec32_proj_kernel<MODE, HILO, 5><<<grid, 256, 0, st>>>(a);
    else if (tw % 4 == 0) dec32_proj_kernel<MODE, HILO, 4><<<grid, 256, 0, st>>>(a);
    else if (tw % 3 == 0) dec32_proj_kernel<MODE, HILO, 3><<<grid, 256, 0, st>>>(a);
    else if (tw % 2 == 0) dec32_proj_kernel<MODE, HILO, 2><<<grid, 256, 0, st>>>(a);
    else dec32_proj_kernel<MODE, HILO, 1><<<grid, 256, 0, st>>>(a);
}

unsigned long long* debug_buffer();
void launch_dec32_proj(int mode, const P32Args& a_in, int n_bt, hipStream_t st) {
    P32Args a = a_in;
    a.dbg = (debug_buffer() && a.prof_kind >= 0) ? debug_buffer() + (size_t)a.prof_kind * 4096 * 8 : nullptr;   // WH_DBG=1 timeline probe
    const bool hilo = a.zlo != nullptr;
    a.ks = dec32_ksplit(mode, a.N, a.K, !hilo);
    a.tw = a.K / (64 * a.ks);
    a.n_bt = n_bt;
    const int nx = ((a.N + 31) / 32) * a.ks;
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
