// DRAFT - NOT BUILT INTO libwhisperhip.so, NEVER RUN ON HARDWARE.  Round-2 starting point.
// `make -C whisperkit_amd/csrc experimental` only compiles it for gfx950 so that register use / LDS / ISA can be inspected.
//
// Decoder projections as an MFMA skinny GEMM for 16 < B <= 32 sequences per step:
//     Y[b][n] = sum_k act(b, k) * W[n][k]          b < 32, W f16 [N][K] row-major (K contiguous)
// Why: at B = 32 the lane-per-K GEMVs of decoder.hip take 18-29 us per launch (0.44 TB/s on 56 MB per layer,
// profiles/r01g_bench_largev3_b32_kernels.json): their FMA loops grow with the batch.  With the batch as the 32-wide side of a
// v_mfma_f32_32x32x16_f16 tile the arithmetic is 20 MFMAs per wave and the kernel is a weight stream.
//
// Work split: one workgroup (256 threads = 4 waves) owns 32 consecutive output rows n and all 32 batch columns; its 4 waves
// split K, each keeping one f32x16 accumulator, reduced through LDS at the end (deterministic order: wave 0..3).
// Fragments need no LDS staging and no transposition (same layout fact gemm.hip relies on):
//   a = W fragment   : lane l -> row n0 + (l & 31), 8 consecutive k at kb + 8 * (l >> 5): ONE 16-byte global load
//   b = act fragment : lane l -> batch row (l & 31), same 8 k: 2 x float4 of x (L2 hits), LayerNorm applied on the fly - the row
//                      statistics are per-LANE scalars because a lane always works on the same batch row
//   acc              : lane l holds batch row b = l & 31 and W rows n0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15
// All W loads of a wave's K range are issued before anything else (20 x 16 B in flight per lane for K = 1280): the LayerNorm
// statistics and the activation staging run underneath the weight stream.
//
// Modes drafted here: FC1 (LN + W1, GELU, f16 out), RESID (x += W * in + bias for f32 `att` or f16 `hbuf` input), QKV (LN, layer-0
// embedding prologue, q out + K/V cache scatter at token_index) and LOGITS (final LN, tied embedding, partial last tile; the sampler
// statistics stay in sampler_kernel for now).  The folded cross query (K = 4d hi|lo pairs of [x ; att]) is not drafted yet.
//
// NUMERICS CAVEAT: the GEMV path multiplies f32 activations by f16 weights; here the activations are rounded to f16 for the MFMA
// (relative error 2^-11 per element).  Emulated on the CPU oracle (tests/estimate_f16_activation_error.py) that costs 1.7e-3 in the logits
// after 12 layers (whisper-small shape) - above the 1e-3 bar of the parity tests - while an f16 hi | lo pair costs 1.6e-6.
// Planned remedy, to be decided by measurement: feed the activation as an f16 hi | lo pair (a = hi + lo, two MFMAs per K-step - the
// kernel stays a weight stream; LDS then holds [32][K/2] hi and lo per barrier-separated K half, 82 KB), exactly the trick the
// folded cross-query weights already use on the other operand.
//
// Wiring plan (round 2): launch_decoder_step (decoder.hip) picks this path when batch > 16:
//   grid = (N + 31) / 32 workgroups of 256 threads, dynamic LDS = dec_gemm32_lds_bytes(mode, K) (99 KB at K = 1280: set
//   hipFuncAttributeMaxDynamicSharedMemorySize once per instantiation, as mel.hip does), STEPS = d / 64;
//   QKV: N = 3d, K = d | out-proj / cross-out-proj: RESID_F32, N = K = d | FC1: N = 4d, K = d | FC2: RESID_F16, N = d, K = 4d
//   (four chunks per wave; double-buffer the chunks if the gaps show in the trace) | LOGITS: N = n_vocab, K = d, bias = null.
//   First validation: tests/test_gpu_parity.py at batch 32 against the existing GEMV path (batch-invariance tests compare B = 10
//   with single slots - extend them to 32), then tools/probe_decode.py for the per-kernel times.
#include "../kernels.h"

namespace wh {

enum { G32_FC1 = 0, G32_RESID_F32 = 1, G32_RESID_F16 = 2, G32_QKV = 3, G32_LOGITS = 4 };

struct Gemm32Args {
    int batch, N, K;              // batch <= 32, N % 32 == 0, K % 64 == 0
    const f16* W;                 // [N][K]
    const float* bias;            // [N]
    const float *ln_g, *ln_b;     // FC1: LayerNorm over K (= d) of x
    float* x;                     // residual stream [B][d]: FC1 input, RESID in/out
    const float* ain;             // RESID_F32 input [B][K]
    f16* hbuf;                    // FC1 output [B][N] / RESID_F16 input [B][K]
    const SeqState* seq;          // liveness of the slots, token_index (QKV cache position), next_token (layer-0 embedding)
    // QKV: q f32 [B][d], this layer's self-attention cache [Bmax][H][224][64] f16; layer 0 builds x = emb[token] + pos[position] first
    int d, n_head, n_vocab, layer;
    float* q; f16* self_k; f16* self_v;
    const f16* emb; const float* pos;
    float* logits;                // LOGITS: [B][N] f32 (N = n_vocab, last tile partial)
};

// f32 inputs (x with LayerNorm, att without) keep f32-level accuracy as an f16 hi | lo pair (a = hi + lo: two MFMAs per K-step).  They
// are staged per workgroup in LDS in ROUNDS = 2 K-rounds: round h holds, for each of the 4 waves, columns [w kq + h kq/2, + kq/2) of
// all 32 rows as hi and lo planes ([32][2 kq + 8] halves each: 2 planes x 32 x 648 x 2 B = 83 KB at K = 1280),
// read back as conflict-free ds_read_b128.  STEPS = K-steps of 16 per round (d = 1280: 10, 1024: 8, 768: 6, 384: 3); every weight
// load of the wave's K quarter (ROUNDS x STEPS x 16 B per lane) is issued before the first barrier, so the MFMA loop waits on nothing
// but the weight stream.  An f16 input (hbuf, K = 4d; already f16 in the GEMV path too) is read straight into fragments together
// with the weights, in chunks of ROUNDS x STEPS steps.
template <int MODE, int STEPS>
__global__ __launch_bounds__(256) void dec_gemm32_kernel(Gemm32Args a) {
    constexpr int ROUNDS = 2;
    extern __shared__ __align__(16) unsigned char s_dyn[];
    float (*s_red)[32][33] = reinterpret_cast<float (*)[32][33]>(s_dyn);                    // [4][n_local][b] partial tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kq = a.K >> 2;                    // K range of this wave: [wave * kq, (wave + 1) * kq)
    constexpr int RK = STEPS * 16;              // columns per wave and round
    constexpr int LDA = 4 * RK + 8;             // halves per staged row (4 waves' slices + 16 B pad)
    f16* s_hi = reinterpret_cast<f16*>(s_dyn + sizeof(float) * 4 * 32 * 33);                // [32][LDA]
    f16* s_lo = s_hi + 32 * LDA;

    f32x16 acc = {0};
    const f16* wrow = a.W + (size_t)min(n0 + b, a.N - 1) * a.K + (size_t)wave * kq + kh * 8;   // lane's W row (b doubles as n_local; rows past N are clamped, masked at the store)
    constexpr bool kLN = MODE == G32_FC1 || MODE == G32_QKV || MODE == G32_LOGITS;

    if (MODE != G32_RESID_F16) {
        // the wave's whole K quarter of weights goes in flight now (kq == ROUNDS * RK is a launch precondition)
        f16x8 wfrag[ROUNDS][STEPS];
#pragma unroll
        for (int h = 0; h < ROUNDS; ++h)
#pragma unroll
            for (int s = 0; s < STEPS; ++s) wfrag[h][s] = *reinterpret_cast<const f16x8*>(wrow + h * RK + s * 16);
        // ---- activations: 8 threads per batch row; LayerNorm statistics first (sum / sum of squares in f32)
        const int r = threadIdx.x >> 3, part = threadIdx.x & 7;
        const float* src = (kLN ? a.x : a.ain) + (size_t)r * a.K;
        const bool row_ok = r < a.batch;
        const bool embed = MODE == G32_QKV && a.layer == 0;     // x = token embedding + learned position (decoder.hip MODE_QKV prologue)
        int tok = 0, tpos = 0;
        if (embed && row_ok) { tok = min(max(a.seq[r].next_token, 0), a.n_vocab - 1); tpos = min(max(a.seq[r].token_index, 0), kMaxTok - 1); }
        auto load_x = [&](int i) -> float4 {
            if (!row_ok) return float4{0, 0, 0, 0};
            if (embed) {
                const f16x4 e = *reinterpret_cast<const f16x4*>(a.emb + (size_t)tok * a.K + 4 * i);
                const float4 pz = *reinterpret_cast<const float4*>(a.pos + (size_t)tpos * a.K + 4 * i);
                return float4{(float)e[0] + pz.x, (float)e[1] + pz.y, (float)e[2] + pz.z, (float)e[3] + pz.w};
            }
            return reinterpret_cast<const float4*>(src)[i];
        };
        float mean = 0.f, rstd = 1.f;
        if (kLN) {
            float s = 0.f, ss = 0.f;
            if (row_ok)
                for (int i = part; i < a.K / 4; i += 8) {
                    float4 v = load_x(i);
                    s += v.x + v.y + v.z + v.w; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    if (embed && blockIdx.x == 0) reinterpret_cast<float4*>(a.x + (size_t)r * a.K)[i] = v;     // the residual stream starts here
                }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
            mean = s / (float)a.K;
            rstd = rsqrtf(fmaxf(ss / (float)a.K - mean * mean, 0.f) + 1e-5f);
        }
#pragma unroll
        for (int h = 0; h < ROUNDS; ++h) {
            if (h) __syncthreads();                                           // everyone is done reading round h - 1
            // stage round h: float4 index j of the staged row <-> wave slice w = j / (RK / 4), column w kq + h RK + 4 (j % (RK / 4))
            for (int j = part; j < RK; j += 8) {                              // RK float4 per row (4 waves x RK / 4)
                const int w = j / (RK / 4), c4 = j - w * (RK / 4);
                const int col = w * kq + h * RK + 4 * c4;
                float4 v = load_x(col >> 2);
                if (kLN) {
                    const float4 g = *reinterpret_cast<const float4*>(a.ln_g + col), be = *reinterpret_cast<const float4*>(a.ln_b + col);
                    v.x = (v.x - mean) * rstd * g.x + be.x; v.y = (v.y - mean) * rstd * g.y + be.y;
                    v.z = (v.z - mean) * rstd * g.z + be.z; v.w = (v.w - mean) * rstd * g.w + be.w;
                    if (!row_ok) v = float4{0, 0, 0, 0};
                }
                const f16x4 hi = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                const f16x4 lo = {(f16)(v.x - (float)hi[0]), (f16)(v.y - (float)hi[1]), (f16)(v.z - (float)hi[2]), (f16)(v.w - (float)hi[3])};
                *reinterpret_cast<f16x4*>(s_hi + (size_t)r * LDA + 4 * j) = hi;
                *reinterpret_cast<f16x4*>(s_lo + (size_t)r * LDA + 4 * j) = lo;
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const size_t off = (size_t)b * LDA + wave * RK + s * 16 + kh * 8;
                const f16x8 bhi = *reinterpret_cast<const f16x8*>(s_hi + off), blo = *reinterpret_cast<const f16x8*>(s_lo + off);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[h][s], bhi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[h][s], blo, acc, 0, 0, 0);
            }
        }
    } else {
        constexpr int CH = ROUNDS * STEPS;                                    // K-steps per chunk
        const f16* arow = a.hbuf + (size_t)(b < a.batch ? b : 0) * a.K + (size_t)wave * kq + kh * 8;   // rows past the batch: any valid row, masked at the store
        for (int k0 = 0; k0 < kq; k0 += CH * 16) {
            f16x8 wfrag[CH], afrag[CH];
#pragma unroll
            for (int s = 0; s < CH; ++s) wfrag[s] = *reinterpret_cast<const f16x8*>(wrow + k0 + s * 16);    // the weight stream
#pragma unroll
            for (int s = 0; s < CH; ++s) afrag[s] = *reinterpret_cast<const f16x8*>(arow + k0 + s * 16);
            __builtin_amdgcn_sched_barrier(0);      // keep every load of the chunk ahead of its first MFMA (the scheduler would sink them to depth 2)
#pragma unroll
            for (int s = 0; s < CH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[s], afrag[s], acc, 0, 0, 0);
        }
    }

    // ---- reduce the 4 K-quarters: lane holds batch row b, W rows (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][(r & 3) + 8 * (r >> 2) + 4 * kh][b] = acc[r];
    __syncthreads();
    // thread t: batch row bb = t & 31, four consecutive output rows n0 + 4 * (t >> 5) .. + 3
    const int bb = threadIdx.x & 31, ng = (threadIdx.x >> 5) * 4;
    if (bb >= a.batch || !(a.seq[bb].active && !a.seq[bb].done)) return;
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        y[j] = ((s_red[0][ng + j][bb] + s_red[1][ng + j][bb]) + s_red[2][ng + j][bb]) + s_red[3][ng + j][bb] + (a.bias && n0 + ng + j < a.N ? a.bias[n0 + ng + j] : 0.f);
    if (MODE == G32_QKV) {            // decoder.hip MODE_QKV epilogue: q stays f32, k / v go to the cache row of this step
        const int n = n0 + ng, d = a.d;
        if (n < d) *reinterpret_cast<float4*>(a.q + (size_t)bb * d + n) = float4{y[0], y[1], y[2], y[3]};
        else {
            int c = n - d;
            f16* dst = a.self_k;
            if (c >= d) { c -= d; dst = a.self_v; }
            const int tp = min(max(a.seq[bb].token_index, 0), kMaxTok - 1);
            *reinterpret_cast<f16x4*>(dst + (((size_t)bb * a.n_head + (c >> 6)) * kMaxTok + tp) * kHeadDim + (c & 63)) = f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
        }
    } else if (MODE == G32_LOGITS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n0 + ng + j < a.N) a.logits[(size_t)bb * a.N + n0 + ng + j] = y[j];
    } else if (MODE == G32_FC1) {
        *reinterpret_cast<f16x4*>(a.hbuf + (size_t)bb * a.N + n0 + ng) = f16x4{(f16)gelu_erf(y[0]), (f16)gelu_erf(y[1]), (f16)gelu_erf(y[2]), (f16)gelu_erf(y[3])};
    } else {
        float4* xp = reinterpret_cast<float4*>(a.x + (size_t)bb * a.N + n0 + ng);
        float4 xo = *xp;
        *xp = float4{xo.x + y[0], xo.y + y[1], xo.z + y[2], xo.w + y[3]};
    }
}

// dynamic LDS: partial tiles + (f32-input modes) one round of staged hi | lo activations; > 64 KB needs
// hipFuncSetAttribute(MaxDynamicSharedMemorySize).  steps = STEPS of the instantiation = K / 128.
inline size_t dec_gemm32_lds_bytes(int mode, int steps) { return sizeof(float) * 4 * 32 * 33 + (mode == G32_RESID_F16 ? 0 : (size_t)2 * 32 * (4 * steps * 16 + 8) * 2); }

template __global__ void dec_gemm32_kernel<G32_FC1, 10>(Gemm32Args);
template __global__ void dec_gemm32_kernel<G32_RESID_F32, 10>(Gemm32Args);
template __global__ void dec_gemm32_kernel<G32_RESID_F16, 10>(Gemm32Args);
template __global__ void dec_gemm32_kernel<G32_QKV, 10>(Gemm32Args);
template __global__ void dec_gemm32_kernel<G32_LOGITS, 10>(Gemm32Args);
template __global__ void dec_gemm32_kernel<G32_FC1, 3>(Gemm32Args);

}  // namespace wh
