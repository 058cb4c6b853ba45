// f16 MFMA GEMM for gfx950 with fused epilogues: C[M][N] = A[M][K] * W[N][K]^T (+bias).
//
// Both operands are K-contiguous, which is exactly the v_mfma_f32_32x32x16_f16 fragment shape
// (lane l holds 8 consecutive K of row l&31, K-half l>>5), so tiles go global -> registers -> LDS
// (80-byte padded rows: conflict-free ds_read_b128 for every lane group) -> fragments with no
// transposition anywhere.  256 threads = 4 waves (2x2), wave tile (BM/2)x(BN/2), BK = 32, two LDS
// stages with the next tile's global loads in flight under the current tile's MFMAs.
//
// Used for: conv1/conv2 as GEMMs over an overlapping-row view of the time-major input (K = 3*C_in),
// encoder QKV / out-proj / MLP, and the cross-attention K/V projection of all decoder layers.
// These are the encoder FLOPs of SURVEY.md section 8(d): MFMA-bound.
#include "kernels.h"

namespace wh {

constexpr int BK = 32;
constexpr int LDT = 40;  // LDS row stride in halves (32 + 8 pad)

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int CA = BM * 4 / 256, CB = BN * 4 / 256;   // 16-byte chunks per thread per tile
    __shared__ __attribute__((aligned(16))) f16 As[2][BM * LDT];
    __shared__ __attribute__((aligned(16))) f16 Bs[2][BN * LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // per-thread global row pointers (fixed across the K loop)
    const f16* a_ptr[CA];
    bool a_ok[CA];
    int a_lds[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        int c = tid + 256 * i, row = c >> 2, cc = c & 3;
        int m = m0 + row;
        a_ok[i] = m < a.M;
        int mm = a_ok[i] ? m : 0;
        long long off = (long long)(mm / a.a_rows_per_batch) * a.a_batch_stride + (long long)(mm % a.a_rows_per_batch) * a.lda;
        a_ptr[i] = a.A + off + cc * 8;
        a_lds[i] = row * LDT + cc * 8;
    }
    const f16* b_ptr[CB];
    bool b_ok[CB];
    int b_lds[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        int c = tid + 256 * i, row = c >> 2, cc = c & 3;
        int n = n0 + row;
        b_ok[i] = n < a.N;
        b_ptr[i] = a.W + (long long)(b_ok[i] ? n : 0) * a.K + cc * 8;
        b_lds[i] = row * LDT + cc * 8;
    }
    const int kc = (tid & 3) * 8;  // this thread's k offset inside a tile

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    uint4 ra[CA], rb[CB];
    auto gload = [&](int k0) {
        bool kin = (k0 + kc) < a.K;
#pragma unroll
        for (int i = 0; i < CA; ++i) ra[i] = (a_ok[i] && kin) ? *reinterpret_cast<const uint4*>(a_ptr[i] + k0) : uint4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < CB; ++i) rb[i] = (b_ok[i] && kin) ? *reinterpret_cast<const uint4*>(b_ptr[i] + k0) : uint4{0, 0, 0, 0};
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CA; ++i) *reinterpret_cast<uint4*>(&As[buf][a_lds[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < CB; ++i) *reinterpret_cast<uint4*>(&Bs[buf][b_lds[i]]) = rb[i];
    };

    const int nk = (a.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f16x8*>(&As[cur][(wm * WM + i * 32 + fr) * LDT + ks * 16 + fk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(&Bs[cur][(wn * WN + j * 32 + fr) * LDT + ks * 16 + fk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C layout (32x32): col n = lane & 31, row m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        if (n >= a.N) continue;
        const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {        // 4 groups of 4 consecutive rows
                const int mg = mb + 8 * g;
                if (mg >= a.M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][g * 4 + r] + bias;
                if constexpr (EPI == EPI_QKV_ENC) {
                    const int d = a.d_model;
                    if (n >= 2 * d) {   // V^T[(b*H + h)*64 + c][t], 4 consecutive t -> one 8-byte store
                        int call = n - 2 * d;
                        int bb = mg / kCtx, t = mg - bb * kCtx;
                        f16x4 pk = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        *reinterpret_cast<f16x4*>(a.vt16 + ((size_t)bb * d + call) * kCtxPad + t) = pk;
                        continue;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mg + r;
                    if (m >= a.M) continue;
                    const float x = v[r];
                    if constexpr (EPI == EPI_F16) {
                        a.out16[(size_t)m * a.ldc + n] = (f16)x;
                    } else if constexpr (EPI == EPI_GELU_F16) {
                        a.out16[(size_t)m * a.ldc + n] = (f16)gelu_erf(x);
                    } else if constexpr (EPI == EPI_RESID_F32) {
                        a.out32[(size_t)m * a.ldc + n] += x;
                    } else if constexpr (EPI == EPI_F32) {
                        a.out32[(size_t)m * a.ldc + n] = x;
                    } else if constexpr (EPI == EPI_QKV_ENC) {
                        const int d = a.d_model;
                        if (n < d) a.out16[(size_t)m * d + n] = (f16)x;
                        else a.k16[(size_t)m * d + (n - d)] = (f16)x;
                    } else if constexpr (EPI == EPI_CROSS_KV) {
                        const int d = a.d_model, H = d >> 6;
                        const int l = n / (2 * d), rem = n - l * 2 * d, kv = rem >= d, hc = rem - kv * d;
                        const int bb = m / kCtx, t = m - bb * kCtx;
                        f16* dst = kv ? a.vt16 : a.k16;
                        dst[((((size_t)l * a.max_batch + bb) * H + (hc >> 6)) * kCtx + t) * kHeadDim + (hc & 63)] = (f16)x;
                    } else if constexpr (EPI == EPI_CONV1) {
                        int bb = m / a.rows_per_batch_out, t = m - bb * a.rows_per_batch_out;
                        a.out16[((size_t)bb * kFramesPad + t + 1) * a.ldc + n] = (f16)gelu_erf(x);
                    } else if constexpr (EPI == EPI_CONV2) {
                        int t = m % a.rows_per_batch_out;
                        a.out32[(size_t)m * a.ldc + n] = gelu_erf(x) + a.pos[(size_t)t * a.ldc + n];
                    }
                }
            }
        }
    }
}

template <int EPI>
static void launch_epi(const GemmArgs& a, hipStream_t st) {
    // small problems get 64x64 tiles so that more than a handful of CUs are busy
    long long tiles128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (tiles128 >= 192) {
        dim3 g((a.N + 127) / 128, (a.M + 127) / 128);
        gemm_kernel<128, 128, EPI><<<g, 256, 0, st>>>(a);
    } else {
        dim3 g((a.N + 63) / 64, (a.M + 63) / 64);
        gemm_kernel<64, 64, EPI><<<g, 256, 0, st>>>(a);
    }
}

void launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t st) {
    ProfScope ps_(a.prof_kind, st);
    switch (epi) {
        case EPI_F16: launch_epi<EPI_F16>(a, st); break;
        case EPI_GELU_F16: launch_epi<EPI_GELU_F16>(a, st); break;
        case EPI_RESID_F32: launch_epi<EPI_RESID_F32>(a, st); break;
        case EPI_QKV_ENC: launch_epi<EPI_QKV_ENC>(a, st); break;
        case EPI_CONV1: launch_epi<EPI_CONV1>(a, st); break;
        case EPI_CONV2: launch_epi<EPI_CONV2>(a, st); break;
        case EPI_F32: launch_epi<EPI_F32>(a, st); break;
        case EPI_CROSS_KV: launch_epi<EPI_CROSS_KV>(a, st); break;
    }
}

}  // namespace wh
