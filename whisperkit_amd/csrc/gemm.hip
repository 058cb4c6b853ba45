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
//
// gemm256_kernel (below) is the large-problem path; since round 5 its row-major epilogues turn the tile through the dead LDS stages
// so that store instructions write whole 128-byte lines (epi_stage.h; bit-identical outputs, encoder 2.79 -> 2.57 ms per chunk).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "epi_stage.h"
#include "kernels.h"

namespace wh {

constexpr int BK = 32;
constexpr int LDT = 40;  // LDS row stride in halves (32 + 8 pad)

// Fused epilogue of one wave's TM x TN accumulator tiles.  C layout of v_mfma_f32_32x32x16: col n = lane & 31,
// row m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  m_wave / n_wave = first row / column of the wave's sub-tile.
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[TM][TN], int m_wave, int n_wave, int lane) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_wave + j * 32 + (lane & 31);
        if (n >= a.N) continue;
        const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m_wave + i * 32 + 4 * (lane >> 5);
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
                        if (t + 3 < kCtx) {
                            f16x4 pk = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                            *reinterpret_cast<f16x4*>(a.vt16 + ((size_t)bb * d + call) * kCtxPad + t) = pk;
                        } else {        // the 4-row group straddles a slot boundary (1500 is a multiple of 4, so this is never taken)
                            for (int r = 0; r < 4; ++r) {
                                int m = mg + r;
                                if (m < a.M) { int b2 = m / kCtx, t2 = m - b2 * kCtx; a.vt16[((size_t)b2 * d + call) * kCtxPad + t2] = (f16)v[r]; }
                            }
                        }
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
                        a.out16[(size_t)m * a.ldc + n] = (f16)gelu_erf_fast(x);
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
                        const size_t o = ((((size_t)l * a.max_batch + bb) * H + (hc >> 6)) * kCtx + t) * kHeadDim + (hc & 63);
                        f16 hi_; signed char lo_;                // 24-bit rows (round 5, kernels.h hr24): Float16 keys under a sharp softmax cost 7e-3 sigma of the logits
                        hr24_encode(x, hi_, lo_);
                        (kv ? a.kv_v_hi : a.kv_k_hi)[o] = hi_;
                        (kv ? a.kv_v_lo : a.kv_k_lo)[o] = lo_;
                    } else if constexpr (EPI == EPI_CONV1) {
                        int bb = m / a.rows_per_batch_out, t = m - bb * a.rows_per_batch_out;
                        a.out16[((size_t)bb * kFramesPad + t + 1) * a.ldc + n] = (f16)gelu_erf_fast(x);
                    } else if constexpr (EPI == EPI_CONV2) {
                        int t = m % a.rows_per_batch_out;
                        a.out32[(size_t)m * a.ldc + n] = gelu_erf_fast(x) + a.pos[(size_t)t * a.ldc + n];
                    }
                }
            }
        }
    }
}

// The bias of a wave's 64 columns in the swapped accumulator layout (columns j * 32 + 8 g + 4 (lane >> 5) .. + 3), fetched in ONE batch.
// The direct epilogues fetch each float4 where it is used, behind a branch on a.bias: every fetch is followed by s_waitcnt vmcnt(0),
// which on this ISA also waits for every store issued before it - 32 serial round trips per wave.
__device__ __forceinline__ void load_tile_bias(const GemmArgs& a, float4 (&bias)[2][4], int nw, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[j][g] = float4{0, 0, 0, 0};
    if (a.bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[j][g] = *reinterpret_cast<const float4*>(a.bias + nw + j * 32 + 8 * g + 4 * (lane >> 5));
    }
}

// Epilogue for accumulators produced with the operands swapped (mfma(W fragment, A fragment)): the 32 x 32 tile is C^T,
// so a lane owns ONE output row m = lane & 31 and, per register group, 4 CONSECUTIVE columns n - row-major outputs go
// out as 8-byte (f16x4) / 16-byte (float4) accesses instead of 2- and 4-byte ones.
template <int EPI, int TM, int TN, bool BATCH_BIAS = false>
__device__ __forceinline__ void gemm_epilogue_swapped(const GemmArgs& a, f32x16 (&acc)[TM][TN], int m_wave, int n_wave, int lane) {
    float4 tile_bias[2][4];
    if constexpr (BATCH_BIAS) {       // gemm256_kernel, N % 64 == 0: the wave's 64 columns are inside or outside as a whole
        static_assert(TM == 4 && TN == 2, "wave tile of gemm256_kernel");
        if (n_wave >= a.N) return;
        load_tile_bias(a, tile_bias, n_wave, lane);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_wave + i * 32 + (lane & 31);
        if (m >= a.M) continue;
        int bb = 0, t = m;
        if constexpr (EPI == EPI_CROSS_KV) { bb = m / kCtx; t = m - bb * kCtx; }
        if constexpr (EPI == EPI_CONV1 || EPI == EPI_CONV2) { bb = m / a.rows_per_batch_out; t = m - bb * a.rows_per_batch_out; }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_wave + j * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= a.N) continue;
                float4 bias = float4{0, 0, 0, 0};
                if constexpr (BATCH_BIAS) bias = tile_bias[j][g];
                else if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + n);
                float v0 = acc[i][j][4 * g] + bias.x, v1 = acc[i][j][4 * g + 1] + bias.y;
                float v2 = acc[i][j][4 * g + 2] + bias.z, v3 = acc[i][j][4 * g + 3] + bias.w;
                if constexpr (EPI == EPI_F16) {
                    *reinterpret_cast<f16x4*>(a.out16 + (size_t)m * a.ldc + n) = f16x4{(f16)v0, (f16)v1, (f16)v2, (f16)v3};
                } else if constexpr (EPI == EPI_GELU_F16) {
                    *reinterpret_cast<f16x4*>(a.out16 + (size_t)m * a.ldc + n) =
                        f16x4{(f16)gelu_erf_fast(v0), (f16)gelu_erf_fast(v1), (f16)gelu_erf_fast(v2), (f16)gelu_erf_fast(v3)};
                } else if constexpr (EPI == EPI_RESID_F32) {
                    float4* p = reinterpret_cast<float4*>(a.out32 + (size_t)m * a.ldc + n);
                    float4 o = *p;
                    o.x += v0; o.y += v1; o.z += v2; o.w += v3;
                    *p = o;
                } else if constexpr (EPI == EPI_F32) {
                    *reinterpret_cast<float4*>(a.out32 + (size_t)m * a.ldc + n) = float4{v0, v1, v2, v3};
                } else if constexpr (EPI == EPI_QKV_ENC) {   // q / k columns only (the V^T tiles use the unswapped order)
                    const int d = a.d_model;
                    f16* dst = n < d ? a.out16 + (size_t)m * d + n : a.k16 + (size_t)m * d + (n - d);
                    *reinterpret_cast<f16x4*>(dst) = f16x4{(f16)v0, (f16)v1, (f16)v2, (f16)v3};
                } else if constexpr (EPI == EPI_CROSS_KV) {
                    const int d = a.d_model, H = d >> 6;
                    const int l = n / (2 * d), rem = n - l * 2 * d, kv = rem >= d, hc = rem - kv * d;
                    const size_t o = ((((size_t)l * a.max_batch + bb) * H + (hc >> 6)) * kCtx + t) * kHeadDim + (hc & 63);
                    f16x4 hi4; char4 lo4;
                    { f16 h_; signed char l_; hr24_encode(v0, h_, l_); hi4[0] = h_; lo4.x = l_; hr24_encode(v1, h_, l_); hi4[1] = h_; lo4.y = l_;
                      hr24_encode(v2, h_, l_); hi4[2] = h_; lo4.z = l_; hr24_encode(v3, h_, l_); hi4[3] = h_; lo4.w = l_; }
                    *reinterpret_cast<f16x4*>((kv ? a.kv_v_hi : a.kv_k_hi) + o) = hi4;
                    *reinterpret_cast<char4*>((kv ? a.kv_v_lo : a.kv_k_lo) + o) = lo4;
                } else if constexpr (EPI == EPI_CONV1) {
                    *reinterpret_cast<f16x4*>(a.out16 + ((size_t)bb * kFramesPad + t + 1) * a.ldc + n) =
                        f16x4{(f16)gelu_erf_fast(v0), (f16)gelu_erf_fast(v1), (f16)gelu_erf_fast(v2), (f16)gelu_erf_fast(v3)};
                } else if constexpr (EPI == EPI_CONV2) {
                    const float4 ps = *reinterpret_cast<const float4*>(a.pos + (size_t)t * a.ldc + n);
                    *reinterpret_cast<float4*>(a.out32 + (size_t)m * a.ldc + n) =
                        float4{gelu_erf_fast(v0) + ps.x, gelu_erf_fast(v1) + ps.y, gelu_erf_fast(v2) + ps.z, gelu_erf_fast(v3) + ps.w};
                }
            }
        }
    }
}

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs a) {     // (256, 2): accumulators stay in arch VGPRs (no AccVGPR copies)
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
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    // Operand order per wave, the same rule as gemm256_kernel (so that both kernels produce bit-identical results and a
    // chunk encodes the same alone or in a large batch): swapped (lane owns 4 consecutive columns) except for the V^T columns
    // of the encoder QKV projection.  Both paths execute the same number of barriers.
    auto body = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        gload(0);
        lstore(0);
        __syncthreads();
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
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
            }
            if (kt + 1 < nk) lstore(cur ^ 1);
            __syncthreads();
        }
        if constexpr (SWAP) gemm_epilogue_swapped<EPI, TM, TN>(a, acc, m0 + wm * WM, n0 + wn * WN, lane);
        else gemm_epilogue<EPI, TM, TN>(a, acc, m0 + wm * WM, n0 + wn * WN, lane);
    };
    if constexpr (EPI == EPI_QKV_ENC) {
        if (n0 + wn * WN >= 2 * a.d_model) body(std::false_type{});
        else body(std::true_type{});
    } else {
        body(std::true_type{});
    }
}

// ---------------------------------------------------------------------------------------------- LDS-staged epilogues of gemm256_kernel
// (round 5; index maps and the reason in epi_stage.h, replayed on the CPU by tests/native/epi_stage_check.cpp).  A wave's 128 x 64 tile
// goes through its private 16 KB slice `wl` of the dead operand stages in passes; the arithmetic (bias add, GELU, conversion, the
// residual's x + v) is that of the direct epilogues above, term for term: the outputs are the same bits, only the store instructions
// differ - 16 bytes per lane with 8 (f16) / 16 (fp32) neighbouring lanes on one row instead of 8 bytes per lane on 32 different rows.
// Requires N % 64 == 0 (a wave's 64 columns are inside or outside as a whole), M % 4 == 0, 16-byte aligned rows (launch_epi checks).
__device__ __forceinline__ void wave_lds_turn() {      // LDS operations of ONE wave execute in order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int EPI>
__device__ __forceinline__ void epi_staged_f16(const GemmArgs& a, f32x16 (&acc)[4][2], unsigned char* wl, int mw, int nw, int lane) {
    if (nw >= a.N) return;
    f16* base = a.out16;
    int ld = a.ldc, nc = nw;
    if constexpr (EPI == EPI_QKV_ENC) {      // q / k columns (the V^T tiles take epi_staged_vt)
        ld = a.d_model;
        if (nw >= ld) { base = a.k16; nc = nw - ld; }
    }
    float4 bias[2][4];
    load_tile_bias(a, bias, nw, lane);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& c = acc[2 * hh + i2][j];
                    const float v0 = c[4 * g] + bias[j][g].x, v1 = c[4 * g + 1] + bias[j][g].y, v2 = c[4 * g + 2] + bias[j][g].z, v3 = c[4 * g + 3] + bias[j][g].w;
                    f16x4 pk;
                    if constexpr (EPI == EPI_GELU_F16) pk = f16x4{(f16)gelu_erf_fast(v0), (f16)gelu_erf_fast(v1), (f16)gelu_erf_fast(v2), (f16)gelu_erf_fast(v3)};
                    else pk = f16x4{(f16)v0, (f16)v1, (f16)v2, (f16)v3};
                    *reinterpret_cast<f16x4*>(wl + epi::f16_write_off(lane, i2, j, g)) = pk;
                }
        wave_lds_turn();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = mw + hh * 64 + epi::f16_read_row(lane, it);
            const uint4 v = *reinterpret_cast<const uint4*>(wl + epi::f16_read_off(lane, it));
            if (m < a.M) *reinterpret_cast<uint4*>(base + (size_t)m * ld + nc + epi::f16_read_col(lane)) = v;
        }
        wave_lds_turn();
    }
}

__device__ __forceinline__ void epi_staged_resid(const GemmArgs& a, f32x16 (&acc)[4][2], unsigned char* wl, int mw, int nw, int lane) {
    if (nw >= a.N) return;
    float4 bias[2][4];
    load_tile_bias(a, bias, nw, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m_first = mw + i * 32 + epi::f32_read_row(lane, 0);
        float* const col0 = a.out32 + nw + epi::f32_read_col(lane);
        float4 old[8];      // the residual rows of this pass, in flight under the LDS turn (rows past M: clamped loads, no store)
#pragma unroll
        for (int it = 0; it < 8; ++it) old[it] = *reinterpret_cast<const float4*>(col0 + (size_t)min(m_first + it * 4, a.M - 1) * a.ldc);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& c = acc[i][j];
                *reinterpret_cast<float4*>(wl + epi::f32_write_off(lane, j, g)) =
                    float4{c[4 * g] + bias[j][g].x, c[4 * g + 1] + bias[j][g].y, c[4 * g + 2] + bias[j][g].z, c[4 * g + 3] + bias[j][g].w};
            }
        wave_lds_turn();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float4 v = *reinterpret_cast<const float4*>(wl + epi::f32_read_off(lane, it));
            float4 o = old[it];
            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            if (m_first + it * 4 < a.M) *reinterpret_cast<float4*>(col0 + (size_t)(m_first + it * 4) * a.ldc) = o;
        }
        wave_lds_turn();
    }
}

// V^T columns of the encoder QKV projection (unswapped accumulators): V^T[(b * d + c) * kCtxPad + t] is contiguous along the product's rows.
__device__ __forceinline__ void epi_staged_vt(const GemmArgs& a, f32x16 (&acc)[4][2], unsigned char* wl, int mw, int nw, int lane) {
    if (nw >= a.N) return;
    const int d = a.d_model;
    float bias2[2] = {0.0f, 0.0f};
    if (a.bias) { bias2[0] = a.bias[nw + (lane & 31)]; bias2[1] = a.bias[nw + 32 + (lane & 31)]; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float bias = bias2[j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& c = acc[i][j];
                *reinterpret_cast<f16x4*>(wl + epi::vt_write_off(lane, i, g)) =
                    f16x4{(f16)(c[4 * g] + bias), (f16)(c[4 * g + 1] + bias), (f16)(c[4 * g + 2] + bias), (f16)(c[4 * g + 3] + bias)};
            }
        wave_lds_turn();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const uint4 v = *reinterpret_cast<const uint4*>(wl + epi::vt_read_off(lane, it));
            const int call = nw - 2 * d + j * 32 + epi::vt_read_col(lane, it);
            const int m = mw + epi::vt_read_row(lane);
            // two groups of 4 rows: 1500 is a multiple of 4, so a group never straddles two windows; 8 rows may (1500 % 8 == 4)
            if (m < a.M) {
                const int bb = m / kCtx, t = m - bb * kCtx;
                *reinterpret_cast<uint2*>(a.vt16 + ((size_t)bb * d + call) * kCtxPad + t) = uint2{v.x, v.y};
            }
            if (m + 4 < a.M) {
                const int bb = (m + 4) / kCtx, t = m + 4 - bb * kCtx;
                *reinterpret_cast<uint2*>(a.vt16 + ((size_t)bb * d + call) * kCtxPad + t) = uint2{v.z, v.w};
            }
        }
        wave_lds_turn();
    }
}

template <int EPI>
constexpr bool kHasStagedEpilogue = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_RESID_F32 || EPI == EPI_QKV_ENC;

// ---------------------------------------------------------------------------------------------- 256 x 256 x 64 tile, ping-pong
// Large-problem path (encoder GEMMs at batch >= 2, cross-K/V projection): 8 waves (2 along M x 4 along N, wave tile 128 x 64 =
// 4 x 2 MFMA tiles), operands staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into two 64 KB
// stages.  An LDS row is the 128-byte K-slice of one operand row; the DMA image is lane-linear, so the bank swizzle (16-byte chunk
// c of row r lives in slot c ^ ((r >> 1) & 7): conflict-free ds_read_b128 for every lane group) is applied to the SOURCE address:
// a row's 8 chunks are fetched permuted inside the same 128-byte line, coalescing intact.  Workgroup ids are remapped so that the
// 8 XCDs (id % 8) each walk their own contiguous range of tiles, in 8 x 4 blocks (grouped walk below): the tiles an XCD works on
// concurrently share their A and weight panels through its L2.  The k-steps of a tile accumulate in ascending order into one
// accumulator per output tile, the same order as gemm_kernel: both kernels produce the same bits.
// The two wave groups of a workgroup (wm = 0 / 1: waves w and w + 4 share a SIMD) run half a K-tile apart.  A K-tile is two half-tiles of
// two 16-wide k-steps; a wave alternates X (12 ds_read_b128: the fragments of one half-tile into registers) and Y (its 16 MFMAs),
// one raw s_barrier per slot, and in every slot one group is in X while the other is in Y: the matrix pipe of a SIMD always has a
// wave's MFMAs queued while its partner fetches.  The LDS-DMA of a K-tile is issued among the MFMAs of a Y slot and waited for
// (the issuing wave's vmcnt) three slots later, a barrier before the first read.  Slots (barrier b_s ends slot s):
//     group 0:  X(t,0) = 4t      Y(t,0) = 4t+1 [+ DMA(t+1)]   X(t,1) = 4t+2   Y(t,1) = 4t+3 [vmcnt(0) before b]      (+ one barrier at the end)
//     group 1:  (one barrier first: slot 0, DMA(1))   X(t,0) = 4t+1   Y(t,0) = 4t+2   X(t,1) = 4t+3 [vmcnt(0) before b]   Y(t,1) = 4t+4 [+ DMA(t+2)]
// Hazards: tile t is read in slots 4t .. 4t+3 (every X ends with lgkmcnt(0) before its barrier); DMA(t+1) overwrites the stage of
// tile t-1 from slot 4t on (its last read was slot 4t-1) and every wave has waited for its pieces before b_{4t+3}; the first read
// of tile t+1 is slot 4t+4.  Both groups execute 4 nk + 1 barriers.
template <int EPI, int MODE>      // MODE 0: direct epilogues; 1: LDS-staged (kHasStagedEpilogue); 2: direct with the bias fetched in one batch
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmArgs a) {
    constexpr int TM = 4, TN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [2 stages][A 32 KB | B 32 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tiles_n = (a.N + 255) >> 8, tiles_m = (a.M + 255) >> 8;
    constexpr int GM = 8;
    const int gsz = GM * tiles_n, grp = wg / gsz, first_m = grp * GM;
    const int gm = min(tiles_m - first_m, GM), in_g = wg - grp * gsz;
    const int m0 = (first_m + in_g % gm) << 8, n0 = (in_g / gm) << 8;

    const int srow = tid >> 3;
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    const f16* src[8];      // pieces 0..3: A rows j*64 + srow, 4..7: W rows
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = min(m0 + j * 64 + srow, a.M - 1);
        src[j] = a.A + (long long)(m / a.a_rows_per_batch) * a.a_batch_stride + (long long)(m % a.a_rows_per_batch) * a.lda + chunk * 8;
        const int n = min(n0 + j * 64 + srow, a.N - 1);
        src[4 + j] = a.W + (long long)n * a.K + chunk * 8;
    }
    auto piece = [&](int p, int kt) {
        unsigned char* dst = smem + (kt & 1) * 65536 + wave * 1024 + (p >> 2) * 32768 + (p & 3) * 8192;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[p] + kt * 64),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 1) & 7;
    const int a_row_off = (wm * 128 + fr) * 128, b_row_off = 32768 + (wn * 64 + fr) * 128;
    const int nk = a.K >> 6;
#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_XBAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_VMWAIT() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
    auto body = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        f16x8 af[2][TM], bf[2][TN];
        auto X = [&](int kt, int h) {
            const unsigned char* sb = smem + (kt & 1) * 65536;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int slot = ((2 * (2 * h + s2) + fh) ^ swz) * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[s2][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 4096 + slot);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[s2][i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 4096 + slot);
            }
        };
        auto Y = [&](auto dma_tag, int dma_kt) {   // 16 MFMAs; DMA: the wave's 8 LDS-DMA pieces of K-tile dma_kt go out among them
            constexpr bool DMA = decltype(dma_tag)::value;
            __builtin_amdgcn_s_setprio(1);      // the matrix cluster outranks the partner wave's fetch slot
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s2][i], bf[s2][j], acc[i][j], 0, 0, 0);
                    }
                    if constexpr (DMA) { __builtin_amdgcn_sched_barrier(0); piece(s2 * 4 + i, dma_kt); __builtin_amdgcn_sched_barrier(0); }
                }
            __builtin_amdgcn_s_setprio(0);
        };
        constexpr std::true_type kDma{};
        constexpr std::false_type kNoDma{};
        // prologue: tile 0 (all waves), landed and visible
#pragma unroll
        for (int p = 0; p < 8; ++p) piece(p, 0);
        PP_VMWAIT();
        PP_BAR();
        if (wm == 0) {
            for (int t = 0; t + 1 < nk; ++t) {
                X(t, 0); PP_XBAR();
                Y(kDma, t + 1); PP_BAR();
                X(t, 1); PP_XBAR();
                Y(kNoDma, 0); PP_VMWAIT(); PP_BAR();
            }
            X(nk - 1, 0); PP_XBAR();
            Y(kNoDma, 0); PP_BAR();
            X(nk - 1, 1); PP_XBAR();
            Y(kNoDma, 0); PP_BAR();
            PP_BAR();
        } else {
            if (nk > 1) {
#pragma unroll
                for (int p = 0; p < 8; ++p) piece(p, 1);
            }
            PP_BAR();
            for (int t = 0; t + 2 < nk; ++t) {
                X(t, 0); PP_XBAR();
                Y(kNoDma, 0); PP_BAR();
                X(t, 1); PP_VMWAIT(); PP_XBAR();
                Y(kDma, t + 2); PP_BAR();
            }
            for (int t = max(nk - 2, 0); t < nk; ++t) {
                X(t, 0); PP_XBAR();
                Y(kNoDma, 0); PP_BAR();
                X(t, 1); PP_VMWAIT(); PP_XBAR();
                Y(kNoDma, 0); PP_BAR();
            }
        }
        // every wave is past its last fragment read and its last LDS-DMA wait here (both precede the final barrier): the stages are dead
        if constexpr (MODE == 1 && kHasStagedEpilogue<EPI>) {
            unsigned char* wl = smem + wave * epi::kWaveRegion;
            if constexpr (!SWAP) epi_staged_vt(a, acc, wl, m0 + wm * 128, n0 + wn * 64, lane);
            else if constexpr (EPI == EPI_RESID_F32) epi_staged_resid(a, acc, wl, m0 + wm * 128, n0 + wn * 64, lane);
            else epi_staged_f16<EPI>(a, acc, wl, m0 + wm * 128, n0 + wn * 64, lane);
        } else if constexpr (SWAP) gemm_epilogue_swapped<EPI, TM, TN, MODE == 2>(a, acc, m0 + wm * 128, n0 + wn * 64, lane);
        else gemm_epilogue<EPI, TM, TN>(a, acc, m0 + wm * 128, n0 + wn * 64, lane);
    };
#undef PP_BAR
#undef PP_XBAR
#undef PP_VMWAIT
    if constexpr (EPI == EPI_QKV_ENC) {
        if (n0 + wn * 64 >= 2 * a.d_model) body(std::false_type{});
        else body(std::true_type{});
    } else {
        body(std::true_type{});
    }
}

// (Round 6, built, measured and rejected, profiles/r06ae_*: gemm256w_kernel - the same tile, stages, LDS image, walk and epilogues with ONE wave per SIMD: 4 waves of 128 x 128
// wave tiles, the 256 accumulators of a lane in AGPRs, 128 KB instead of 192 KB of fragment reads per K-tile, the next k-step's 8 ds_read_b128 and the wave's 16 LDS-DMA
// pieces per K-tile in the issue gaps of its 64 MFMAs, one workgroup barrier per K-tile; 179 VGPRs + 256 AGPRs, no scratch, no accumulator copies in the loop.  Bit-identical
// in every epilogue mode (7 width / slot cases) and NOT faster: qkv 3512 -> 3631 us, fc1 5354 -> 5542, fc2 4846 -> 4800, out projection 1590 -> 1668 at 256 chunks.  Two main
// loops this different landing on the same time says the limit is not the loop: the same binary on all-zero operands runs the encoder in 2.05 instead of 2.55 ms per chunk
// (qkv - 25 %, fc1 - 21 %, fc2 - 12 %, attention - 29 %; profiles/r06af_encoder_zero_operand_probe.jsonl) - the chip clocks these kernels to its power budget
// (MI355X_MICROARCH.md "DVFS give-back").  Code in git history, commit "gemm256w_kernel".)

// ---------------------------------------------------------------------------------------------- persistent tile loop (round 6; opt-in: WH_GEMM_PERSIST=1)
// gemm256_kernel as a loop over tiles: one workgroup per CU walks its XCD's share of the grouped tile order, and the first K-tile of tile
// i + 1 is requested (LDS-DMA into stage 0, dead since the K loop's last barrier) BEFORE the epilogue of tile i runs, so the store
// acknowledgements of a tile, the workgroup launch, the address set-up and the first DMA's latency of the next one no longer sit between two
// tiles' matrix work.  The staged epilogues turn their passes through [64 KB, 160 KB) - stage 1 plus the 32 KB of the CU's LDS that the two
// stages leave free, 12 KB per wave (a pass needs 9 216 bytes) - so they never touch the stage the prefetch lands in; the next tile's K-tile 1
// goes into stage 1 only after the barrier that follows every wave's epilogue.  The arithmetic of a tile is gemm256_kernel's, instruction for
// instruction: the outputs are the same bits (encoder-output MD5s of 7 width / slot cases equal, profiles/r06d_*).
//
// MEASURED, AND NOT FASTER (profiles/r06d_encoder_time_persist*.jsonl, 128 chunks): encoder 2.538 -> 2.531 ms per chunk (qkv 1773 -> 1753, out
// projection 782 -> 812, fc1 2669 -> 2624, fc2 2342 -> 2366 us): the tile boundary was not what the K loop waits for.  The same kernel then served
// as the probe that says what is (profiles/r06e_*; `probe` argument, WH_GEMM_STAGGER): per 64-wide K-tile and CU the loop takes ~4 800 cycles
// for 2 048 cycles of matrix work; with the operand fetch removed from the loop (probe -1, garbage results) ~3 000 (the X / Y slot structure:
// fragment reads + barriers); with the fetch issued but never waited for (probe -2) ~4 000.  So of the 1 800 cycles the fetch costs, ~1 000 are
// interference of the LDS-DMA stream with the loop (the landing 64 KB per K-tile share the LDS with 192 KB of fragment reads) and ~800 are the
// wait itself; staggering the workgroups that share a panel by 0.05 .. 1 us (so that followers find the leader's lines resident in the L2) and
// re-shaping the XCD's resident tile block from 8 x 4 to 4 x 8, 2 x 16, 1 x 32 (m x n tiles: who shares what) change nothing (+- 1 %).  The
// next step for this GEMM is therefore fewer LDS bytes per matrix instruction (one wave per SIMD with 128 x 128 wave tiles and 512 registers, or
// weight fragments from global memory in fragment order), not its boundary.  Default stays one workgroup per tile.
constexpr int kPersistWaveRegion = 12288;      // 8 waves x 12 KB = [64 KB, 160 KB)
static_assert(64 * epi::kRow16 <= kPersistWaveRegion && 32 * epi::kRow32 <= kPersistWaveRegion && 32 * epi::kRowT <= kPersistWaveRegion, "a pass fits the wave's slice");
template <int EPI, int MODE>
__global__ __launch_bounds__(512) void gemm256p_kernel(const GemmArgs a, int n_tiles, int stagger, int GM) {
    constexpr int TM = 4, TN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [2 stages][A 32 KB | B 32 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // XCD x (= id % 8) owns the contiguous range [lo, lo + cnt) of the grouped tile order (the ranges of gemm256_kernel); its workgroups
    // (gridDim / 8 of them) take the range's tiles round robin, so an XCD's resident workgroups sit on neighbouring tiles at any time
    const int orig = blockIdx.x, xcd = orig & 7, q = n_tiles >> 3, r = n_tiles & 7;
    const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
    const int per_xcd = gridDim.x >> 3;
    int it = orig >> 3;
    if (it >= cnt) return;
    // A/B probe (WH_GEMM_STAGGER, default 0): the workgroups of an XCD that share a weight panel (local ids j .. j + 7) or an A panel (j, j + 8, ...) start
    // `stagger` x 64 cycles apart, so that a panel's lines are resident in the L2 when the followers ask for them instead of pending behind the leader's miss
    if (stagger > 0) {
        const int rank = (it & 7) * 4 + ((it >> 3) & 3);
        for (int i = 0; i < rank * stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const int tiles_n = (a.N + 255) >> 8, tiles_m = (a.M + 255) >> 8;
    int m0, n0;               // GM: m-tiles per group of the walk (GM x tiles_n tiles, m fastest): an XCD's 32 resident tiles are GM m-tiles x 32 / GM n-tiles
    auto tile_origin = [&](int wg, int& m0_, int& n0_) {
        const int gsz = GM * tiles_n, grp = wg / gsz, first_m = grp * GM;
        const int gm = min(tiles_m - first_m, GM), in_g = wg - grp * gsz;
        m0_ = (first_m + in_g % gm) << 8; n0_ = (in_g / gm) << 8;
    };
    const int srow = tid >> 3;
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    const f16* src[8];      // pieces 0..3: A rows j*64 + srow, 4..7: W rows
    auto set_src = [&](int m0_, int n0_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(m0_ + j * 64 + srow, a.M - 1);
            src[j] = a.A + (long long)(m / a.a_rows_per_batch) * a.a_batch_stride + (long long)(m % a.a_rows_per_batch) * a.lda + chunk * 8;
            const int n = min(n0_ + j * 64 + srow, a.N - 1);
            src[4 + j] = a.W + (long long)n * a.K + chunk * 8;
        }
    };
    tile_origin(lo + it, m0, n0);
    set_src(m0, n0);
    auto piece = [&](int p, int kt) {
        unsigned char* dst = smem + (kt & 1) * 65536 + wave * 1024 + (p >> 2) * 32768 + (p & 3) * 8192;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[p] + kt * 64),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 1) & 7;
    const int a_row_off = (wm * 128 + fr) * 128, b_row_off = 32768 + (wn * 64 + fr) * 128;
    const int nk = a.K >> 6;
#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_XBAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_VMWAIT() do { __builtin_amdgcn_sched_barrier(0); if (stagger != -2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)      /* (stagger -2: timing probe that never waits for its operands, garbage results) */
    bool have_next = false;
    int m0n = 0, n0n = 0;
    auto body = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        f16x8 af[2][TM], bf[2][TN];
        auto X = [&](int kt, int h) {
            const unsigned char* sb = smem + (kt & 1) * 65536;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int slot = ((2 * (2 * h + s2) + fh) ^ swz) * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[s2][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 4096 + slot);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[s2][i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 4096 + slot);
            }
        };
        auto Y = [&](auto dma_tag, int dma_kt) {   // 16 MFMAs; DMA: the wave's 8 LDS-DMA pieces of K-tile dma_kt go out among them
            constexpr bool DMA = decltype(dma_tag)::value;
            __builtin_amdgcn_s_setprio(1);      // the matrix cluster outranks the partner wave's fetch slot
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s2][i], bf[s2][j], acc[i][j], 0, 0, 0);
                    }
                    if constexpr (DMA) { __builtin_amdgcn_sched_barrier(0); if (stagger != -1) piece(s2 * 4 + i, dma_kt); __builtin_amdgcn_sched_barrier(0); }   // (stagger < 0: fetch-ablation probe, garbage results)
                }
            __builtin_amdgcn_s_setprio(0);
        };
        constexpr std::true_type kDma{};
        constexpr std::false_type kNoDma{};
        // prologue: K-tile 0 (all waves) - requested by the previous tile's tail (or before the loop) - landed and visible; the barrier also
        // ends every wave's epilogue of the previous tile: stage 1 is free for K-tile 1 from here on
        PP_VMWAIT();
        PP_BAR();
        if (wm == 0) {
            for (int t = 0; t + 1 < nk; ++t) {
                X(t, 0); PP_XBAR();
                Y(kDma, t + 1); PP_BAR();
                X(t, 1); PP_XBAR();
                Y(kNoDma, 0); PP_VMWAIT(); PP_BAR();
            }
            X(nk - 1, 0); PP_XBAR();
            Y(kNoDma, 0); PP_BAR();
            X(nk - 1, 1); PP_XBAR();
            Y(kNoDma, 0); PP_BAR();
            PP_BAR();
        } else {
            if (nk > 1) {
#pragma unroll
                for (int p = 0; p < 8; ++p) piece(p, 1);
            }
            PP_BAR();
            for (int t = 0; t + 2 < nk; ++t) {
                X(t, 0); PP_XBAR();
                Y(kNoDma, 0); PP_BAR();
                X(t, 1); PP_VMWAIT(); PP_XBAR();
                Y(kDma, t + 2); PP_BAR();
            }
            for (int t = max(nk - 2, 0); t < nk; ++t) {
                X(t, 0); PP_XBAR();
                Y(kNoDma, 0); PP_BAR();
                X(t, 1); PP_VMWAIT(); PP_XBAR();
                Y(kNoDma, 0); PP_BAR();
            }
        }
        // every wave is past its last fragment read and its last LDS-DMA wait here (both precede the final barrier): the stages are dead.
        // The next tile's first K-tile goes out now, under this tile's epilogue.
        const int m0c = m0, n0c = n0;
        {
            const int nx = it + per_xcd;
            have_next = nx < cnt;
            if (have_next) {
                tile_origin(lo + nx, m0n, n0n);
                set_src(m0n, n0n);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 8; ++p) piece(p, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (MODE == 1 && kHasStagedEpilogue<EPI>) {
            unsigned char* wl = smem + 65536 + wave * kPersistWaveRegion;
            if constexpr (!SWAP) epi_staged_vt(a, acc, wl, m0c + wm * 128, n0c + wn * 64, lane);
            else if constexpr (EPI == EPI_RESID_F32) epi_staged_resid(a, acc, wl, m0c + wm * 128, n0c + wn * 64, lane);
            else epi_staged_f16<EPI>(a, acc, wl, m0c + wm * 128, n0c + wn * 64, lane);
        } else if constexpr (SWAP) gemm_epilogue_swapped<EPI, TM, TN, MODE == 2>(a, acc, m0c + wm * 128, n0c + wn * 64, lane);
        else gemm_epilogue<EPI, TM, TN>(a, acc, m0c + wm * 128, n0c + wn * 64, lane);
    };
#undef PP_BAR
#undef PP_XBAR
#undef PP_VMWAIT
#pragma unroll
    for (int p = 0; p < 8; ++p) piece(p, 0);              // the first tile's first K-tile
    while (true) {
        if constexpr (EPI == EPI_QKV_ENC) {
            if (n0 + wn * 64 >= 2 * a.d_model) body(std::false_type{});
            else body(std::true_type{});
        } else {
            body(std::true_type{});
        }
        if (!have_next) break;                             // (workgroup-uniform)
        it += per_xcd; m0 = m0n; n0 = n0n;
    }
}

template <int EPI>
static void launch_epi(const GemmArgs& a, hipStream_t st) {
    // large problems: 256 x 256 x 64 LDS-DMA kernel (needs whole 64-wide K tiles and 16-byte aligned rows)
    const long long tiles256 = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    static const bool no256 = [] { const char* e = getenv("WH_NO_GEMM256"); return e && e[0] == '1'; }();
    if (!no256 && tiles256 >= 64 && a.K % 64 == 0 && a.lda % 8 == 0 && a.a_batch_stride % 8 == 0 && a.N % 4 == 0) {
        // measured and rejected (profiles/r02s_*): staggering the first-round workgroups by up to a tile time to spread the store
        // epilogues of the 256 CUs over each other's K loops - no change (1476 vs 1475 us, large-v3 fc1 at 64 chunks)
        static const int epi_mode = [] { const char* e = getenv("WH_GEMM_EPI_MODE"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1; }();
        // WH_GEMM_PERSIST=1: the persistent tile loop (gemm256p_kernel: one workgroup per CU, the next tile's first K-tile requested under the epilogue;
        // bit-identical, measured no faster: the comment above the kernel); default 0 = one workgroup per tile (rounds 2 - 5).
        // WH_GEMM_PERSIST_WGS: workgroups of the persistent grid (default = the CUs, a multiple of 8: a smaller grid confines the encoder to that many CUs)
        static const int persist = [] { const char* e = getenv("WH_GEMM_PERSIST"); return e ? atoi(e) : 0; }();
        static const int persist_wgs = [] { const char* e = getenv("WH_GEMM_PERSIST_WGS"); int v = e ? atoi(e) : 0; return v > 0 ? (v + 7) / 8 * 8 : 0; }();
        auto go = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            if (persist) {
                static PerDeviceOnce raised_p;
                static int cus = 256;
                raised_p.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<EPI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
                                  int dev = 0, n = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) cus = n / 8 * 8; });
                const int grid = (int)std::min<long long>(persist_wgs ? persist_wgs : cus, (tiles256 + 7) / 8 * 8);
                static const int stagger = [] { const char* e = getenv("WH_GEMM_STAGGER"); return e ? atoi(e) : 0; }();
                static const int gm_env = [] { const char* e = getenv("WH_GEMM_GM"); int v = e ? atoi(e) : 8; return v >= 1 && v <= 64 ? v : 8; }();
                gemm256p_kernel<EPI, MODE><<<(unsigned)grid, 512, 163840, st>>>(a, (int)tiles256, stagger, gm_env);
                return;
            }
            static PerDeviceOnce raised;
            raised.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); });
            gemm256_kernel<EPI, MODE><<<(unsigned)tiles256, 512, 131072, st>>>(a);
        };
        if constexpr (kHasStagedEpilogue<EPI>) {
            // the staged epilogues store uint4 / float4 vectors: every output base must be 16-byte aligned (hipMalloc bases + multiples of 8
            // elements today; an offset view handed in by a future caller falls back to the direct epilogue instead of faulting - ADVICE r05)
            const bool aligned = (((uintptr_t)a.out16 | (uintptr_t)a.out32 | (uintptr_t)a.k16 | (uintptr_t)a.vt16) & 15) == 0;
            const bool shape_ok = aligned && a.N % 64 == 0 && a.M % 4 == 0 && a.ldc % 8 == 0 && a.d_model % 64 == 0;
            if (shape_ok && epi_mode == 1) { go(std::integral_constant<int, 1>{}); return; }
            if (shape_ok && epi_mode == 2) { go(std::integral_constant<int, 2>{}); return; }
        }
        go(std::integral_constant<int, 0>{});
        return;
    }
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
