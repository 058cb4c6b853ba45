// Encoder self-attention (T = 1500, head_dim 64, non-causal) for gfx950, flash style: the 1500x1500
// score matrix never leaves registers.  One workgroup = 128 query rows of one (batch, head); each of
// the 4 waves owns 32 query rows and walks the keys in tiles of 64.
//
// Both MFMAs are issued "swapped" so that the query index stays on lane & 31 from the first
// instruction to the last (softmax statistics are then per-lane scalars, no cross-lane traffic
// except one xor-32 exchange per tile):
//     S^T[key][q]  = K[key][:] . Q[q][:]          A = K tile (LDS, row-major, 144-byte padded rows),  B = Q (registers)
//     O^T[c][q]   += V^T[c][key] * P^T[key][q]    A = V^T tile (LDS; V^T is written by the QKV GEMM epilogue),  B = P
// The key->k-slot permutation implied by the S^T accumulator layout is absorbed by reading V^T with the
// same permutation (two 8-byte LDS reads per fragment), so P goes from accumulator registers to the next
// MFMA's B operand with only an f32->f16 convert.  Query scale (1/8) is folded into W_q at load time.
// FLOPs: 4*T^2*d per layer and chunk -> MFMA-bound.
#include <cstdlib>

#include "kernels.h"

namespace wh {

constexpr int AT_LD = 72;  // LDS row stride in halves (64 + 8 pad = 144 B)

// __launch_bounds__(256, 2): two workgroups per CU = at least two waves per SIMD caps the wave at 256 unified registers, which makes
// the compiler keep the MFMA accumulators in arch VGPRs.  With one wave per SIMD allowed it placed them in AccVGPRs and moved the
// 64 score / output registers through 200 v_accvgpr_read / write per key tile - on a kernel that is VALU-bound (softmax) already.
__global__ __launch_bounds__(256, 2) void encoder_attention_kernel(const f16* __restrict__ q16, const f16* __restrict__ k16,
                                                                const f16* __restrict__ vt16, f16* __restrict__ out16,
                                                                int n_head, int d) {
    __shared__ __attribute__((aligned(16))) f16 Ks[64 * AT_LD];
    __shared__ __attribute__((aligned(16))) f16 Vs[64 * AT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int ql = lane & 31, half = lane >> 5;
    const int q_row = blockIdx.x * 128 + wave * 32 + ql;
    const int q_ld = q_row < kCtx ? q_row : kCtx - 1;

    f16x8 qf[4];
    {
        const f16* qp = q16 + ((size_t)b * kCtx + q_ld) * d + h * kHeadDim + 8 * half;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f16x8*>(qp + 16 * s);
    }
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
    float m_run = -1e30f, l_run = 0.0f;

    const f16* kbase = k16 + (size_t)b * kCtx * d + h * kHeadDim;
    const f16* vbase = vt16 + ((size_t)b * d + h * kHeadDim) * kCtxPad;

    for (int kv0 = 0; kv0 < kCtx; kv0 += 64) {
        // ---- stage K [64 keys][64] and V^T [64 c][64 keys]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int c_ = tid + 256 * i, row = c_ >> 3, cc = c_ & 7;
            int key = kv0 + row;
            uint4 kvv = key < kCtx ? *reinterpret_cast<const uint4*>(kbase + (size_t)key * d + cc * 8) : uint4{0, 0, 0, 0};
            *reinterpret_cast<uint4*>(&Ks[row * AT_LD + cc * 8]) = kvv;
            uint4 vv = *reinterpret_cast<const uint4*>(vbase + (size_t)row * kCtxPad + kv0 + cc * 8);
            *reinterpret_cast<uint4*>(&Vs[row * AT_LD + cc * 8]) = vv;
        }
        __syncthreads();

        // ---- S^T = K Q^T : two 32-key tiles
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8 kf = *reinterpret_cast<const f16x8*>(&Ks[(kt * 32 + ql) * AT_LD + 16 * st + 8 * half]);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], s[kt], 0, 0, 0);
            }
        }
        // ---- mask the key tail (last tile only)
        if (kv0 + 64 > kCtx) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= kCtx) s[kt][r] = -INFINITY;
                }
        }
        // ---- online softmax (q = lane & 31; the two lane halves hold disjoint keys of the same q)
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kt][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float psum = 0.0f;
        f16x8 pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __expf(s[u >> 1][8 * (u & 1) + j] - m_new);
                psum += p;
                pb[u][j] = (f16)p;
            }
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f16* vp = &Vs[(ct * 32 + ql) * AT_LD + 16 * u + 4 * half];
                f16x4 v0 = *reinterpret_cast<const f16x4*>(vp);
                f16x4 v1 = *reinterpret_cast<const f16x4*>(vp + 8);
                f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pb[u], o[ct], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < kCtx) {
        f16* op = out16 + ((size_t)b * kCtx + q_row) * d + h * kHeadDim;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 pk = {(f16)(o[ct][4 * g] * inv), (f16)(o[ct][4 * g + 1] * inv), (f16)(o[ct][4 * g + 2] * inv), (f16)(o[ct][4 * g + 3] * inv)};
                *reinterpret_cast<f16x4*>(op + 32 * ct + 8 * g + 4 * half) = pk;
            }
    }
}

// ---------------------------------------------------------------------------------------------- version 2 (round 3)
// The same arithmetic plan (swapped MFMAs, query on lane & 31), restructured around what the round-2 profile showed
// (profiles/r02z_pmc_sq2.csv: SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles, K / V re-read 4.6 x, VALU-bound softmax):
//   * K and V^T tiles double-buffered in LDS: the global loads of tile t + 2 are issued before the MFMAs of tile t, the LDS writes of
//     tile t + 1 follow them, ONE barrier per 64-key tile (version 1: load -> write -> barrier -> compute -> barrier);
//   * V^T rows padded to 136 bytes (34 dwords): the two 8-byte fragment reads of a lane group walk all 64 banks once (version 1's
//     144-byte rows put 32 rows on 16 bank pairs: 2-way); K rows stay at 144 bytes (conflict-free ds_read_b128);
//   * softmax in base 2 with the scale folded into one FMA per score (exp2(s * log2 e - m * log2 e): v_fma + v_exp instead of
//     v_sub + v_mul + v_exp), 3-input maxima, and the running-maximum rescale of O / l deferred while the maximum grows by less
//     than 8 (exp(8) = 2981 fits the f16 P operand with the same relative rounding): the 32 + multiplies per tile mostly vanish;
//   * workgroup id -> (batch, head, query tile) remapped so that the 12 query tiles of one (batch, head) run on ONE XCD and share
//     its K / V through that L2.
// Results differ from version 1 in the last bits (different but equally valid rounding points); the encoder tolerance is unchanged.
constexpr int AT_LDK = 72;   // K row stride in halves (144 B)
constexpr int AT_LDV = 68;   // V^T row stride in halves (136 B)
constexpr float kLog2e = 1.4426950408889634f;

__global__ __launch_bounds__(256, 2) void encoder_attention_v2_kernel(const f16* __restrict__ q16, const f16* __restrict__ k16,
                                                                   const f16* __restrict__ vt16, f16* __restrict__ out16,
                                                                   int n_head, int d, int n_pairs) {
    __shared__ __attribute__((aligned(16))) f16 Ks[2][64 * AT_LDK];
    __shared__ __attribute__((aligned(16))) f16 Vs[2][64 * AT_LDV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NQT = (kCtx + 127) / 128;
    // ids id, id + 8, id + 16, ... run on one XCD: give each XCD whole (batch, head) pairs, 12 query tiles back to back
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pair = (j / NQT) * 8 + xcd, qt = j % NQT;
    if (pair >= n_pairs) return;
    const int h = pair % n_head, b = pair / n_head;
    const int ql = lane & 31, half = lane >> 5;
    const int q_row = qt * 128 + wave * 32 + ql;
    const int q_ld = q_row < kCtx ? q_row : kCtx - 1;

    f16x8 qf[4];
    {
        const f16* qp = q16 + ((size_t)b * kCtx + q_ld) * d + h * kHeadDim + 8 * half;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f16x8*>(qp + 16 * s);
    }
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
    float m_run = -1e30f, l_run = 0.0f;      // m_run: the maximum the accumulators are scaled to (natural units)

    const f16* kbase = k16 + (size_t)b * kCtx * d + h * kHeadDim;
    const f16* vbase = vt16 + ((size_t)b * d + h * kHeadDim) * kCtxPad;
    // staging: thread -> (row, 16-byte chunk) of both tiles, two chunks each
    const int srow0 = tid >> 3, scc = tid & 7;
    uint4 kr[2], vr[2];
    auto gload = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i, key = kv0 + row;
            kr[i] = key < kCtx ? *reinterpret_cast<const uint4*>(kbase + (size_t)key * d + scc * 8) : uint4{0, 0, 0, 0};
            vr[i] = *reinterpret_cast<const uint4*>(vbase + (size_t)row * kCtxPad + kv0 + scc * 8);     // V^T rows are padded to kCtxPad keys (zeros)
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            *reinterpret_cast<uint4*>(&Ks[buf][row * AT_LDK + scc * 8]) = kr[i];
            uint2* vp = reinterpret_cast<uint2*>(&Vs[buf][row * AT_LDV + scc * 8]);      // 136-byte rows: 8-byte aligned only
            vp[0] = uint2{vr[i].x, vr[i].y};
            vp[1] = uint2{vr[i].z, vr[i].w};
        }
    };
    constexpr int NT = (kCtx + 63) / 64;
    gload(0);
    lstore(0);
    gload(64);
    __syncthreads();

    for (int t = 0; t < NT; ++t) {
        const int kv0 = t * 64, cur = t & 1;
        const f16* Kc = Ks[cur];
        const f16* Vc = Vs[cur];
        // ---- S^T = K Q^T : two 32-key tiles
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8 kf = *reinterpret_cast<const f16x8*>(&Kc[(kt * 32 + ql) * AT_LDK + 16 * st + 8 * half]);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], s[kt], 0, 0, 0);
            }
        }
        // the staged registers of tile t + 1 go to the other buffer (last read in iteration t - 1, a barrier ago), then tile t + 2 is requested
        if (t + 1 < NT) lstore(cur ^ 1);
        if (t + 2 < NT) gload(kv0 + 128);
        // ---- mask the key tail (last tile only)
        if (kv0 + 64 > kCtx) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= kCtx) s[kt][r] = -INFINITY;
                }
        }
        // ---- online softmax (q = lane & 31; the two lane halves hold disjoint keys of the same q)
        float mt = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mt = fmaxf(fmaxf(mt, s[0][r]), s[0][r + 1]);
        mt = fmaxf(mt, s[0][15]);
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[1][r]), s[1][r + 1]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (!__all(mt - m_run <= 8.0f)) {          // wave-uniform: rescale only when some query's maximum moved by more than 8
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        const float mb = -m_run * kLog2e;
        float psum = 0.0f;
        f16x8 pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[u >> 1][8 * (u & 1) + jj], kLog2e, mb));
                psum += p;
                pb[u][jj] = (f16)p;
            }
        }
        l_run += psum;
        // ---- O^T += V^T P^T
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f16* vp = &Vc[(ct * 32 + ql) * AT_LDV + 16 * u + 4 * half];
                f16x4 v0 = *reinterpret_cast<const f16x4*>(vp);
                f16x4 v1 = *reinterpret_cast<const f16x4*>(vp + 8);
                f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pb[u], o[ct], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < kCtx) {
        f16* op = out16 + ((size_t)b * kCtx + q_row) * d + h * kHeadDim;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 pk = {(f16)(o[ct][4 * g] * inv), (f16)(o[ct][4 * g + 1] * inv), (f16)(o[ct][4 * g + 2] * inv), (f16)(o[ct][4 * g + 3] * inv)};
                *reinterpret_cast<f16x4*>(op + 32 * ct + 8 * g + 4 * half) = pk;
            }
    }
}

void launch_encoder_attention(const f16* q16, const f16* k16, const f16* vt16, f16* out16, int batch, int n_head, int d, hipStream_t st) {
    ProfScope ps_(KK_ENC_ATTN, st);
    static const bool v1 = [] { const char* e = getenv("WH_ENC_ATTN_V1"); return e && e[0] == '1'; }();     // A/B knob: the round-2 kernel
    if (v1) {
        dim3 g((kCtx + 127) / 128, n_head, batch);
        encoder_attention_kernel<<<g, 256, 0, st>>>(q16, k16, vt16, out16, n_head, d);
        return;
    }
    const int n_pairs = n_head * batch, nqt = (kCtx + 127) / 128;
    encoder_attention_v2_kernel<<<(unsigned)(((n_pairs + 7) / 8) * 8 * nqt), 256, 0, st>>>(q16, k16, vt16, out16, n_head, d, n_pairs);
}

}  // namespace wh
