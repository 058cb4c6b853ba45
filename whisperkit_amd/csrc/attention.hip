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

void launch_encoder_attention(const f16* q16, const f16* k16, const f16* vt16, f16* out16, int batch, int n_head, int d, hipStream_t st) {
    dim3 g((kCtx + 127) / 128, n_head, batch);
    ProfScope ps_(KK_ENC_ATTN, st);
    encoder_attention_kernel<<<g, 256, 0, st>>>(q16, k16, vt16, out16, n_head, d);
}

}  // namespace wh
