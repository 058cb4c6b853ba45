// Row LayerNorm (eps 1e-5) over the fp32 residual stream -> f16 GEMM operand (and/or f32 copy),
// plus small conversion kernels.  One wave per row, row cached in registers, two-pass variance
// (matches torch.nn.functional.layer_norm numerics).  HBM-bound: 4 B read + 2 B written per element.
#include "kernels.h"

namespace wh {

constexpr int LN_MAXE = 20;  // d <= 1280

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ bta, int rows, int d,
                                                        f16* __restrict__ y16, float* __restrict__ y32) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    float v[LN_MAXE];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
        int c = lane + 64 * i;
        v[i] = c < d ? xr[c] : 0.0f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
        int c = lane + 64 * i;
        float t = c < d ? v[i] - mean : 0.0f;
        q += t * t;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
        int c = lane + 64 * i;
        if (c < d) {
            float o = (v[i] - mean) * rstd * g[c] + bta[c];
            if (y16) y16[(size_t)row * d + c] = (f16)o;
            if (y32) y32[(size_t)row * d + c] = o;
        }
    }
}

// The same row LayerNorm with 16-byte loads and 8-byte Float16 stores (round 6): lane l owns channels 4 l .. 4 l + 3 of every 256-channel group, so a wave instruction
// moves 1 KB in and 512 B out instead of 256 B / 128 B (the dword / 2-byte form above: 642 us per 256 x 1500 rows of 1280 = 4.6 TB/s).  Same two-pass arithmetic; the
// lane's partial sums cover different channels than in the scalar form, so results agree to the last bits of the fp32 statistics, not bit for bit.
template <bool NT>
__global__ __launch_bounds__(256) void layernorm_v4_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ bta, int rows, int d,
                                                           f16* __restrict__ y16, float* __restrict__ y32) {
    constexpr int NV = LN_MAXE / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    float4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        v[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
        if (c < d) {
            if constexpr (NT) { const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + c)); v[i] = float4{t[0], t[1], t[2], t[3]}; }
            else v[i] = *reinterpret_cast<const float4*>(xr + c);
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < d) {
            const float t0 = v[i].x - mean, t1 = v[i].y - mean, t2 = v[i].z - mean, t3 = v[i].w - mean;
            q += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < d) {
            const float4 gg = *reinterpret_cast<const float4*>(g + c), bb = *reinterpret_cast<const float4*>(bta + c);
            const float o0 = (v[i].x - mean) * rstd * gg.x + bb.x, o1 = (v[i].y - mean) * rstd * gg.y + bb.y;
            const float o2 = (v[i].z - mean) * rstd * gg.z + bb.z, o3 = (v[i].w - mean) * rstd * gg.w + bb.w;
            if (y16) {
                const f16x4 o = {(f16)o0, (f16)o1, (f16)o2, (f16)o3};
                if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<f16x4*>(y16 + (size_t)row * d + c));
                else *reinterpret_cast<f16x4*>(y16 + (size_t)row * d + c) = o;
            }
            if (y32) *reinterpret_cast<float4*>(y32 + (size_t)row * d + c) = float4{o0, o1, o2, o3};
        }
    }
}

void launch_layernorm(const float* x, const float* g, const float* b, int rows, int d, f16* y16, float* y32, hipStream_t st) {
    ProfScope ps_(KK_LAYERNORM, st);
    static const int v4 = [] { const char* e = getenv("WH_LN_V4"); return e ? atoi(e) : 2; }();      // 2 (default): non-temporal row loads and Float16 stores (514 -> 488 us per 384 000 rows: each is touched once before 3 GB of other traffic), 1: plain, 0: the scalar form; 1 and 2 give the same bits
    const bool aligned = d % 4 == 0 && (((uintptr_t)x | (uintptr_t)g | (uintptr_t)b | (uintptr_t)y32) % 16) == 0 && ((uintptr_t)y16 % 8) == 0;
    if (v4 == 2 && aligned) layernorm_v4_kernel<true><<<(rows + 3) / 4, 256, 0, st>>>(x, g, b, rows, d, y16, y32);
    else if (v4 && aligned) layernorm_v4_kernel<false><<<(rows + 3) / 4, 256, 0, st>>>(x, g, b, rows, d, y16, y32);
    else layernorm_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, g, b, rows, d, y16, y32);
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, f16* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (f16)in[i];
}
__global__ void f16_to_f32_kernel(const f16* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
void launch_f32_to_f16(const float* in, f16* out, size_t n, hipStream_t st) {
    f32_to_f16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, out, n);
}
void launch_f16_to_f32(const f16* in, float* out, size_t n, hipStream_t st) {
    f16_to_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, out, n);
}

}  // namespace wh
