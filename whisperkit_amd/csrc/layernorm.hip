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

void launch_layernorm(const float* x, const float* g, const float* b, int rows, int d, f16* y16, float* y32, hipStream_t st) {
    ProfScope ps_(KK_LAYERNORM, st);
    layernorm_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, g, b, rows, d, y16, y32);
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
