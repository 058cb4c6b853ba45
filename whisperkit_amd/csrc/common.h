// Shared device/host helpers for the gfx950 Whisper kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

namespace wh {

constexpr int kWindowSamples = 480000;
constexpr int kFrames = 3000;     // mel frames per window
constexpr int kFramesPad = 3002;  // time-major mel / conv1 output carry one zero row before and after (conv padding=1)
constexpr int kCtx = 1500;        // encoder positions
constexpr int kCtxPad = 1536;     // V^T rows padded to a multiple of 64 keys
constexpr int kMaxTok = 224;      // decoder positions (Constants.maxTokenContext)
constexpr int kHeadDim = 64;
constexpr int kNFFT = 400;
constexpr int kHop = 160;
constexpr int kBins = 201;
constexpr int kBinsPad = 208;     // 13 MFMA column tiles of 16

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-form GELU (activation_function="gelu" in Whisper): x * Phi(x) with erfc(|x|) from Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 on erf; measured |gelu error| <= 4e-7 over [-8, 8], i.e. far below the fp16 rounding of the result).
// libdevice erff costs ~3x the instructions and is the dominant VALU cost of the fc1 epilogues.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = p * t * __expf(-ax * ax);          // erfc(|x| / sqrt 2)
    return 0.5f * x * (x >= 0.0f ? 2.0f - q : q);
}

// The same formula with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the correctly rounded division: 16 instead of 26
// VALU instructions per element.  Used by the encoder GEMM epilogues, where GELU over the 1500 x 4 d_model fc1 outputs of a chunk is
// VALU time with the matrix cores idle (0.31 ms of a 1.6 ms large-v3 fc1 launch at 64 chunks); the result differs from gelu_erf by
// <= 1 ulp of t, far below the f16 rounding of the stored activation.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = p * t * __expf(-ax * ax);
    return 0.5f * x * (x >= 0.0f ? 2.0f - q : q);
}

// monotone float <-> uint key for atomicMax on floats of either sign
__device__ __forceinline__ unsigned float_key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

}  // namespace wh
