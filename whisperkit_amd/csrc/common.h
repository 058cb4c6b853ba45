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

// Cross-lane sums / maxima without the LDS crossbar (`__shfl_xor` is ds_bpermute_b32: an LDS-pipe instruction with its latency and an
// address VGPR): DPP modifiers fold the exchanges inside a row of 16 lanes into the VALU instruction itself, the two exchanges
// across rows are gfx950's v_permlane16_swap / v_permlane32_swap.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;            // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int kDppHalfMirror = 0x141, kDppMirror = 0x140;  // lane i <-> 7 - i inside 8 lanes / i <-> 15 - i inside a row
constexpr int kDppRor8 = 0x128;                            // lane i <- lane (i + 8) % 16 of its row = lane i ^ 8
// sum / max over the 8 lanes of an aligned group, result in every lane of the group
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);       // the quads are uniform by now: any lane of the other quad will do
    return v;
}
// after the 8-lane groups are uniform: rows, then the four rows of the wave (the swap of a register with itself leaves
// {rows 0 0 2 2, rows 1 1 3 3} resp. {lanes 0-31 twice, lanes 32-63 twice}: their sum / max is the butterfly step)
__device__ __forceinline__ float rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = group8_sum(v);
    v += dpp_mov<kDppMirror>(v);
    return rows_sum(v);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_mov<kDppXor1>(v));
    v = fmaxf(v, dpp_mov<kDppXor2>(v));
    v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_mov<kDppMirror>(v));
    return rows_max(v);
}
// sum over the 8 lanes that share (lane & 7): l, l ^ 8, l ^ 16, ... - result in all of them
__device__ __forceinline__ float stride8_sum(float v) {
    v += dpp_mov<kDppRor8>(v);
    return rows_sum(v);
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
