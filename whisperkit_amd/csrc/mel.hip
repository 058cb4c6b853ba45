// Log-mel spectrogram for gfx950: replaces the CoreML MelSpectrogram call of
// Sources/WhisperKit/Core/FeatureExtractor.swift:40-56 (+ padOrTrim, AudioProcessor.swift:151-174).
//
// Algorithm (openai/whisper audio.py): reflect pad 200, hann(400) STFT hop 160 -> |X|^2 (201 bins)
// -> slaney mel filterbank -> log10(max(.,1e-10)) -> max(., global_max - 8) -> (x + 4) / 4.
//
// Kernel 1 (mel_power_kernel): one workgroup = 64 frames (4 groups of 16 sharing every basis fragment).  The 400-point real DFT is evaluated as two
// K=200 real GEMMs on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 fma chain):
//     Re X[k] = sum_{n=1..200} (x[n] + x[400-n]) * w[n] cos(2 pi n k / 400)      (x[200] counted once)
//     Im X[k] = sum_{n=1..199} (x[n] - x[400-n]) * w[n] sin(2 pi n k / 400)
// (the periodic Hann window is symmetric about n = 200 and w[0] = 0, so the window folds into the
// basis and the even/odd fold halves the MFMA work).  The signal under the 64 frames (10480 samples) sits in LDS
// once, skewed for conflict-free fragment reads, and is folded on the fly; the windowed basis
// (2 x 200 x 208 f32 = 333 KB) streams from L2.  Power -> LDS ->
// sparse triangular mel filters -> log10 -> f32 scratch [n_mels][3000] + per-chunk atomic max.
// Kernel 2 (mel_finalize_kernel): clamp to max-8, scale, emit the time-major f16 [3002][n_mels]
// operand of the conv1 GEMM (and the reference-layout f32 [n_mels][3000] copy for the C ABI).
#include "common.h"
#include "kernels.h"

namespace wh {

constexpr int LDP = 209;   // power row stride

constexpr int FG = 4;      // 16-frame groups per workgroup: every basis fragment fetched from L2 feeds FG MFMAs
constexpr int kSpan = (16 * FG - 1) * kHop + kNFFT;    // 10480 samples: the reflect-padded signal under a workgroup's 64 frames
// sample j of the span sits at j + 2 (j / 160): frame i's sample n at i * 162 + n + 2 (n / 160), so the 16 frames an MFMA A-fragment
// read touches (same n, lanes ai = 0..15, k slots ak, ak + 1) fall on banks 2 ai + ak: conflict-free ds_read_b32
constexpr int kSpanLds = kSpan + 2 * (kSpan / kHop) + 2;

// __launch_bounds__(256, 2): two workgroups per CU (the LDS allows it since round 3) = at most 256 registers per wave, which keeps the 128
// accumulator registers in arch VGPRs (865 -> 634 us per 64 chunks; unrolling the k loop 5 / 10 deep instead of 2 is slower: profiles/r03r_*)
__global__ __launch_bounds__(256, 2) void mel_power_kernel(const float* __restrict__ pcm_all, const int* __restrict__ n_valid_all,
                                                        const float* __restrict__ basis_c, const float* __restrict__ basis_s,
                                                        const float* __restrict__ filt_c, const int* __restrict__ filt_off, int filt_nnz,
                                                        const int2* __restrict__ filt_range,
                                                        int n_mels, float* __restrict__ logspec, unsigned* __restrict__ maxkey) {
    extern __shared__ __attribute__((aligned(16))) float mel_smem[];
    float* xs = mel_smem;                        // [kSpanLds]  the signal under the 64 frames (skewed, see above)
    float* pw = xs + kSpanLds;                   // [16][LDP]   power of one frame group at a time
    __shared__ float red[4];
    __shared__ float fc_l[1024];                 // compact mel filter weights (a global-memory filter loop is a chain of
    __shared__ int2 rg_l[128];                   // dependent L2 round trips: it was the whole kernel time)
    __shared__ int fo_l[128];
    for (int i = threadIdx.x; i < filt_nnz; i += 256) fc_l[i] = filt_c[i];
    for (int i = threadIdx.x; i < n_mels; i += 256) { rg_l[i] = filt_range[i]; fo_l[i] = filt_off[i]; }
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * (16 * FG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pcm = pcm_all + (size_t)b * kWindowSamples;
    const int n_valid = n_valid_all[b];

    // The span in ONE round of independent, coalesced loads (round 2 folded x[n] +- x[400 - n] into two LDS images with two dependent
    // global loads per element: 50 latency-bound iterations per thread, 1.09 ms per 64 chunks = 0.2 TB/s; and 117 KB of LDS = one
    // workgroup per CU).  Index of span sample j in the window: reflect padding at both ends, zeros past n_valid (padOrTrim).
    {
        constexpr int NLD = (kSpan + 255) / 256;
        float v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            int p = f0 * kHop + tid + 256 * u - kNFFT / 2;
            if (p < 0) p = -p;
            if (p >= kWindowSamples) p = 2 * (kWindowSamples - 1) - p;
            p = min(max(p, 0), kWindowSamples - 1);                      // (only the unused tail of the last span gets here)
            const float x = pcm[p];
            v[u] = p < n_valid ? x : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int j = tid + 256 * u;
            if (j < kSpan) xs[j + 2 * (j / kHop)] = v[u];
        }
    }
    __syncthreads();

    f32x4 acc_re[FG][4], acc_im[FG][4];
#pragma unroll
    for (int g = 0; g < FG; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc_re[g][t] = f32x4{0, 0, 0, 0}; acc_im[g][t] = f32x4{0, 0, 0, 0}; }
    const int ai = lane & 15, ak = lane >> 4;
    // wave w owns bin tiles w, w+4, w+8, w+12 (13 tiles of 16 bins)
#pragma unroll 2
    for (int ks = 0; ks < 50; ++ks) {
        const int k = ks * 4 + ak, n = k + 1;
        // fold on the fly: a_e = x[n] + x[400 - n], a_o = x[n] - x[400 - n] (x[200] counted once) - the same two LDS reads per
        // operand pair the folded images cost, the same values
        const int o1 = n + 2 * (n >= kHop), o2 = (kNFFT - n) + 2 * (1 + ((kNFFT - n) >= 2 * kHop));
        const bool mid = n == kNFFT / 2;
        float a_e[FG], a_o[FG];
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            const int rowoff = (g * 16 + ai) * (kHop + 2);
            const float x1 = xs[rowoff + o1], x2 = xs[rowoff + o2];
            a_e[g] = mid ? x1 : x1 + x2;
            a_o[g] = mid ? 0.0f : x1 - x2;
        }
        const float* bc = basis_c + (size_t)k * kBinsPad + ai;
        const float* bs = basis_s + (size_t)k * kBinsPad + ai;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int tile = wave + 4 * t;
            if (tile < 13) {
                float b_c = bc[tile * 16];
                float b_s = bs[tile * 16];
#pragma unroll
                for (int g = 0; g < FG; ++g) {
                    acc_re[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_e[g], b_c, acc_re[g][t], 0, 0, 0);
                    acc_im[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_o[g], b_s, acc_im[g][t], 0, 0, 0);
                }
            }
        }
    }
    float lmax = -INFINITY;
#pragma unroll
    for (int g = 0; g < FG; ++g) {
        // C layout 16x16: col = lane & 15 (bin), row = (lane >> 4) * 4 + r (frame)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int tile = wave + 4 * t;
            if (tile < 13) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int i = ak * 4 + r;
                    float re = acc_re[g][t][r], im = acc_im[g][t][r];
                    pw[i * LDP + tile * 16 + ai] = re * re + im * im;
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 16 * n_mels; idx += 256) {
            int i = idx & 15, m = idx >> 4;
            int f = f0 + g * 16 + i;
            const int2 rg = rg_l[m];
            const float* fw = fc_l + fo_l[m] - rg.x;
            float v = 0.0f;
            for (int bin = rg.x; bin <= rg.y; ++bin) v = fmaf(pw[i * LDP + bin], fw[bin], v);
            float lg = log10f(fmaxf(v, 1e-10f));
            if (f < kFrames) {
                logspec[((size_t)b * n_mels + m) * kFrames + f] = lg;
                lmax = fmaxf(lmax, lg);
            }
        }
        __syncthreads();
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    if (tid == 0) {
        float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(maxkey + b, float_key(mx));
    }
}

__global__ __launch_bounds__(256) void mel_finalize_kernel(const float* __restrict__ logspec, const unsigned* __restrict__ maxkey,
                                                           int n_mels, f16* __restrict__ mel_t /* [B][3002][n_mels] */,
                                                           float* __restrict__ mel_f32 /* [B][n_mels][3000] or null */) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [n_mels][65]
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const float floor_v = key_float(maxkey[b]) - 8.0f;
    for (int idx = tid; idx < n_mels * 64; idx += 256) {
        int m = idx >> 6, i = idx & 63;
        int f = f0 + i;
        float v = 0.0f;
        if (f < kFrames) {
            size_t o = ((size_t)b * n_mels + m) * kFrames + f;
            v = (fmaxf(logspec[o], floor_v) + 4.0f) * 0.25f;
            if (mel_f32) mel_f32[o] = v;
        }
        tile[m * 65 + i] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < n_mels * 64; idx += 256) {
        int i = idx / n_mels, m = idx - i * n_mels;
        int f = f0 + i;
        if (f < kFrames) mel_t[((size_t)b * kFramesPad + f + 1) * n_mels + m] = (f16)tile[m * 65 + i];
    }
}

// [n_mels][3000] f32 (reference layout) -> time-major f16 operand; used by wh_set_mel
__global__ void mel_import_kernel(const float* __restrict__ mel_f32, int n_mels, f16* __restrict__ mel_t) {
    int b = blockIdx.y;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_mels * kFrames) return;
    int f = idx / n_mels, m = idx - f * n_mels;
    mel_t[((size_t)b * kFramesPad + f + 1) * n_mels + m] = (f16)mel_f32[((size_t)b * n_mels + m) * kFrames + f];
}

void launch_log_mel(const MelTables& t, const float* pcm, const int* n_valid, int batch, float* logspec, unsigned* maxkey,
                    f16* mel_t, float* mel_f32, hipStream_t st) {
    hipMemsetAsync(maxkey, 0, sizeof(unsigned) * batch, st);
    dim3 g1((kFrames + 16 * FG - 1) / (16 * FG), batch);
    const size_t smem1 = (size_t)(kSpanLds + 16 * LDP) * sizeof(float);   // 55.8 KB: two workgroups per CU
    static PerDeviceOnce raised;
    raised.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1); });
    { ProfScope ps_(KK_MEL_POWER, st); mel_power_kernel<<<g1, 256, smem1, st>>>(pcm, n_valid, t.basis_c, t.basis_s, t.filt_c, t.filt_off, t.filt_nnz, t.filt_range, t.n_mels, logspec, maxkey); }
    dim3 g2((kFrames + 63) / 64, batch);
    { ProfScope ps_(KK_MEL_FINALIZE, st); mel_finalize_kernel<<<g2, 256, t.n_mels * 65 * sizeof(float), st>>>(logspec, maxkey, t.n_mels, mel_t, mel_f32); }
}

void launch_mel_import(const float* mel_f32, int n_mels, int batch, f16* mel_t, hipStream_t st) {
    dim3 g((n_mels * kFrames + 255) / 256, batch);
    mel_import_kernel<<<g, 256, 0, st>>>(mel_f32, n_mels, mel_t);
}

}  // namespace wh
