// Co-residence probe (round 6; VERDICT r05 "next round" 3): can a projection-shaped kernel run ON the CUs that an
// xabs_attn-shaped stream kernel occupies, and what does each pay?  Standalone:
//     hipcc --offload-arch=gfx950 -O3 tools/coresidence_probe.hip -o /tmp/coresidence_probe && /tmp/coresidence_probe
//
//   stream kernel  S<SV>   512 threads, one workgroup per CU (256 workgroups), an LDS ring fed by LDS-DMA (buffer of R x 40 KB = the xabs_attn
//                          ring of 7 half tiles when R x 40960 = 143 360 ... here LDS bytes are a launch parameter), its register allocation pinned
//                          to SV VGPRs (xabs_attn<5, 2>: 214 -> 216 allocated, 2 waves per SIMD = 432 of 512), streaming `mb_per_wg` per launch
//   small kernel   P<PV>   256 threads (one wave per SIMD), allocation pinned to PV VGPRs, `lds` bytes of LDS, 160 .. 640 workgroups; every wave
//                          streams TW 1 KB weight tiles in chunks of TC (loads of chunk c + 1 in flight under the MFMAs of chunk c, as
//                          dec32_proj_kernel does) - the latency-bound shape of the decoder's projection launches
//
// Three timings per configuration (HIP events on each stream, host wall for the pair): S alone, P alone (back-to-back launches), and both
// streams together.  If P's workgroups fit beside S's on a CU (registers: 512 - 2 x SV >= PV; LDS: 160 KB - S's - P's >= 0) the pair's
// wall time approaches max(S, P'); if they do not fit, it approaches S + P (P's workgroups only run in the gaps between S launches).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V> __device__ __forceinline__ void pin_vgprs();      // touches the highest register of the wanted allocation
#define PIN(V, R) template <> __device__ __forceinline__ void pin_vgprs<V>() { asm volatile("v_mov_b32 " R ", 0" ::: R); }
PIN(40, "v39") PIN(64, "v63") PIN(80, "v79") PIN(96, "v95") PIN(112, "v111") PIN(128, "v127") PIN(176, "v175") PIN(192, "v191") PIN(200, "v199") PIN(216, "v215")

constexpr int TILE = 40960;      // 16 keys x 1280 x 2 bytes: one xabs_attn tile
template <int SV, int THREADS = 512>
__global__ __launch_bounds__(THREADS) void stream_kernel(const unsigned char* src, size_t wg_stride, int n_tiles, int ring, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    pin_vgprs<SV>();
    constexpr int PPW = TILE / (THREADS * 16);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * wg_stride;
    auto issue = [&](int t) {
        unsigned char* dst = smem + (t % ring) * TILE + wave * (PPW * 1024);
        const unsigned char* s = base + (size_t)t * TILE + wave * (PPW * 1024) + lane * 16;
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + p * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 2);
    };
    for (int t = 0; t < ring - 1 && t < n_tiles; ++t) issue(t);
    float acc = 0.0f;
    for (int t = 0; t < n_tiles; ++t) {
        const int younger = min(t + ring - 2, n_tiles - 1) - t;
        if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        __syncthreads();
        if (t + ring - 1 < n_tiles) issue(t + ring - 1);
        const float4 v = *reinterpret_cast<const float4*>(smem + (t % ring) * TILE + wave * 1024 + lane * 16);
        acc += v.x;
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

template <int PV, int TC>
__global__ __launch_bounds__(256) void small_kernel(const u32x4* w, const u32x4* z, int tw, int lds_words, float* out) {
    extern __shared__ float red[];
    pin_vgprs<PV>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* wp = w + ((size_t)blockIdx.x * 4 + wave) * tw * 64 + lane;
    const u32x4* zp = z + (size_t)wave * tw * 64 + lane;
    f32x16 acc = {0};
    u32x4 wa[TC], za[TC], wb[TC], zb[TC];
    auto ld = [&](u32x4 (&ww)[TC], u32x4 (&zz)[TC], int c) {
#pragma unroll
        for (int i = 0; i < TC; ++i) { ww[i] = __builtin_nontemporal_load(wp + (size_t)(c * TC + i) * 64); zz[i] = zp[(size_t)(c * TC + i) * 64]; }
    };
    auto mm = [&](const u32x4 (&ww)[TC], const u32x4 (&zz)[TC]) {
#pragma unroll
        for (int i = 0; i < TC; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ww[i]), __builtin_bit_cast(f16x8, zz[i]), acc, 0, 0, 0);
    };
    const int nch = tw / TC;
    ld(wa, za, 0);
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) ld(wb, zb, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(wa, za);
        if (c + 2 < nch) ld(wa, za, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nch) mm(wb, zb);
    }
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (lds_words >= 256) {                      // the cross-wave meeting of the projection kernel, when the probe carries LDS
        red[threadIdx.x] = s;
        __syncthreads();
        s = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

struct Timed { double us_per_launch; };

template <typename F>
static double time_stream(hipStream_t st, int reps, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.0 / reps;
}

template <int SV, int PV, int TC, int STHREADS = 512, int SWGS = 256>
static void run(const unsigned char* buf, const u32x4* w, const u32x4* z, float* sink, float* out, int ring, int s_lds, int p_lds, int p_wgs, int tw) {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    // s_lds: the stream workgroup's LDS allocation (xabs_attn today: 161 920 bytes = ring of 7 half tiles + partial / P^T / alpha scratch); the probe's own ring uses `ring` whole tiles of it
    const int n_tiles = 94;                                // one slot: 1500 keys
    const size_t per_wg = (size_t)n_tiles * TILE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<SV, STHREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, s_lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_kernel<PV, TC>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    auto S = [&] { stream_kernel<SV, STHREADS><<<SWGS, STHREADS, s_lds, s1>>>(buf, per_wg, n_tiles, ring, sink); };
    auto P = [&] { small_kernel<PV, TC><<<p_wgs, 256, p_lds, s2>>>(w, z, tw, p_lds / 4, out); };
    const int rs = 20, rp = 200;
    const double s_alone = time_stream(s1, rs, S);
    const double p_alone = time_stream(s2, rp, P);
    // together: host wall over both streams, the number of P launches chosen so that both streams are busy for about the same time alone
    const int np = (int)(s_alone * rs / p_alone);
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    double p_done_us = 0.0;
    std::thread th([&] { for (int i = 0; i < np; ++i) P(); CK(hipStreamSynchronize(s2));
                         p_done_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); });
    for (int i = 0; i < rs; ++i) S();
    CK(hipStreamSynchronize(s1));
    const double s_done_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    th.join();
    const double wall = s_done_us > p_done_us ? s_done_us : p_done_us;
    const double s_tot = s_alone * rs, p_tot = p_alone * np, serial = s_tot + p_tot;
    const double overlap = (serial - wall) / (s_tot < p_tot ? s_tot : p_tot);      // 1 = the shorter stream is hidden completely, 0 = the two add up
    const int regs_left = 512 - (STHREADS / 256) * ((SV + 7) / 8 * 8);
    printf("{\"stream_threads\": %d, \"stream_wgs\": %d, \"stream_vgprs\": %d, \"stream_lds\": %d, \"small_vgprs\": %d, \"small_lds\": %d, \"small_wgs\": %d, \"small_tiles_per_wave\": %d, \"tc\": %d, "
           "\"fits_registers\": %s, \"fits_lds\": %s, \"stream_alone_us\": %.1f, \"stream_tbps_alone\": %.2f, \"small_alone_us\": %.2f, "
           "\"small_launches\": %d, \"pair_wall_us\": %.0f, \"sum_alone_us\": %.0f, \"overlap\": %.2f, \"stream_us_beside_small\": %.1f, \"small_us_beside_stream\": %.2f}\n",
           STHREADS, SWGS, SV, s_lds, PV, p_lds, p_wgs, tw, TC, regs_left >= PV ? "true" : "false", (163840 - s_lds >= p_lds) ? "true" : "false",
           s_alone, per_wg * (double)SWGS / (s_alone * 1e-6) / 1e12, p_alone, np, wall, serial, overlap, s_done_us / rs, p_done_us / np);
    fflush(stdout);
    CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
}

int main() {
    const size_t total = (size_t)256 * 94 * TILE;
    unsigned char* buf; float *sink, *out; u32x4 *w, *z;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096));
    const size_t wbytes = (size_t)640 * 4 * 80 * 1024;                       // 640 workgroups x 4 waves x 80 tiles of 1 KB
    CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0, wbytes)); CK(hipMalloc(&z, (size_t)4 * 80 * 1024)); CK(hipMemset(z, 0, (size_t)4 * 80 * 1024));
    CK(hipMalloc(&out, (size_t)640 * 256 * 4));
    const int TODAY = 161920, DIET6 = 6 * 20480 + 18560 /* 141 440 */, DIET5 = 5 * 20480 + 18560 /* 120 960 */;
    // today's footprints: xabs_attn 216 VGPRs x 2 waves per SIMD + 158 KB; the projection 196 .. 204 VGPRs (TC = 5) + 23.5 KB: no fit
    run<216, 200, 5>(buf, w, z, sink, out, 3, TODAY, 24064, 160, 20);
    // a small kernel that fits beside today's xabs_attn: <= 80 VGPRs, <= 1.5 KB of LDS (chunks of ONE tile)
    run<216, 80, 1>(buf, w, z, sink, out, 3, TODAY, 1024, 160, 20);
    run<216, 80, 1>(buf, w, z, sink, out, 3, TODAY, 0, 640, 5);
    run<216, 64, 1>(buf, w, z, sink, out, 3, TODAY, 0, 640, 5);
    // ... and the same small kernel when only its LDS does not fit (control: registers fit, 16 KB of LDS do not)
    run<216, 80, 1>(buf, w, z, sink, out, 3, TODAY, 16384, 160, 20);
    // xabs_attn on a register / LDS diet (<= 200 / 192 VGPRs, ring of 6 / 5 half tiles): room for 112 / 128 registers and 16 - 40 KB
    run<200, 112, 2>(buf, w, z, sink, out, 2, DIET6, 16384, 160, 20);
    run<192, 128, 2>(buf, w, z, sink, out, 2, DIET5, 16384, 160, 20);
    run<192, 128, 2>(buf, w, z, sink, out, 2, DIET5, 16384, 640, 5);
    // a stream kernel that leaves a third of the registers (176 x 2 = 352): two 80-register waves per SIMD
    run<176, 80, 1>(buf, w, z, sink, out, 2, DIET5, 16384, 320, 10);
    // controls.  (a) the stream on HALF of the CUs: the small kernel has 128 free CUs (what the bench does today)
    run<216, 200, 5, 512, 128>(buf, w, z, sink, out, 3, TODAY, 24064, 160, 20);
    // (b) a stream workgroup of ONE wave per SIMD (256 threads): its registers are one contiguous block per SIMD whatever their number
    run<216, 80, 1, 256>(buf, w, z, sink, out, 2, DIET5, 16384, 160, 20);
    run<216, 200, 5, 256>(buf, w, z, sink, out, 2, DIET5, 24064, 160, 20);
    run<128, 128, 2, 256>(buf, w, z, sink, out, 2, DIET5, 16384, 160, 20);
    // (c) two waves per SIMD that leave half of the file (2 x 128) and (d) a 40-register small kernel beside today's footprint (2 x 40 free)
    run<128, 128, 2>(buf, w, z, sink, out, 2, DIET5, 16384, 160, 20);
    run<216, 40, 1>(buf, w, z, sink, out, 3, TODAY, 0, 640, 5);
    return 0;
}
