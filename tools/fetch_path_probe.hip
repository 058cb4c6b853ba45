// Fetch-path probe (round 6): how many bytes per clock does ONE CU get through (a) LDS-DMA (global_load_lds_dwordx4) and (b) ordinary vector
// loads into registers (global_load_dwordx4), when the data is L2-resident, and do the two paths add up?  Standalone:
//     hipcc --offload-arch=gfx950 -O3 tools/fetch_path_probe.hip -o tools/build/fetch_path_probe
//
// Why: gemm256_kernel moves 64 KB of operands per 64-wide K-tile and CU through LDS-DMA and spends ~5 400 cycles per K-tile (2 048 of them
// matrix work): 12 bytes per clock and CU - the same figure the HBM-bound xabs_attn stream reaches.  If that is a property of the LDS-DMA
// path and not of the memory behind it, a second path (weights as register fragments) would lift the encoder's GEMMs.
//
// Every workgroup (one per CU, 512 threads) re-reads its own window of `win` bytes `reps` times inside one launch; windows of 256 KB .. 8 MB
// per workgroup: 64 MB .. 2 GB over the chip (L2 4 MB per XCD, Infinity Cache 256 MB).  `shared` > 1: groups of `shared` consecutive
// workgroup ids of an XCD read the SAME window (the GEMM's panel sharing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA only; 1: register loads only; 2: both (half of the bytes each)
template <int MODE>
__global__ __launch_bounds__(512) void fetch_kernel(const unsigned char* src, size_t win, int reps, int shared, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];      // 2 x 64 KB stages for the DMA
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const size_t widx = (size_t)xcd * ((gridDim.x >> 3) / shared) + local / shared;
    const unsigned char* base = src + widx * win;
    const int chunks = (int)(win / 65536);              // 64 KB per step and workgroup: 8 KB per wave = 8 x 1 KB pieces
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int c = 0; c < chunks; ++c) {
            const unsigned char* p = base + (size_t)c * 65536 + wave * 8192 + lane * 16;
            unsigned char* dst = smem + (c & 1) * 65536 + wave * 8192;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 1024), (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                if (c > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // the previous step's pieces have landed
            } else if (MODE == 1) {
                u32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + i * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += v[i].x ^ v[i].w;
            } else {
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 1024), (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(p + (4 + i) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += v[i].x ^ v[i].w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[tid] = acc;
}

template <int MODE>
static void run(const unsigned char* buf, size_t win, int shared, unsigned* sink, const char* what) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fetch_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int wgs = 256;
    const int reps = (int)((size_t)64 * 1024 * 1024 / win) > 0 ? (int)((size_t)64 * 1024 * 1024 / win) : 1;      // 64 MB per workgroup and launch
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fetch_kernel<MODE><<<wgs, 512, 131072>>>(buf, win, 2, shared, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    fetch_kernel<MODE><<<wgs, 512, 131072>>>(buf, win, reps, shared, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)win * reps * wgs;
    int clk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("{\"path\": \"%s\", \"window_kb_per_wg\": %zu, \"sharers\": %d, \"unique_mb\": %.0f, \"reps\": %d, \"ms\": %.3f, \"tbps\": %.2f, \"gb_s_per_cu\": %.1f, \"bytes_per_clk_per_cu_at_2400mhz\": %.1f}\n",
           what, win / 1024, shared, (double)win * wgs / shared / 1e6, reps, ms, bytes / (ms * 1e-3) / 1e12, bytes / wgs / (ms * 1e-3) / 1e9,
           bytes / wgs / (ms * 1e-3) / 2.4e9);
    fflush(stdout);
}

int main() {
    const size_t total = (size_t)2 << 30;
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096));
    for (size_t win : {(size_t)256 << 10, (size_t)1 << 20, (size_t)8 << 20}) {
        for (int shared : {1, 8}) {
            run<0>(buf, win, shared, sink, "lds_dma");
            run<1>(buf, win, shared, sink, "registers");
            run<2>(buf, win, shared, sink, "half_and_half");
        }
    }
    return 0;
}
