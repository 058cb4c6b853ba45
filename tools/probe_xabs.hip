// Probe for the weight-absorbed cross-attention kernel (round 4), standalone: hipcc --offload-arch=gfx950 -O3 tools/probe_xabs.hip -o tools/build/probe_xabs
//   1. ds_read_b64_tr_b16: which supplier lane's 8 bytes end up in which lane / element (the kernel's P V operand reads depend on it)
//   2. an LDS-DMA ring stream (global_load_lds_dwordx4, 8 waves, ring of R tiles of T KB, one workgroup per CU): the HBM rate a
//      one-workgroup-per-CU kernel reaches when its in-flight bytes are bounded by LDS (the question DESIGN §7.1 left open)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void tr_probe(unsigned short* out, const int* addr_halves) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_halves[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

// ring stream: workgroup w streams n_tiles tiles of TILE bytes starting at src + w * wg_stride, ring of R LDS slots; every wave issues
// TILE / 8192 pieces per tile; per tile: wait own pieces of tile t (vmcnt leaves the younger tiles in flight), barrier, optional LDS reads
template <int TILE, int R, int READS>
__global__ __launch_bounds__(512) void ring_stream(const unsigned char* src, size_t wg_stride, int n_tiles, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int PPW = TILE / 8192;     // pieces per wave per tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * wg_stride;
    auto issue = [&](int t) {
        unsigned char* dst = smem + (t % R) * TILE + wave * (PPW * 1024);
        const unsigned char* s = base + (size_t)t * TILE + wave * (PPW * 1024) + lane * 16;
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + p * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    };
    float acc = 0.0f;
    for (int t = 0; t < R - 1 && t < n_tiles; ++t) issue(t);
    for (int t = 0; t < n_tiles; ++t) {
        // outstanding before the wait: tiles t .. min(t + R - 2, n - 1); leave the younger ones in flight
        const int younger = min(t + R - 2, n_tiles - 1) - t;
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PPW) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PPW) : "memory"); break;
        }
        __syncthreads();
        if (t + R - 1 < n_tiles) issue(t + R - 1);      // slot (t - 1) % R: its readers passed the barrier above
        if (READS) {
            const unsigned char* tile = smem + (t % R) * TILE;
#pragma unroll
            for (int i = 0; i < READS; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(tile + ((wave * READS + i) * 1024 + lane * 16) % TILE);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

template <int TILE, int R, int READS>
static void run_stream(const unsigned char* buf, size_t total, int wgs, const char* tag, float* sink) {
    const size_t per_wg = total / wgs / TILE * TILE;
    const int n_tiles = (int)(per_wg / TILE);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_stream<TILE, R, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, TILE * R));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) ring_stream<TILE, R, READS><<<wgs, 512, TILE * R>>>(buf, per_wg, n_tiles, sink);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) ring_stream<TILE, R, READS><<<wgs, 512, TILE * R>>>(buf, per_wg, n_tiles, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)per_wg * wgs;
    printf("{\"probe\": \"ring_stream\", \"tag\": \"%s\", \"tile_kb\": %d, \"ring\": %d, \"reads\": %d, \"wgs\": %d, \"mb\": %.1f, \"us\": %.2f, \"tbps\": %.3f}\n", tag, TILE / 1024, R,
           READS, wgs, bytes / 1e6, ms * 1000 / reps, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    // ---- 1. tr16 semantics
    {
        unsigned short* out; int* addr;
        CK(hipMalloc(&out, 256 * 2)); CK(hipMalloc(&addr, 64 * 4));
        std::vector<int> a(64);
        std::vector<unsigned short> o(256);
        for (int pat = 0; pat < 3; ++pat) {
            for (int l = 0; l < 64; ++l) a[l] = pat == 0 ? l * 4 : pat == 1 ? l * 64 : ((l & 15) >> 2) * 1288 + (l & 3) * 4 + (l >> 4) * 16;
            CK(hipMemcpy(addr, a.data(), 256, hipMemcpyHostToDevice));
            tr_probe<<<1, 64>>>(out, addr);
            CK(hipMemcpy(o.data(), out, 512, hipMemcpyDeviceToHost));
            printf("tr16 pattern %d (lane address in halves: %s)\n", pat, pat == 0 ? "4*lane" : pat == 1 ? "64*lane" : "row (l&15)>>2 stride 1288, col 4*(l&3) + 16*(l>>4)");
            for (int l = 0; l < 64; ++l) {
                printf("  lane %2d:", l);
                for (int j = 0; j < 4; ++j) {
                    // decode: which supplier lane / element this value came from
                    int v = o[l * 4 + j], sl = -1, se = -1;
                    for (int s = 0; s < 64; ++s) if (v >= a[s] && v < a[s] + 4) { sl = s; se = v - a[s]; }
                    printf(" %5d(s%02d.e%d)", v, sl, se);
                }
                printf("\n");
            }
        }
    }
    // ---- 2. ring stream
    {
        const size_t total = (size_t)1 << 30;
        unsigned char* buf; float* sink;
        CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096));
        run_stream<40960, 3, 0>(buf, total, 256, "1GiB", sink);
        run_stream<40960, 3, 10>(buf, total, 256, "1GiB", sink);
        run_stream<40960, 2, 0>(buf, total, 256, "1GiB", sink);
        run_stream<16384, 8, 0>(buf, total, 256, "1GiB", sink);
        run_stream<16384, 4, 0>(buf, total, 256, "1GiB", sink);
        run_stream<32768, 4, 0>(buf, total, 256, "1GiB", sink);
        run_stream<65536, 2, 0>(buf, total, 256, "1GiB", sink);
        run_stream<40960, 3, 0>(buf, total, 512, "1GiB", sink);
        run_stream<40960, 3, 0>(buf, (size_t)245760000, 256, "246MB (one session's encoder outputs: Infinity-Cache sized)", sink);
        run_stream<40960, 3, 0>(buf, (size_t)245760000 / 2, 256, "123MB", sink);
        run_stream<40960, 3, 0>(buf, (size_t)245760000 / 8, 256, "31MB", sink);
    }
    return 0;
}
