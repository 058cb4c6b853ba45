// Which compute units does a CU-masked HIP stream use?  (round 6 probe for WH_CU_PARTS, whisperkit_amd/csrc/capi.hip create_session_stream)
// For a set of masks of hipExtStreamCreateWithCUMask: launch many one-wave workgroups that record HW_REG_XCC_ID and HW_REG_HW_ID, print per mask the
// distinct (xcc, se, cu) triples seen, grouped by XCC.   hipcc --offload-arch=gfx950 -O2 -o tools/build/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out) {
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that the launch spreads over every CU the mask allows
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("{\"multiprocessors\": %d}\n", n_cu);
    const int n_wg = 4096;
    unsigned* d = nullptr;
    hipMalloc(&d, n_wg * 8);
    std::vector<unsigned> h(2 * n_wg);
    struct Case { const char* name; int lo, hi; };
    const Case cases[] = {{"all", 0, 256}, {"first_128", 0, 128}, {"second_128", 128, 256}, {"first_64", 0, 64}, {"first_32", 0, 32}, {"first_8", 0, 8}, {"bits_85_170", 85, 170}};
    for (const Case& c : cases) {
        uint32_t mask[8] = {};
        for (int i = c.lo; i < c.hi; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("{\"mask\": \"%s\", \"error\": \"create failed\"}\n", c.name); continue; }
        hipMemsetAsync(d, 0xff, n_wg * 8, st);
        where_kernel<<<n_wg, 64, 0, st>>>(d);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, n_wg * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int i = 0; i < n_wg; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        size_t total = 0;
        printf("{\"mask\": \"%s\", \"bits\": [%d, %d], \"cus_per_xcc\": {", c.name, c.lo, c.hi);
        bool first = true;
        for (auto& kv : per_xcc) { printf("%s\"%u\": %zu", first ? "" : ", ", kv.first, kv.second.size()); first = false; total += kv.second.size(); }
        printf("}, \"distinct_cus\": %zu}\n", total);
        hipStreamDestroy(st);
    }
    // the same through a captured graph launched into a masked stream: do the kernel nodes keep the mask?
    {
        uint32_t mask[8] = {};
        for (int i = 0; i < 64; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t st;
        hipExtStreamCreateWithCUMask(&st, 8, mask);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        where_kernel<<<n_wg, 64, 0, st>>>(d);
        where_kernel<<<n_wg, 64, 0, st>>>(d);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, n_wg * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> all;
        for (int i = 0; i < n_wg; ++i) all.insert(((h[2 * i + 1] & 0xf) << 16) | ((h[2 * i] >> 8) & 0xff));
        printf("{\"mask\": \"first_64 through a captured graph\", \"distinct_cus\": %zu}\n", all.size());
    }
    return 0;
}
