// Does hipExtStreamCreateWithCUMask confine kernels to the CUs of the mask on this GPU, and in which bit order?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o tools/probes/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void who(unsigned *out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        out[blockIdx.x] = ((xcc & 0xF) << 16) | ((hw >> 8) & 0xFF);                      // xcc | se_id, sh_id, cu_id
        for (volatile int i = 0; i < 20000; i++) { }
    }
}

static void run(const char *name, hipStream_t s) {
    const int B = 4096;
    unsigned *d; hipMalloc(&d, 4 * B);
    hipLaunchKernelGGL(who, dim3(B), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(B);
    hipMemcpy(h.data(), d, 4 * B, hipMemcpyDeviceToHost);
    std::set<unsigned> cus, xccs;
    for (unsigned v : h) { cus.insert(v); xccs.insert(v >> 16); }
    printf("%-28s distinct (xcc, se/sh/cu) = %zu, xccs = %zu\n", name, cus.size(), xccs.size());
    hipFree(d);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    hipStream_t s0; hipStreamCreate(&s0); run("unmasked", s0);
    const int words = (p.multiProcessorCount + 31) / 32;
    for (int variant = 0; variant < 3; variant++) {
        std::vector<uint32_t> m(words, 0);
        const char *name = "";
        if (variant == 0) { name = "every 16th CU"; for (int c = 0; c < p.multiProcessorCount; c += 16) m[c / 32] |= 1u << (c % 32); }
        if (variant == 1) { name = "first 16 CUs"; for (int c = 0; c < 16; c++) m[c / 32] |= 1u << (c % 32); }
        if (variant == 2) { name = "all but every 16th"; for (int c = 0; c < p.multiProcessorCount; c++) if (c % 16) m[c / 32] |= 1u << (c % 32); }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, m.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); continue; }
        run(name, s);
        hipStreamDestroy(s);
    }
    return 0;
}
