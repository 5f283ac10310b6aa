// Latency of one dependent scalar-path hop from a workgroup's point of view: pointer chase through device memory (cold lines,
// 4 KiB apart) and through pinned host memory (what fw_k_update_range's per-frame records live in).  One wave, 512 hops.
//   hipcc --offload-arch=gfx950 -O3 tools/hoplat.hip -o tools/hoplat && tools/hoplat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_chase(const uint32_t *p, uint32_t hops, uint32_t *out, unsigned long long *ticks) {
    uint32_t i = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t h = 0; h < hops; h++) i = p[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    *out = i, *ticks = t1 - t0;
}
int main() {
    const uint32_t hops = 512, stride = 1024;  // uint32 elements: 4 KiB apart
    const size_t n = (size_t)hops * stride + 1;
    std::vector<uint32_t> h(n, 0);
    for (uint32_t k = 0; k < hops; k++) h[(size_t)k * stride] = (k + 1) * stride;
    uint32_t *d, *pin, *out;
    unsigned long long *ticks, hticks;
    hipMalloc(&d, n * 4), hipMalloc(&out, 4), hipMalloc(&ticks, 8);
    hipHostMalloc(&pin, n * 4, hipHostMallocDefault);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    memcpy(pin, h.data(), n * 4);
    for (int rep = 0; rep < 3; rep++)
        for (int which = 0; which < 2; which++) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, which ? pin : d, hops, out, ticks);
            hipDeviceSynchronize();
            hipMemcpy(&hticks, ticks, 8, hipMemcpyDeviceToHost);
            printf("%-20s %7.1f ns per dependent hop\n", which ? "pinned host memory" : "device memory (cold)", hticks * 10.0 / hops);
        }
    return 0;
}
