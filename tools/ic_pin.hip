// Can a slice of an in-place streaming working set be kept in the 256 MiB Infinity Cache across launches while the rest
// streams past it non-temporally?  In-place float4 update (read 16 B + write 16 B per element) over `total` MiB; the
// first `pinned` MiB use plain loads/stores, the rest non-temporal ones.  Reports us per pass and GB/s moved.
//   hipcc --offload-arch=gfx950 -O3 tools/ic_pin.hip -o tools/ic_pin && tools/ic_pin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 plain, 1 nt loads + stores, 2 nt stores only
__global__ __launch_bounds__(256) void k_update(f4 *p, size_t n4_plain, size_t n4) {
    const size_t per_block = 4096;  // elements per workgroup: 16 rounds of 256
    const size_t base = (size_t)blockIdx.x * per_block;
    const bool plain = base < n4_plain;
#pragma unroll 4
    for (int r = 0; r < 16; r++) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        if (i >= n4) return;
        f4 v;
        if (plain || MODE == 0 || MODE == 2) v = p[i];
        else v = __builtin_nontemporal_load(&p[i]);
        v = v * 1.0001f + 1.0f;
        if (plain || MODE == 0) p[i] = v;
        else __builtin_nontemporal_store(v, &p[i]);
    }
}
int main() {
    const size_t MiB = 1 << 20;
    for (size_t total : {(size_t)512, (size_t)1024, (size_t)2048}) {
        f4 *d;
        hipMalloc(&d, total * MiB);
        hipMemset(d, 0, total * MiB);
        const size_t n4 = total * MiB / 16;
        const unsigned grid = (unsigned)((n4 + 4095) / 4096);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        for (int mode : {0, 1, 2})
            for (size_t pinned : {(size_t)0, (size_t)64, (size_t)128, (size_t)192, (size_t)256}) {
                if (mode == 0 && pinned) continue;
                const size_t n4p = pinned * MiB / 16;
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k_update<0>, dim3(grid), dim3(256), 0, 0, d, n4p, n4);
                    else if (mode == 1) hipLaunchKernelGGL(k_update<1>, dim3(grid), dim3(256), 0, 0, d, n4p, n4);
                    else hipLaunchKernelGGL(k_update<2>, dim3(grid), dim3(256), 0, 0, d, n4p, n4);
                };
                for (int i = 0; i < 5; i++) launch();
                hipEventRecord(e0, 0);
                const int reps = 20;
                for (int i = 0; i < reps; i++) launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1000.0 / reps;
                printf("total %4zu MiB  mode %s  plain slice %3zu MiB : %8.1f us/pass  %7.1f GB/s moved\n", total,
                       mode == 0 ? "plain   " : mode == 1 ? "nt ld+st" : "nt st   ", pinned, us, 2.0 * total * MiB / us / 1e3);
            }
        hipFree(d);
    }
    return 0;
}
