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
// in place again, two assignments of elements to workgroups: TILE = a workgroup owns 16 consecutive rounds (64 KiB), as the
// particle kernels' tiles do; STRIDE = round r of workgroup b is chunk r * gridDim + b (all resident workgroups sweep one
// compact window of memory together, the shape of the fastest plain copy); U loads in flight per lane
template <bool STRIDE, int U, bool NT>
__global__ __launch_bounds__(256) void k_shape(f4 *p, size_t n4) {
    const size_t chunks = (n4 + 255) / 256;
    for (size_t r0 = 0; r0 < 16; r0 += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t c = STRIDE ? (r0 + u) * gridDim.x + blockIdx.x : (size_t)blockIdx.x * 16 + r0 + u;
            const size_t i = c * 256 + threadIdx.x;
            if (c < chunks && i < n4) v[u] = NT ? __builtin_nontemporal_load(&p[i]) : p[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t c = STRIDE ? (r0 + u) * gridDim.x + blockIdx.x : (size_t)blockIdx.x * 16 + r0 + u;
            const size_t i = c * 256 + threadIdx.x;
            if (c < chunks && i < n4) {
                const f4 w = v[u] * 1.0001f + 1.0f;
                if (NT) __builtin_nontemporal_store(w, &p[i]);
                else p[i] = w;
            }
        }
    }
}
// the pinning question asked properly: workgroups whose tile lies in the first `plain_blocks` tiles run the plain body, all
// others the non-temporal one (two instantiations of the body, chosen per workgroup)
template <bool NT, bool NTS = NT>
__device__ __forceinline__ void tile_body(f4 *p, size_t n4) {
    for (size_t r0 = 0; r0 < 16; r0 += 4) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = ((size_t)blockIdx.x * 16 + r0 + u) * 256 + threadIdx.x;
            if (i < n4) v[u] = NT ? __builtin_nontemporal_load(&p[i]) : p[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = ((size_t)blockIdx.x * 16 + r0 + u) * 256 + threadIdx.x;
            if (i < n4) {
                const f4 w = v[u] * 1.0001f + 1.0f;
                if (NTS) __builtin_nontemporal_store(w, &p[i]);
                else p[i] = w;
            }
        }
    }
}
template <int REST>  // the rest: 0 nt loads + nt stores, 1 nt loads + plain stores, 2 plain loads + nt stores
__global__ __launch_bounds__(256) void k_pin(f4 *p, size_t n4, unsigned plain_blocks) {
    if (blockIdx.x < plain_blocks) tile_body<false>(p, n4);
    else if (REST == 0) tile_body<true, true>(p, n4);
    else if (REST == 1) tile_body<true, false>(p, n4);
    else tile_body<false, true>(p, n4);
}
template <int REST>
static void run_pin(f4 *d, size_t n4, size_t plain_mib) {
    const unsigned grid = (unsigned)((n4 + 4095) / 4096), pb = (unsigned)(plain_mib * 16);  // 64 KiB per workgroup
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_pin<REST>, dim3(grid), dim3(256), 0, 0, d, n4, pb);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_pin<REST>, dim3(grid), dim3(256), 0, 0, d, n4, pb);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / 20;
    printf("in place, %zu MiB, first %4zu MiB plain, rest %s: %8.1f us/pass  %7.1f GB/s moved\n", n4 * 16 >> 20, plain_mib,
           REST == 0 ? "nt loads + nt stores   " : REST == 1 ? "nt loads + plain stores" : "plain loads + nt stores", us, 2.0 * n4 * 16 / us / 1e3);
}
template <bool STRIDE, int U, bool NT>
static void run_shape(f4 *d, size_t n4, const char *name) {
    const unsigned grid = (unsigned)((n4 + 4095) / 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_shape<STRIDE, U, NT>), dim3(grid), dim3(256), 0, 0, d, n4);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_shape<STRIDE, U, NT>), dim3(grid), dim3(256), 0, 0, d, n4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / 20;
    printf("in place, 1 GiB, %-34s: %8.1f us/pass  %7.1f GB/s moved\n", name, us, 2.0 * n4 * 16 / us / 1e3);
}
// the same loop shape: read only (sum kept alive), write only, out-of-place copy -- what the memory system gives each mix
template <int WHAT>  // 0 read, 1 write, 2 copy src -> dst
__global__ __launch_bounds__(256) void k_mix(const f4 *src, f4 *dst, size_t n4, float *sink) {
    const size_t base = (size_t)blockIdx.x * 4096;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = 0; r < 16; r++) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        if (i >= n4) break;
        if (WHAT == 0) acc += src[i];
        else if (WHAT == 1) dst[i] = f4{1.f, 2.f, 3.f, (float)r};
        else dst[i] = src[i] * 1.0001f + 1.0f;
    }
    if (WHAT == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.0f;
}
int main() {
    {
        const size_t MiB = 1 << 20, total = 1024;
        f4 *a, *b;
        float *sink;
        hipMalloc(&a, total * MiB), hipMalloc(&b, total * MiB), hipMalloc(&sink, 64);
        hipMemset(a, 0, total * MiB), hipMemset(b, 0, total * MiB);
        const size_t n4 = total * MiB / 16;
        const unsigned grid = (unsigned)((n4 + 4095) / 4096);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        for (int what = 0; what < 3; what++) {
            auto launch = [&]() {
                if (what == 0) hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(256), 0, 0, a, b, n4, sink);
                else if (what == 1) hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(256), 0, 0, a, b, n4, sink);
                else hipLaunchKernelGGL(k_mix<2>, dim3(grid), dim3(256), 0, 0, a, b, n4, sink);
            };
            for (int i = 0; i < 5; i++) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; i++) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1000.0 / 20, bytes = (what == 2 ? 2.0 : 1.0) * total * MiB;
            printf("1 GiB arrays, %-28s: %8.1f us/pass  %7.1f GB/s moved\n",
                   what == 0 ? "read only" : what == 1 ? "write only" : "copy a -> b (read + write)", us, bytes / us / 1e3);
        }
        run_shape<false, 1, false>(a, n4, "tile,   1 load in flight");
        run_shape<false, 4, false>(a, n4, "tile,   4 loads in flight");
        run_shape<false, 4, true>(a, n4, "tile,   4 loads in flight, nt");
        run_shape<true, 1, false>(a, n4, "stride, 1 load in flight");
        run_shape<true, 4, false>(a, n4, "stride, 4 loads in flight");
        run_shape<true, 4, true>(a, n4, "stride, 4 loads in flight, nt");
        run_shape<true, 8, true>(a, n4, "stride, 8 loads in flight, nt");
        for (size_t mib : {(size_t)0, (size_t)64, (size_t)128, (size_t)160, (size_t)192, (size_t)224, (size_t)256, (size_t)320, (size_t)1024}) run_pin<0>(a, n4, mib);
        for (size_t mib : {(size_t)0, (size_t)128, (size_t)192, (size_t)256}) run_pin<1>(a, n4, mib);
        for (size_t mib : {(size_t)0, (size_t)128, (size_t)192, (size_t)256}) run_pin<2>(a, n4, mib);
        hipFree(a), hipFree(b), hipFree(sink);
    }
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
