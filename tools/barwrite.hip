// tools/barwrite.hip -- can the host WRITE per-frame records straight into device memory (large BAR), so that the kernels read
// them locally instead of over the bus from pinned host memory?  What a dependent read costs a one-workgroup kernel from
//   (a) pinned host memory (hipHostMalloc: what the range / small / general launches read their per-frame records from),
//   (b) fine-grained device memory the host wrote through its own pointer right before the launch (if the platform maps it),
//   (c) ordinary device memory filled by hipMemcpyAsync in the stream (the staged form),
// and whether (b) is always seen by the kernel launched after the write (10 000 launches, a new value each).
//   hipcc --offload-arch=gfx950 -O2 tools/barwrite.hip -o tools/barwrite && tools/barwrite
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <immintrin.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static sigjmp_buf g_jb;
static void on_segv(int) { siglongjmp(g_jb, 1); }

// one workgroup per record: a record names where the data is (a dependent hop), like desc -> record -> particles
__global__ void k_read(const uint32_t *rec, const uint32_t *data, uint32_t *out, int hops) {
    uint32_t v = rec[blockIdx.x * 32 + 0];  // record: an index
    for (int h = 1; h < hops; h++) v = rec[(blockIdx.x * 32 + (v & 15u) + h) & 0xFFFFu];
    if (threadIdx.x == 0) out[blockIdx.x] = v + data[v & 1023u];
}

static double time_launches(hipStream_t s, const uint32_t *rec, const uint32_t *data, uint32_t *out, int blocks, int hops, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    double tot = 0;
    for (int i = 0; i < n; i++) {
        hipExtLaunchKernelGGL(k_read, dim3(blocks), dim3(64), 0, s, e0, e1, 0, rec, data, out, hops);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= n / 4) tot += ms * 1e3;
    }
    return tot / (n - n / 4);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t N = 1 << 16;
    uint32_t *pinned = nullptr, *fine = nullptr, *dev = nullptr, *data = nullptr, *out = nullptr, *h_out = nullptr;
    CK(hipHostMalloc((void **)&pinned, N * 4, hipHostMallocDefault));
    CK(hipMalloc((void **)&dev, N * 4));
    CK(hipMalloc((void **)&data, 1024 * 4));
    CK(hipMemset(data, 0, 1024 * 4));
    CK(hipMalloc((void **)&out, 4096 * 4));
    CK(hipHostMalloc((void **)&h_out, 4096 * 4, hipHostMallocDefault));
    hipError_t fe = hipExtMallocWithFlags((void **)&fine, N * 4, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(fe));
    bool fine_ok = fe == hipSuccess;
    if (fine_ok) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, fine) == hipSuccess) printf("  type %d device %d hostPointer %p devicePointer %p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
        signal(SIGSEGV, on_segv), signal(SIGBUS, on_segv);
        if (sigsetjmp(g_jb, 1) == 0) {
            volatile uint32_t *f = fine;
            f[0] = 123u;
            _mm_sfence();
            printf("  host write through the device pointer: ok (reads back %u)\n", f[0]);
        } else {
            printf("  host write through the device pointer: FAULT -- the platform does not map it\n");
            fine_ok = false;
        }
        signal(SIGSEGV, SIG_DFL), signal(SIGBUS, SIG_DFL);
    }
    for (size_t i = 0; i < N; i++) pinned[i] = (uint32_t)(i * 2654435761u);
    CK(hipMemcpy(dev, pinned, N * 4, hipMemcpyHostToDevice));
    if (fine_ok) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 16; r++) memcpy(fine, pinned, N * 4);
        _mm_sfence();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 16;
        printf("  host memcpy of %zu KiB into it: %.1f us (%.2f GB/s)\n", N * 4 / 1024, us, N * 4 / us / 1e3);
        const auto t1 = std::chrono::steady_clock::now();
        for (int r = 0; r < 16; r++) memcpy(pinned, pinned + N / 2, N * 2);
        printf("  (the same bytes pinned -> pinned: %.1f us)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / 16 * 2);
    }
    for (int blocks : {1, 64, 512, 2048})
        for (int hops : {1, 2}) {
            printf("blocks %4d hops %d: kernel us  pinned %.2f", blocks, hops, time_launches(s, pinned, data, out, blocks, hops, 400));
            printf("  device %.2f", time_launches(s, dev, data, out, blocks, hops, 400));
            if (fine_ok) printf("  host-written device (fine-grained) %.2f", time_launches(s, fine, data, out, blocks, hops, 400));
            printf("\n");
        }
    if (fine_ok) {  // visibility: the host writes a record, launches at once, the kernel must see THIS value
        int bad = 0;
        const int n = 10000;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= n; i++) {
            for (int b = 0; b < 8; b++) ((volatile uint32_t *)fine)[b * 32] = (uint32_t)i * 1024u;  // (data[v & 1023] = data[0] = 0)
            _mm_sfence();
            hipLaunchKernelGGL(k_read, dim3(8), dim3(64), 0, s, fine, data, out, 1);
            if (i % 64 == 0 || i == n) {  // the stream is in order: checking the last launch of a batch checks an unsynchronised pipeline
                hipMemcpyAsync(h_out, out, 8 * 4, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                for (int b = 0; b < 8; b++) bad += h_out[b] != (uint32_t)i * 1024u;
            }
        }
        printf("visibility: %d launches pipelined, a new host-written value each, checked every 64: %d stale reads (%.1f us per launch)\n", n, bad,
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n);
        // ... and with the write issued while the PREVIOUS launch may still be running (distinct slots, as a parameter ring would)
    }
    return 0;
}
