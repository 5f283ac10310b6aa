// Does a kernel see what the previous kernel of the same stream wrote, when several PROCESSES share the GPU?
// Kernel W writes pattern(iter) into a buffer, kernel R -- launched right behind it on the same stream, with its blocks shifted
// by `shift` so that every block reads what a block of ANOTHER XCD wrote -- checks it and counts mismatches per reader block.
// Run alone: 0 mismatches expected.  Run as N concurrent processes (tools/xcd_coherence_load.sh) to see whether sharing
// the device changes that.  Independent of the particle library (no code shared).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_write(uint32_t *buf, uint32_t n, uint32_t iter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = i * 2654435761u + iter;
}
__global__ void k_read(const uint32_t *buf, uint32_t n, uint32_t iter, uint32_t shift, uint32_t *bad, uint32_t *bad_block) {
    const uint32_t b = (blockIdx.x + shift) % gridDim.x;
    const uint32_t i = b * blockDim.x + threadIdx.x;
    if (i < n && buf[i] != i * 2654435761u + iter) {
        atomicAdd(bad, 1u);
        atomicAdd(&bad_block[blockIdx.x & 7u], 1u);  // reader block index mod 8 (blocks go round-robin over the 8 XCDs)
    }
}
int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
    const uint32_t n = argc > 2 ? (uint32_t)atoi(argv[2]) : 20000u;  // ~80 blocks of 256, like a small read-back
    const uint32_t blocks = (n + 255) / 256;
    // mode 0: both kernels on one stream.  mode 1: writer on stream A, host waits for A, reader on stream B (what a library
    // with a side stream and synchronising read-backs does).  mode 2: the same with an event instead of the host wait.
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s, sb, extra[4];
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (auto &x : extra) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint32_t *buf, *bad, *bad_block, *side;
    CK(hipMalloc(&side, 768 * 4));
    CK(hipMalloc(&buf, (size_t)n * 4));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&bad_block, 32));
    CK(hipMemset(buf, 0, (size_t)n * 4));
    CK(hipMemset(bad, 0, 4));
    CK(hipMemset(bad_block, 0, 32));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint64_t iters = 0, host_bad = 0;
    std::vector<uint32_t> h(n);
    CK(hipEventRecord(e0, s));
    if (mode == 5) {
        // what a read-back with a per-call staging buffer does: hipMalloc, a kernel fills it, wait, blocking copy, hipFree --
        // the sizes rotate, so that the address range comes back mapped to other pages
        const bool big = argc > 4;  // 1.2 - 9 MB per call instead of 20 - 260 KB (past the runtime's sub-allocator: real map / unmap)
        uint64_t calls = 0, bad_calls = 0, bad_words = 0;
        uint32_t worst_mod8[8] = {};
        for (;;) {
            for (int k = 0; k < 256; k++, calls++) {
                const uint32_t m = big ? 300000u + (uint32_t)((calls * 7919u) % 2000000u) : 5000u + (uint32_t)((calls * 7919u) % 60000u);
                uint32_t *tmp = nullptr, *by = nullptr;
                CK(hipMalloc(&tmp, (size_t)m * 4));
                if (calls % 3 == 0) CK(hipMalloc(&by, (size_t)(1 + calls % 5) * 400000));
                hipLaunchKernelGGL(k_write, dim3((m + 255) / 256), dim3(256), 0, s, tmp, m, (uint32_t)calls);
                CK(hipStreamSynchronize(s));
                h.resize(m);
                CK(hipMemcpy(h.data(), tmp, (size_t)m * 4, hipMemcpyDeviceToHost));
                uint32_t nb = 0;
                for (uint32_t i = 0; i < m; i++)
                    if (h[i] != i * 2654435761u + (uint32_t)calls) nb++, worst_mod8[(i / 256) & 7u]++;
                bad_words += nb, bad_calls += nb != 0;
                CK(hipFree(tmp));
                if (by) CK(hipFree(by));
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms > seconds * 1e3) break;
        }
        printf("pid %d mode 5: %llu malloc/fill/copy/free calls, %llu with mismatches (%llu words; by writer block mod 8: %u %u %u %u %u %u %u %u)\n",
               (int)getpid(), (unsigned long long)calls, (unsigned long long)bad_calls, (unsigned long long)bad_words, worst_mod8[0],
               worst_mod8[1], worst_mod8[2], worst_mod8[3], worst_mod8[4], worst_mod8[5], worst_mod8[6], worst_mod8[7]);
        return 0;
    }
    for (;;) {
        if (mode >= 4) {  // allocation churn: the buffer (and two bystanders of changing sizes) are freed and allocated again
            static uint32_t *by[2] = {nullptr, nullptr};
            CK(hipFree(buf));
            for (auto &b : by) { if (b) CK(hipFree(b)); b = nullptr; }
            CK(hipMalloc(&by[0], (size_t)(1 + iters % 5) * 300000));
            CK(hipMalloc(&buf, (size_t)n * 4));
            CK(hipMalloc(&by[1], (size_t)(1 + iters % 3) * 2000000));
        }
        for (int k = 0; k < (mode >= 4 ? 4 : 64); k++, iters++) {
            if (mode >= 3)  // ... with four more ACTIVE streams per process (8 processes: more queues in use than the hardware maps at once)
                for (auto &x : extra) hipLaunchKernelGGL(k_write, dim3(3), dim3(256), 0, x, side, 768u, (uint32_t)iters);
            hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, s, buf, n, (uint32_t)iters);
            if (mode == 1 || mode >= 3) CK(hipStreamSynchronize(s));
            if (mode == 2) { CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(sb, ev, 0)); }
            hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, mode ? sb : s, buf, n, (uint32_t)iters, 1u + (uint32_t)(iters % 7), bad, bad_block);
            if (mode == 1 || mode >= 3) CK(hipStreamSynchronize(sb));
            if (mode == 2) { CK(hipEventRecord(ev, sb)); CK(hipStreamWaitEvent(s, ev, 0)); }
        }
        // ... and the host's view: stream sync, then a blocking copy (what a read-back does)
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(sb));
        CK(hipMemcpy(h.data(), buf, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) host_bad += h[i] != i * 2654435761u + (uint32_t)(iters - 1);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms > seconds * 1e3) break;
    }
    uint32_t hb = 0, hbb[8];
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hbb, bad_block, 32, hipMemcpyDeviceToHost));
    printf("pid %d mode %d: %llu write/read pairs, device-side mismatches %u (by reader block mod 8: %u %u %u %u %u %u %u %u), host-side mismatches %llu\n",
           (int)getpid(), mode, (unsigned long long)iters, hb, hbb[0], hbb[1], hbb[2], hbb[3], hbb[4], hbb[5], hbb[6], hbb[7],
           (unsigned long long)host_bad);
    return 0;
}
