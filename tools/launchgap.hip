// launchgap.hip -- do back-to-back launches of a ~20 us streaming kernel keep a steady period?  Looks for periodic
// bubbles in the dispatch path by kernarg size / stream kind / submit pattern.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int PAD>
struct Pad { uint32_t w[PAD]; };

template <int PAD>
__global__ __launch_bounds__(256) void k_stream(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C,
                                                unsigned long long* ring, uint32_t epoch, Pad<PAD> pad) {
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t tid = threadIdx.x, base = blockIdx.x * 1024;
    const float4* p0 = (const float4*)in; const float4* p1 = (const float4*)(in + (size_t)16 * C);
    const float4* p2 = (const float4*)(in + (size_t)32 * C); const float4* p3 = (const float4*)(in + (size_t)48 * C);
    float4 a = make_float4(0,0,0,0), b = a, c = a, d = a;
    if (base + tid < n) { a = p0[base + tid]; b = p1[base + tid]; c = p2[base + tid]; d = p3[base + tid]; }
    float acc = (float)pad.w[PAD - 1];
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        uint32_t i = base + r * 256 + tid;
        float4 an = a, bn = b, cn = c, dn = d;
        if (r + 1 < 4 && i + 256 < n) { an = p0[i + 256]; bn = p1[i + 256]; cn = p2[i + 256]; dn = p3[i + 256]; }
        if (i < n) {
            float4 e = make_float4(a.x + b.x, a.y * c.y, d.z, a.w);
            float4 f = make_float4(b.w, c.x, d.y, e.x);
            ((float4*)(out))[i] = a;
            ((float4*)(out + (size_t)16 * C))[i] = b;
            ((float4*)(out + (size_t)32 * C))[i] = c;
            ((float4*)(out + (size_t)48 * C))[i] = d;
            ((float4*)(out + (size_t)64 * C))[i] = e;
            ((float4*)(out + (size_t)80 * C))[i] = f;
            ((float*)(out + (size_t)96 * C))[i] = e.y;
            acc += e.y;
        }
        a = an; b = bn; c = cn; d = dn;
    }
    if (tid == 0) {
        if (blockIdx.x == 0) ring[(epoch & 255u) * 2u] = ts0;
        if (blockIdx.x == gridDim.x - 1) ring[(epoch & 255u) * 2u + 1u] = __builtin_amdgcn_s_memrealtime() + (acc != acc ? 1 : 0);
    }
}

template <int PAD>
void run(const char* label, hipStream_t s, char* p0, char* p1, uint32_t n, uint32_t C, unsigned long long* ring, int launches) {
    std::vector<unsigned long long> h(512, 0);
    CK(hipMemset(ring, 0, 512 * 8));
    Pad<PAD> pad{};
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < launches; i++)
        hipLaunchKernelGGL((k_stream<PAD>), dim3((n + 1023) / 1024), dim3(256), 0, s, (i & 1) ? p1 : p0, (i & 1) ? p0 : p1, n, C, ring, (uint32_t)i, pad);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), ring, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<double> per;
    for (int k = launches - 200; k < launches - 1; k++) {
        const int i = k & 255, j = (k + 1) & 255;
        per.push_back((double)((long long)h[2 * j] - (long long)h[2 * i]) / 100.0);
    }
    std::vector<double> srt = per; std::sort(srt.begin(), srt.end());
    double mean = 0; for (double v : per) mean += v; mean /= per.size();
    int big = 0; for (double v : per) big += v > srt[srt.size() / 2] * 1.5;
    printf("%-44s period mean %6.2f p50 %6.2f p90 %6.2f max %6.2f  bubbles(>1.5x p50) %d/%zu\n", label, mean, srt[srt.size() / 2], srt[srt.size() * 9 / 10], srt.back(), big, per.size());
}

int main() {
    const uint32_t n = 1000000, C = (n + 1023) / 1024 * 1024 + 262144;
    size_t pb = (size_t)100 * C;
    char *p0, *p1; CK(hipMalloc(&p0, pb)); CK(hipMalloc(&p1, pb));
    CK(hipMemset(p0, 0, pb)); CK(hipMemset(p1, 0, pb));
    unsigned long long* ring; CK(hipMalloc(&ring, 512 * 8));
    hipStream_t nb, blk; CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking)); CK(hipStreamCreate(&blk));
    for (int rep = 0; rep < 2; rep++) {
        run<1>("null stream, 44 B kernarg", 0, p0, p1, n, C, ring, 600);
        run<1>("non-blocking stream, 44 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<1>("blocking stream, 44 B kernarg", blk, p0, p1, n, C, ring, 600);
        run<64>("non-blocking stream, 296 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<256>("non-blocking stream, 1064 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<512>("non-blocking stream, 2088 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<960>("non-blocking stream, 3880 B kernarg", nb, p0, p1, n, C, ring, 600);
    }
    return 0;
}
