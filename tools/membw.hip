// membw.hip -- HBM streaming microbenchmark for gfx950: what a float4 read+write stream can reach,
// by access shape.  Used to set the "measured roofline" line in DESIGN.md and to choose the update
// kernel's load/store shape.   hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld(const float4* p) {
    v4f v = __builtin_nontemporal_load((const v4f*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_st(float4 x, float4* p) {
    v4f v = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(v, (v4f*)p);
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t j = i + (size_t)u * 256;
            if (j < n4) v[u] = NT ? nt_ld(&src[j]) : src[j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t j = i + (size_t)u * 256;
            if (j < n4) { if (NT) nt_st(v[u], &dst[j]); else dst[j] = v[u]; }
        }
    }
}

// contiguous chunk per workgroup (each workgroup streams its own region front to back) instead of a grid-strided
// interleave: fewer DRAM pages open at once
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t j = i + (size_t)u * 256;
            if (j < hi) v[u] = NT ? nt_ld(&src[j]) : src[j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t j = i + (size_t)u * 256;
            if (j < hi) { if (NT) nt_st(v[u], &dst[j]); else dst[j] = v[u]; }
        }
    }
}

// one tile per block (no grid stride): the update kernel's shape. 4 input planes, 7 output planes.
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_planes(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C) {
    const uint32_t base = blockIdx.x * 256 * R;
    float4 a[R], b[R], c[R], d[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            a[r] = ((const float4*)(in))[i];
            b[r] = ((const float4*)(in + (size_t)16 * C))[i];
            c[r] = ((const float4*)(in + (size_t)32 * C))[i];
            d[r] = ((const float4*)(in + (size_t)48 * C))[i];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            float4 e = make_float4(a[r].x + b[r].x, a[r].y * c[r].y, d[r].z, a[r].w);
            float4 f = make_float4(b[r].w, c[r].x, d[r].y, e.x);
            if (NT) {
                nt_st(a[r], &((float4*)(out))[i]);
                nt_st(b[r], &((float4*)(out + (size_t)16 * C))[i]);
                nt_st(c[r], &((float4*)(out + (size_t)32 * C))[i]);
                nt_st(d[r], &((float4*)(out + (size_t)48 * C))[i]);
                nt_st(e, &((float4*)(out + (size_t)64 * C))[i]);
                nt_st(f, &((float4*)(out + (size_t)80 * C))[i]);
                __builtin_nontemporal_store(e.y, &((float*)(out + (size_t)96 * C))[i]);
            } else {
                ((float4*)(out))[i] = a[r];
                ((float4*)(out + (size_t)16 * C))[i] = b[r];
                ((float4*)(out + (size_t)32 * C))[i] = c[r];
                ((float4*)(out + (size_t)48 * C))[i] = d[r];
                ((float4*)(out + (size_t)64 * C))[i] = e;
                ((float4*)(out + (size_t)80 * C))[i] = f;
                ((float*)(out + (size_t)96 * C))[i] = e.y;
            }
        }
    }
}

// same as k_planes<4,false> but occupancy-limited through dynamic LDS (bytes per block chosen by the host)
__global__ __launch_bounds__(256) void k_planes_occ(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C) {
    extern __shared__ float dummy[];
    constexpr int R = 4;
    const uint32_t base = blockIdx.x * 256 * R;
    float4 a[R], b[R], c[R], d[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            a[r] = ((const float4*)(in))[i];
            b[r] = ((const float4*)(in + (size_t)16 * C))[i];
            c[r] = ((const float4*)(in + (size_t)32 * C))[i];
            d[r] = ((const float4*)(in + (size_t)48 * C))[i];
        }
    }
    if (n == 0xFFFFFFFFu) dummy[threadIdx.x] = a[0].x;
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            float4 e = make_float4(a[r].x + b[r].x, a[r].y * c[r].y, d[r].z, a[r].w);
            float4 f = make_float4(b[r].w, c[r].x, d[r].y, e.x);
            ((float4*)(out))[i] = a[r];
            ((float4*)(out + (size_t)16 * C))[i] = b[r];
            ((float4*)(out + (size_t)32 * C))[i] = c[r];
            ((float4*)(out + (size_t)48 * C))[i] = d[r];
            ((float4*)(out + (size_t)64 * C))[i] = e;
            ((float4*)(out + (size_t)80 * C))[i] = f;
            ((float*)(out + (size_t)96 * C))[i] = e.y;
        }
    }
}

// structural twin of fw_k_update v3: planes 0/3 for the whole tile first -> LDS -> barrier -> rolled round loop
// that reads them back, prefetches planes 1/2 one round ahead and stores 7 planes.  MODE bit0: skip LDS
// staging (keep regs), bit1: all 16 loads up front, bit2: no barrier
template <int MODE>
__global__ __launch_bounds__(256) void k_struct(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C) {
    constexpr int R = 4;
    __shared__ float4 s0[1024], s3[1024];
    __shared__ uint32_t s_w[4][4];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * 1024;
    const float4* p0 = (const float4*)in; const float4* p1 = (const float4*)(in + (size_t)16 * C);
    const float4* p2 = (const float4*)(in + (size_t)32 * C); const float4* p3 = (const float4*)(in + (size_t)48 * C);
    float4 t0[R], t3[R];
#pragma unroll
    for (int r = 0; r < R; r++) { uint32_t i = base + r * 256 + tid; if (i < n) { t0[r] = p0[i]; t3[r] = p3[i]; } }
#pragma unroll
    for (int r = 0; r < R; r++) { uint32_t i = base + r * 256 + tid; if (i < n) { s0[r * 256 + tid] = t0[r]; s3[r * 256 + tid] = t3[r]; } }
    float4 q1c = make_float4(0,0,0,0), q2c = q1c;
    if (base + tid < n) { q1c = p1[base + tid]; q2c = p2[base + tid]; }
#pragma unroll
    for (int r = 0; r < R; r++) {
        unsigned long long m = __ballot(s0[r * 256 + tid].w + 0.016f < s3[r * 256 + tid].w + 1e9f);
        if ((tid & 63) == 0) s_w[r][tid >> 6] = __popcll(m);
    }
    if (!(MODE & 4)) __syncthreads();
    uint32_t run = 0;
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + tid;
        float4 q1n = make_float4(0,0,0,0), q2n = q1n;
        if (r + 1 < R && i + 256 < n) { q1n = p1[i + 256]; q2n = p2[i + 256]; }
        float4 a = s0[r * 256 + tid], d = s3[r * 256 + tid];
        for (int w = 0; w < 4; w++) run += s_w[r][w];
        if (i < n) {
            uint32_t o = i + (run & 0);
            float4 e = make_float4(a.x + q1c.x, a.y * q2c.y, d.z, a.w);
            float4 f = make_float4(q1c.w, q2c.x, d.y, e.x);
            ((float4*)(out))[o] = a;
            ((float4*)(out + (size_t)16 * C))[o] = q1c;
            ((float4*)(out + (size_t)32 * C))[o] = q2c;
            ((float4*)(out + (size_t)48 * C))[o] = d;
            ((float4*)(out + (size_t)64 * C))[o] = e;
            ((float4*)(out + (size_t)80 * C))[o] = f;
            ((float*)(out + (size_t)96 * C))[o] = e.y;
        }
        q1c = q1n; q2c = q2n;
    }
}

// streaming twin: per round, prefetch the next round's 4 planes, then store the 7 output planes of the current one
template <int KIND>
__device__ __forceinline__ void st_kind(float4* p, float4 v) {
    if (KIND == 0) { *p = v; }
    else if (KIND == 1) { nt_st(v, p); }
    else if (KIND == 2) { v4f x = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory"); }
    else { v4f x = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(x) : "memory"); }
}
template <int R, int KIND>
__global__ __launch_bounds__(256) void k_stream_st(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C) {
    const uint32_t tid = threadIdx.x, base = blockIdx.x * 256 * R;
    const float4* p0 = (const float4*)in; const float4* p1 = (const float4*)(in + (size_t)16 * C);
    const float4* p2 = (const float4*)(in + (size_t)32 * C); const float4* p3 = (const float4*)(in + (size_t)48 * C);
    float4 a = make_float4(0,0,0,0), b = a, c = a, d = a;
    if (base + tid < n) { a = p0[base + tid]; b = p1[base + tid]; c = p2[base + tid]; d = p3[base + tid]; }
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + tid;
        float4 an = a, bn = b, cn = c, dn = d;
        if (r + 1 < R && i + 256 < n) { an = p0[i + 256]; bn = p1[i + 256]; cn = p2[i + 256]; dn = p3[i + 256]; }
        if (i < n) {
            float4 e = make_float4(a.x + b.x, a.y * c.y, d.z, a.w);
            float4 f = make_float4(b.w, c.x, d.y, e.x);
            st_kind<KIND>(&((float4*)(out))[i], a);
            st_kind<KIND>(&((float4*)(out + (size_t)16 * C))[i], b);
            st_kind<KIND>(&((float4*)(out + (size_t)32 * C))[i], c);
            st_kind<KIND>(&((float4*)(out + (size_t)48 * C))[i], d);
            st_kind<KIND>(&((float4*)(out + (size_t)64 * C))[i], e);
            st_kind<KIND>(&((float4*)(out + (size_t)80 * C))[i], f);
            ((float*)(out + (size_t)96 * C))[i] = e.y;
        }
        a = an; b = bn; c = cn; d = dn;
    }
}

template <int R, bool BAR = false>
__global__ __launch_bounds__(256) void k_stream(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C) {
    __shared__ uint32_t s_c[2][4];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * 256 * R;
    uint32_t run = 0;
    const float4* p0 = (const float4*)in; const float4* p1 = (const float4*)(in + (size_t)16 * C);
    const float4* p2 = (const float4*)(in + (size_t)32 * C); const float4* p3 = (const float4*)(in + (size_t)48 * C);
    float4 a = make_float4(0,0,0,0), b = a, c = a, d = a;
    if (base + tid < n) { a = p0[base + tid]; b = p1[base + tid]; c = p2[base + tid]; d = p3[base + tid]; }
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + tid;
        float4 an = a, bn = b, cn = c, dn = d;
        if (r + 1 < R && i + 256 < n) { an = p0[i + 256]; bn = p1[i + 256]; cn = p2[i + 256]; dn = p3[i + 256]; }
        if (BAR) {  // per-round cross-wave rank exchange (double-buffered LDS, one barrier per round)
            unsigned long long m = __ballot(i < n && a.w + 0.016f < d.w + 1e9f);
            if ((tid & 63) == 0) s_c[r & 1][tid >> 6] = __popcll(m);
            __syncthreads();
            for (int w = 0; w < 4; w++) run += s_c[r & 1][w];
        }
        if (i < n) {
            i += (run & 0);
            float4 e = make_float4(a.x + b.x, a.y * c.y, d.z, a.w);
            float4 f = make_float4(b.w, c.x, d.y, e.x);
            ((float4*)(out))[i] = a;
            ((float4*)(out + (size_t)16 * C))[i] = b;
            ((float4*)(out + (size_t)32 * C))[i] = c;
            ((float4*)(out + (size_t)48 * C))[i] = d;
            ((float4*)(out + (size_t)64 * C))[i] = e;
            ((float4*)(out + (size_t)80 * C))[i] = f;
            ((float*)(out + (size_t)96 * C))[i] = e.y;
        }
        a = an; b = bn; c = cn; d = dn;
    }
}

// stream twin with the launch-span ring of the real kernel (first workgroup start / last workgroup end per launch)
template <int R>
__global__ __launch_bounds__(256) void k_stream_ts(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C,
                                                   unsigned long long* ring, uint32_t epoch) {
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t tid = threadIdx.x, base = blockIdx.x * 256 * R;
    const float4* p0 = (const float4*)in; const float4* p1 = (const float4*)(in + (size_t)16 * C);
    const float4* p2 = (const float4*)(in + (size_t)32 * C); const float4* p3 = (const float4*)(in + (size_t)48 * C);
    float4 a = make_float4(0,0,0,0), b = a, c = a, d = a;
    if (base + tid < n) { a = p0[base + tid]; b = p1[base + tid]; c = p2[base + tid]; d = p3[base + tid]; }
    float acc = 0.f;
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + tid;
        float4 an = a, bn = b, cn = c, dn = d;
        if (r + 1 < R && i + 256 < n) { an = p0[i + 256]; bn = p1[i + 256]; cn = p2[i + 256]; dn = p3[i + 256]; }
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
    __syncthreads();
    if (tid == 0) {
        const unsigned long long tsE = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(acc != acc ? 1 : 0);
        atomicMin(&ring[(epoch & 255u) * 2u], ts0);
        atomicMax(&ring[(epoch & 255u) * 2u + 1u], tsE);
    }
}

template <typename F>
double timeit(F f, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f(i);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) f(i);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / iters;
}

int main(int argc, char** argv) {
    size_t mb = argc > 1 ? atol(argv[1]) : 256;
    size_t bytes = mb << 20;
    char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    size_t n4 = bytes / 16;
    if (argc > 2 && !strcmp(argv[2], "copy")) {  // copy sweep only (what fw_ctx_measure_copy_bandwidth should use)
        printf("copy sweep %zu MiB -> %zu MiB (GB/s = read+write bytes)\n", mb, mb);
        for (int blocks : {256, 512, 1024, 2048, 4096, 8192, 16384}) {
            double t;
#define RUNK(K, tag) t = timeit([&](int i) { hipLaunchKernelGGL((K), dim3(blocks), dim3(256), 0, 0, (const float4*)((i&1)?b:a), (float4*)((i&1)?a:b), n4); }, 20); \
            printf("  %-22s blocks=%5d : %8.1f GB/s\n", tag, blocks, 2.0 * bytes / t / 1e9);
            RUNK((k_copy<1, false>), "stride U=1")
            RUNK((k_copy<4, false>), "stride U=4")
            RUNK((k_copy<4, true>), "stride U=4 nt")
            RUNK((k_copy_chunk<1, false>), "chunk U=1")
            RUNK((k_copy_chunk<4, false>), "chunk U=4")
            RUNK((k_copy_chunk<4, true>), "chunk U=4 nt")
            RUNK((k_copy_chunk<8, false>), "chunk U=8")
        }
        {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; i++) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; i++) CK(hipMemcpyAsync((i&1)?a:b, (i&1)?b:a, bytes, hipMemcpyDeviceToDevice, 0));
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  hipMemcpyAsync D2D                 : %8.1f GB/s\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
        }
        return 0;
    }
    printf("copy %zu MiB -> %zu MiB (GB/s = read+write bytes)\n", mb, mb);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        double t;
        t = timeit([&](int i) { hipLaunchKernelGGL((k_copy<1, false>), dim3(blocks), dim3(256), 0, 0, (const float4*)((i&1)?b:a), (float4*)((i&1)?a:b), n4); }, 20);
        printf("  U=1 blocks=%5d        : %8.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_copy<4, false>), dim3(blocks), dim3(256), 0, 0, (const float4*)((i&1)?b:a), (float4*)((i&1)?a:b), n4); }, 20);
        printf("  U=4 blocks=%5d        : %8.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_copy<4, true>), dim3(blocks), dim3(256), 0, 0, (const float4*)((i&1)?b:a), (float4*)((i&1)?a:b), n4); }, 20);
        printf("  U=4 blocks=%5d nt     : %8.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_copy<8, false>), dim3(blocks), dim3(256), 0, 0, (const float4*)((i&1)?b:a), (float4*)((i&1)?a:b), n4); }, 20);
        printf("  U=8 blocks=%5d        : %8.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
    }
    // update-kernel shape: N particles, 64 B in / 100 B out per particle, ping-pong
    for (uint32_t n : {1000000u, 4000000u, 16000000u}) {
        uint32_t C = (n + 1023) / 1024 * 1024 + 262144;
        size_t pb = (size_t)100 * C;
        char *p0, *p1; CK(hipMalloc(&p0, pb)); CK(hipMalloc(&p1, pb));
        CK(hipMemset(p0, 0, pb)); CK(hipMemset(p1, 0, pb));
        double t;
        t = timeit([&](int i) { hipLaunchKernelGGL((k_planes<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("planes n=%8u R=4        : %7.2f us  %8.1f GB/s (164 B/particle)\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_planes<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("planes n=%8u R=4 nt     : %7.2f us  %8.1f GB/s\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_planes<2, false>), dim3((n + 511) / 512), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("planes n=%8u R=2        : %7.2f us  %8.1f GB/s\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_planes<1, false>), dim3((n + 255) / 256), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("planes n=%8u R=1        : %7.2f us  %8.1f GB/s\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_planes<8, false>), dim3((n + 2047) / 2048), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("planes n=%8u R=8        : %7.2f us  %8.1f GB/s\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_struct<0>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("struct v3 twin n=%8u      : %7.2f us  %8.1f GB/s\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_struct<4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("struct v3 twin nobarrier    : %7.2f us  %8.1f GB/s\n", t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream<4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin R=4  n=%8u    : %7.2f us  %8.1f GB/s\n", n, t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream_st<4, 0>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin stores plain      : %7.2f us\n", t * 1e6);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream_st<4, 1>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin stores nt         : %7.2f us\n", t * 1e6);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream_st<4, 2>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin stores sc1        : %7.2f us\n", t * 1e6);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream_st<4, 3>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin stores sc0 sc1    : %7.2f us\n", t * 1e6);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin R=4 +barrier/round: %7.2f us  %8.1f GB/s\n", t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream<16, true>), dim3((n + 4095) / 4096), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin R=16 +barrier/round: %7.2f us  %8.1f GB/s\n", t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream<8>), dim3((n + 2047) / 2048), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin R=8               : %7.2f us  %8.1f GB/s\n", t * 1e6, 164.0 * n / t / 1e9);
        t = timeit([&](int i) { hipLaunchKernelGGL((k_stream<16>), dim3((n + 4095) / 4096), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
        printf("stream twin R=16              : %7.2f us  %8.1f GB/s\n", t * 1e6, 164.0 * n / t / 1e9);
        {
            unsigned long long* ring; CK(hipMalloc(&ring, 512 * 8));
            std::vector<unsigned long long> h(512);
            for (int i = 0; i < 256; i++) h[2 * i] = ~0ull, h[2 * i + 1] = 0ull;
            CK(hipMemcpy(ring, h.data(), 512 * 8, hipMemcpyHostToDevice));
            t = timeit([&](int i) { hipLaunchKernelGGL((k_stream_ts<4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C, ring, (uint32_t)(i + 64)); }, 50);
            CK(hipMemcpy(h.data(), ring, 512 * 8, hipMemcpyDeviceToHost));
            double span = 0, gap = 0; int cnt = 0;
            for (int i = 70; i < 110; i++) { span += (double)(h[2*i+1] - h[2*i]) / 100.0; gap += (double)((long long)h[2*(i+1)] - (long long)h[2*i+1]) / 100.0; cnt++; }
            printf("stream twin +span ring        : %7.2f us/launch; in-kernel span %.2f us, gap to next launch %.2f us\n", t * 1e6, span / cnt, gap / cnt);
            CK(hipFree(ring));
        }
        for (int blocks_per_cu : {1, 2, 3, 4, 6, 8}) {
            size_t lds = 160 * 1024 / blocks_per_cu - 1024;
            if (lds > 64 * 1024) { CK(hipFuncSetAttribute((const void*)k_planes_occ, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
            t = timeit([&](int i) { hipLaunchKernelGGL(k_planes_occ, dim3((n + 1023) / 1024), dim3(256), lds, 0, (i&1)?p1:p0, (i&1)?p0:p1, n, C); }, 50);
            printf("planes n=%8u R=4 %d blocks/CU : %7.2f us  %8.1f GB/s\n", n, blocks_per_cu, t * 1e6, 164.0 * n / t / 1e9);
        }
        CK(hipFree(p0)); CK(hipFree(p1));
    }
    return 0;
}
