// inplace.hip -- what an in-place (ring) update of a FIFO segment could reach, against the ping-pong shape.
// Planes as in fw_device.h: Q0 Q1 Q2 Q3 (float4, read), Q5 Q6 (float4) + S4 (float), capacity C slots per plane.
//   hipcc --offload-arch=gfx950 -O3 tools/inplace.hip -o tools/inplace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float work(float x, int K) {
#pragma unroll 1
    for (int k = 0; k < K; k++) x = x * 1.0001f + 0.5f;
    return x;
}

// RD: bit i = read plane i (0..3);  WR: bit i = write plane i (0..3), bit 4 = Q5, bit 5 = Q6, bit 6 = S4.
// in == out for in-place; shift: out slot j reads in slot j + shift (aligned stores, misaligned loads)
// the in-place shapes again with non-temporal accesses: NT 1 = the planes nobody reads back (Q5 Q6 S4), 2 = everything
typedef float f4v __attribute__((ext_vector_type(4)));
template <int RD, int WR, int NT>
__global__ __launch_bounds__(256) void k_upd_nt(char* __restrict__ buf, uint32_t n, uint32_t C, int K) {
    constexpr int R = 4;
    const uint32_t base = blockIdx.x * 256 * R;
    f4v q[4][R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = base + r * 256 + threadIdx.x;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            q[p][r] = f4v{1.f, 2.f, 3.f, 4.f};
            if ((RD >> p & 1) && i < n) {
                const f4v* a = (const f4v*)(buf + (size_t)16 * p * C) + i;
                q[p][r] = NT == 2 ? __builtin_nontemporal_load(a) : *a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            f4v a = q[0][r], b = q[1][r], c = q[2][r], d = q[3][r];
            a.x = work(a.x + b.x, K);
            f4v e = {a.x + b.x, a.y * c.y, d.z, a.w}, f = {b.w, c.x, d.y, e.x};
#define ST(cond, off, val, nt) if (cond) { f4v* o = (f4v*)(buf + (size_t)(off) * C) + i; if (nt) __builtin_nontemporal_store(val, o); else *o = val; }
            ST(WR & 1, 0, a, NT == 2) ST(WR & 2, 16, b, NT == 2) ST(WR & 4, 32, c, NT == 2) ST(WR & 8, 48, d, NT == 2)
            ST(WR & 16, 64, e, NT >= 1) ST(WR & 32, 80, f, NT >= 1)
            if (WR & 64) { float* o = (float*)(buf + (size_t)96 * C) + i; if (NT >= 1) __builtin_nontemporal_store(e.y, o); else *o = e.y; }
        }
    }
}


// round 6 question: Q1 = {velocity, initial_scale} and Q3 = {angular velocity, lifetime} carry a constant in .w -- as packed float3
// planes (12 B per lane, dwordx3) an in-place update moves 112 instead of 128 B (spin) / 56 instead of 64 (no spin).  Does the time follow?
typedef float f3v __attribute__((ext_vector_type(3)));
template <int SPIN, int NT>
__global__ __launch_bounds__(256) void k_upd_v3(char* __restrict__ buf, uint32_t n, uint32_t C, int K) {
    constexpr int R = 4;
    const uint32_t base = blockIdx.x * 256 * R;
    f4v q0[R], q2[R];
    f3v q1[R], q3[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = min(base + r * 256 + threadIdx.x, n - 1);
        const f4v* a0 = (const f4v*)(buf) + i;
        const f3v* a1 = (const f3v*)(buf + (size_t)16 * C) + i;
        q0[r] = NT ? __builtin_nontemporal_load(a0) : *a0;
        q1[r] = NT ? __builtin_nontemporal_load(a1) : *a1;
        if (SPIN) {
            const f4v* a2 = (const f4v*)(buf + (size_t)32 * C) + i;
            const f3v* a3 = (const f3v*)(buf + (size_t)48 * C) + i;
            q2[r] = NT ? __builtin_nontemporal_load(a2) : *a2;
            q3[r] = NT ? __builtin_nontemporal_load(a3) : *a3;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            f4v a = q0[r]; f3v b = q1[r];
            a.x = work(a.x + b.x, K); b.y += a.y;
            f4v* o0 = (f4v*)(buf) + i; f3v* o1 = (f3v*)(buf + (size_t)16 * C) + i;
            if (NT) { __builtin_nontemporal_store(a, o0); __builtin_nontemporal_store(b, o1); } else { *o0 = a; *o1 = b; }
            if (SPIN) {
                f4v c = q2[r]; f3v d = q3[r];
                c.x += a.z; d.z += c.y;
                f4v* o2 = (f4v*)(buf + (size_t)32 * C) + i; f3v* o3 = (f3v*)(buf + (size_t)48 * C) + i;
                if (NT) { __builtin_nontemporal_store(c, o2); __builtin_nontemporal_store(d, o3); } else { *o2 = c; *o3 = d; }
            }
        }
    }
}


// ... and the same question with Q1 / Q3 as THREE 4-byte planes each (dword per lane)
template <int SPIN, int NT>
__global__ __launch_bounds__(256) void k_upd_s3(char* __restrict__ buf, uint32_t n, uint32_t C, int K) {
    constexpr int R = 4;
    const uint32_t base = blockIdx.x * 256 * R;
    f4v q0[R], q2[R];
    float q1[R][3], q3[R][3];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = min(base + r * 256 + threadIdx.x, n - 1);
        const f4v* a0 = (const f4v*)(buf) + i;
        q0[r] = NT ? __builtin_nontemporal_load(a0) : *a0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float* a1 = (const float*)(buf + (size_t)(16 + 4 * c) * C) + i;
            q1[r][c] = NT ? __builtin_nontemporal_load(a1) : *a1;
        }
        if (SPIN) {
            const f4v* a2 = (const f4v*)(buf + (size_t)32 * C) + i;
            q2[r] = NT ? __builtin_nontemporal_load(a2) : *a2;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float* a3 = (const float*)(buf + (size_t)(48 + 4 * c) * C) + i;
                q3[r][c] = NT ? __builtin_nontemporal_load(a3) : *a3;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            f4v a = q0[r];
            a.x = work(a.x + q1[r][0], K); q1[r][1] += a.y; q1[r][2] += a.z; q1[r][0] += a.w;
            f4v* o0 = (f4v*)(buf) + i;
            if (NT) __builtin_nontemporal_store(a, o0); else *o0 = a;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float* o1 = (float*)(buf + (size_t)(16 + 4 * c) * C) + i;
                if (NT) __builtin_nontemporal_store(q1[r][c], o1); else *o1 = q1[r][c];
            }
            if (SPIN) {
                f4v cc = q2[r];
                cc.x += a.z; q3[r][2] += cc.y; q3[r][0] += cc.z; q3[r][1] += cc.w;
                f4v* o2 = (f4v*)(buf + (size_t)32 * C) + i;
                if (NT) __builtin_nontemporal_store(cc, o2); else *o2 = cc;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float* o3 = (float*)(buf + (size_t)(48 + 4 * c) * C) + i;
                    if (NT) __builtin_nontemporal_store(q3[r][c], o3); else *o3 = q3[r][c];
                }
            }
        }
    }
}

template <int R, int RD, int WR, bool STSHIFT = false>
__global__ __launch_bounds__(256) void k_upd(const char* __restrict__ in, char* __restrict__ out, uint32_t n, uint32_t C, uint32_t shift, int K) {
    const uint32_t base = blockIdx.x * 256 * R;
    float4 q[4][R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = base + r * 256 + threadIdx.x;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            q[p][r] = make_float4(1.f, 2.f, 3.f, 4.f);
            if ((RD >> p & 1) && i < n) q[p][r] = ((const float4*)(in + (size_t)16 * p * C))[i + (STSHIFT ? 0u : shift)];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            if (STSHIFT) i += shift;
            float4 a = q[0][r], b = q[1][r], c = q[2][r], d = q[3][r];
            a.x = work(a.x + b.x, K);
            float4 e = make_float4(a.x + b.x, a.y * c.y, d.z, a.w);
            float4 f = make_float4(b.w, c.x, d.y, e.x);
            if (WR & 1) ((float4*)(out))[i] = a;
            if (WR & 2) ((float4*)(out + (size_t)16 * C))[i] = b;
            if (WR & 4) ((float4*)(out + (size_t)32 * C))[i] = c;
            if (WR & 8) ((float4*)(out + (size_t)48 * C))[i] = d;
            if (WR & 16) ((float4*)(out + (size_t)64 * C))[i] = e;
            if (WR & 32) ((float4*)(out + (size_t)80 * C))[i] = f;
            if (WR & 64) ((float*)(out + (size_t)96 * C))[i] = e.y;
        }
    }
}

// the same update with a TILED layout: the seven planes of a tile's 1024 slots are contiguous (100 KB per tile), so a
// workgroup reads and writes ONE region instead of seven streams 16 C bytes apart
template <int RD, int WR>
__global__ __launch_bounds__(256) void k_upd_tiled(const char* __restrict__ in, char* __restrict__ out, uint32_t n, int K) {
    constexpr int R = 4;
    const size_t tb = (size_t)blockIdx.x * 102400u;  // 1024 * 100 B
    float4 q[4][R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t l = r * 256 + threadIdx.x;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            q[p][r] = make_float4(1.f, 2.f, 3.f, 4.f);
            if ((RD >> p & 1) && blockIdx.x * 1024u + l < n) q[p][r] = ((const float4*)(in + tb + (size_t)p * 16384u))[l];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t l = r * 256 + threadIdx.x;
        if (blockIdx.x * 1024u + l < n) {
            float4 a = q[0][r], b = q[1][r], c = q[2][r], d = q[3][r];
            a.x = work(a.x + b.x, K);
            float4 e = make_float4(a.x + b.x, a.y * c.y, d.z, a.w);
            float4 f = make_float4(b.w, c.x, d.y, e.x);
            if (WR & 1) ((float4*)(out + tb))[l] = a;
            if (WR & 2) ((float4*)(out + tb + 16384u))[l] = b;
            if (WR & 4) ((float4*)(out + tb + 32768u))[l] = c;
            if (WR & 8) ((float4*)(out + tb + 49152u))[l] = d;
            if (WR & 16) ((float4*)(out + tb + 65536u))[l] = e;
            if (WR & 32) ((float4*)(out + tb + 81920u))[l] = f;
            if (WR & 64) ((float*)(out + tb + 98304u))[l] = e.y;
        }
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

int main() {
    for (uint32_t n : {1000000u, 4000000u, 16000000u}) {
        const uint32_t C = (n + 1023) / 1024 * 1024 + 262144;
        const size_t pb = (size_t)100 * C;
        char *p0, *p1; CK(hipMalloc(&p0, pb)); CK(hipMalloc(&p1, pb));
        CK(hipMemset(p0, 0, pb)); CK(hipMemset(p1, 0, pb));
        const dim3 g((n + 1023) / 1024), b(256);
        for (int K : {0, 150}) {
            double t;
#define RUN_NT(RD, WR, NT, bytes, tag) \
            t = timeit([&](int i) { hipLaunchKernelGGL((k_upd_nt<RD, WR, NT>), g, b, 0, 0, p0, n, C, K); }, 50); \
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, tag, t * 1e6, (double)(bytes) * n / t / 1e9);
            // round 6: the shapes left once scale and colours are the readers' business (FW_TYPE_DERIVED for every type)
            RUN_NT(15, 15, 0, 128, "r6: ONE buffer in place r4 w4 (128 B)")
            RUN_NT(15, 15, 2, 128, "  ... everything non-temporal")
            RUN_NT(3, 3, 0, 64, "r6: ONE buffer in place r Q0 Q1 w Q0 Q1 (64 B)")
            RUN_NT(3, 3, 2, 64, "  ... everything non-temporal")
#define RUN_V3(SPIN, NT, bytes, tag) \
            t = timeit([&](int i) { hipLaunchKernelGGL((k_upd_v3<SPIN, NT>), g, b, 0, 0, p0, n, C, K); }, 50); \
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, tag, t * 1e6, (double)(bytes) * n / t / 1e9);
            RUN_V3(1, 0, 112, "r6: Q0 Q2 float4 + Q1 Q3 float3 in place (112 B)")
            RUN_V3(1, 1, 112, "  ... all non-temporal (112 B)")
            RUN_V3(0, 0, 56, "r6: Q0 float4 + Q1 float3 in place (56 B)")
            RUN_V3(0, 1, 56, "  ... all non-temporal (56 B)")
#define RUN_S3(SPIN, NT, bytes, tag) \
            t = timeit([&](int i) { hipLaunchKernelGGL((k_upd_s3<SPIN, NT>), g, b, 0, 0, p0, n, C, K); }, 50); \
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, tag, t * 1e6, (double)(bytes) * n / t / 1e9);
            RUN_S3(1, 0, 112, "r6: Q0 Q2 float4 + Q1 Q3 as 3 scalar planes (112 B)")
            RUN_S3(1, 1, 112, "  ... all non-temporal, scalar planes (112 B)")
            RUN_S3(0, 0, 56, "r6: Q0 float4 + Q1 as 3 scalar planes (56 B)")
            RUN_S3(0, 1, 56, "  ... all non-temporal, scalar planes (56 B)")
            RUN_NT(15, 127, 0, 164, "ONE buffer in place r4 w7 (164 B)")
            RUN_NT(15, 127, 1, 164, "  ... Q5 Q6 S4 non-temporal")
            RUN_NT(15, 127, 2, 164, "  ... everything non-temporal")
            RUN_NT(3, 115, 0, 100, "ONE buffer in place r Q0 Q1 w Q0 Q1 Q5 Q6 S4 (100 B)")
            RUN_NT(3, 115, 1, 100, "  ... Q5 Q6 S4 non-temporal")
            RUN_NT(3, 115, 2, 100, "  ... everything non-temporal")
#define RUN(RD, WR, inplace, shift, bytes, tag) \
            t = timeit([&](int i) { char* a = (i & 1) ? p1 : p0; char* o = (inplace) ? a : ((i & 1) ? p0 : p1); \
                hipLaunchKernelGGL((k_upd<4, RD, WR>), g, b, 0, 0, a, o, n, C, shift, K); }, 50); \
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, tag, t * 1e6, (double)(bytes) * n / t / 1e9);
            t = timeit([&](int i) { char* a = (i & 1) ? p1 : p0; char* o = (i & 1) ? p0 : p1;
                hipLaunchKernelGGL((k_upd_tiled<15, 127>), g, b, 0, 0, a, o, n, K); }, 50);
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, "TILED layout, ping-pong r4 w7 (164 B)", t * 1e6, 164.0 * n / t / 1e9);
            t = timeit([&](int i) { char* a = (i & 1) ? p1 : p0;
                hipLaunchKernelGGL((k_upd_tiled<15, 127>), g, b, 0, 0, a, a, n, K); }, 50);
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, "TILED layout, in place r4 w7 (164 B)", t * 1e6, 164.0 * n / t / 1e9);
            t = timeit([&](int i) { char* a = (i & 1) ? p1 : p0;
                hipLaunchKernelGGL((k_upd_tiled<3, 83>), g, b, 0, 0, a, a, n, K); }, 50);
            printf("n=%8u K=%3d %-44s: %8.2f us  %7.1f GB/s moved\n", n, K, "TILED layout, in place r Q0 Q1 w Q0 Q1 Q5 S4 (84 B)", t * 1e6, 84.0 * n / t / 1e9);
            RUN(15, 127, false, 0u, 164, "ping-pong r4 w7 (164 B)")
            RUN(15, 127, false, 16667u, 164, "ping-pong shifted loads, aligned stores")
            for (uint32_t sh : {1u, 3u, 4u, 8u, 16667u}) {
                t = timeit([&](int i) { char* a = (i & 1) ? p1 : p0; char* o = (i & 1) ? p0 : p1;
                    hipLaunchKernelGGL((k_upd<4, 15, 127, true>), g, b, 0, 0, a, o, n, C, sh, K); }, 50);
                printf("n=%8u K=%3d ping-pong aligned loads, stores shifted by %5u : %8.2f us  %7.1f GB/s moved\n", n, K, sh, t * 1e6, 164.0 * n / t / 1e9);
            }
            RUN(15, 127, true, 0u, 164, "in place r4 w7 (164 B)")
            RUN(15, 115, true, 0u, 132, "in place r4 w Q0 Q1 Q5 Q6 S4 (132 B)")
            RUN(15, 83, true, 0u, 116, "in place r4 w Q0 Q1 Q5 S4 (116 B)")
            RUN(3, 115, true, 0u, 100, "in place r Q0 Q1 w Q0 Q1 Q5 Q6 S4 (100 B)")
            RUN(3, 83, true, 0u, 84, "in place r Q0 Q1 w Q0 Q1 Q5 S4 (84 B)")
            RUN(3, 3, true, 0u, 64, "in place r Q0 Q1 w Q0 Q1 (64 B)")
        }
        CK(hipFree(p0)); CK(hipFree(p1));
    }
    return 0;
}
