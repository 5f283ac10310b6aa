// launchgap.hip -- do back-to-back launches of a ~20 us streaming kernel keep a steady period?  Looks for periodic
// bubbles in the dispatch path by kernarg size / stream kind / submit pattern.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
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

// ---- feature twins of fw_k_update_stream: which structural feature costs what on top of the plain stream ----
// F_CLAMP   unconditional loads at a clamped index (the real kernel's form) instead of predicated ones
// F_COMPACT every wave drops its last lane and the survivors are stored compacted (misaligned 16 B * 63 runs)
// F_TABLE   prologue: 8 x uint4 per lane from a 16 KB table, wave + block reduction, barrier before the rounds
// F_ROUNDB  per-round ballot -> LDS -> barrier -> prefix
// F_SCALAR  dependent scalar chain (pointer -> struct -> pointer) before the first particle load
// F_ALU     ~350 dependent-ish VALU instructions per particle between the loads' arrival and the stores
// F_DEEP    prefetch two rounds ahead instead of one
enum { F_CLAMP = 1, F_COMPACT = 2, F_TABLE = 4, F_ROUNDB = 8, F_SCALAR = 16, F_ALU = 32, F_DEEP = 64 };
struct TwinSeg { const char* in; char* out; uint32_t C, n; };
template <int F>
__global__ __launch_bounds__(256) void k_twin(const TwinSeg* __restrict__ segs, const uint4* __restrict__ table, uint4* __restrict__ table_out,
                                              const char* __restrict__ in_, char* __restrict__ out_, uint32_t n_, uint32_t C_,
                                              unsigned long long* ring, uint32_t epoch, unsigned long long* stamps = nullptr) {
    __shared__ uint32_t s_c[2][4];
    __shared__ uint32_t s_p[4];
    unsigned long long tsP = 0, tsR0 = 0;
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x, base = tile * 1024;
    const uint32_t n_tiles = gridDim.x;
    uint4 fce[8];
    if (F & F_TABLE) {
#pragma unroll
        for (int j = 0; j < 8; j++) fce[j] = table[min(tid + j * 256u, n_tiles - 1u)];
    }
    const char* in = in_; char* out = out_; uint32_t n = n_, C = C_;
    if (F & F_SCALAR) { const TwinSeg* S = &segs[epoch & 1u]; in = S->in; out = S->out; n = S->n; C = S->C; }
    const float4* p0 = (const float4*)in; const float4* p1 = (const float4*)(in + (size_t)16 * C);
    const float4* p2 = (const float4*)(in + (size_t)32 * C); const float4* p3 = (const float4*)(in + (size_t)48 * C);
    const uint32_t lim = min(base + 1024u, n), last = lim - 1u;
    float4 a = make_float4(0,0,0,0), b = a, c = a, d = a;
    if (F & F_CLAMP) { const uint32_t i0 = min(base + tid, last); a = p0[i0]; b = p1[i0]; c = p2[i0]; d = p3[i0]; }
    else if (base + tid < n) { a = p0[base + tid]; b = p1[base + tid]; c = p2[base + tid]; d = p3[base + tid]; }
    uint32_t excl = 0;
    if (F & F_TABLE) {
        uint32_t part = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { const bool in_t = tid + j * 256u < n_tiles; part += in_t ? (fce[j].z + 1u < tile ? fce[j].x + fce[j].y : (fce[j].z < tile ? fce[j].x : 0u)) : 0u; }
        for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) s_p[wave] = part;
        __syncthreads();
        excl = s_p[0] + s_p[1] + s_p[2] + s_p[3];
    }
    if (F & F_COMPACT) excl = (F & F_TABLE) ? (excl & 0u) + tile * 1008u : tile * 1008u;  // 16 waves x 63 survivors per tile
    else excl = (excl & 0u) + base;
    uint32_t run = excl, fa = 0;
    float4 a1 = a, b1 = b, c1 = c, d1 = d;  // F_DEEP: round r+1, already in flight
    if (F & F_DEEP) { const uint32_t i1 = min(base + 256u + tid, last); a1 = p0[i1]; b1 = p1[i1]; c1 = p2[i1]; d1 = p3[i1]; }
    if (stamps) tsP = __builtin_amdgcn_s_memrealtime() + (excl & 0u);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const uint32_t i = base + r * 256 + tid;
        float4 an = a, bn = b, cn = c, dn = d;
        if (F & F_DEEP) { const uint32_t in2 = min(i + 512u, last); an = p0[in2]; bn = p1[in2]; cn = p2[in2]; dn = p3[in2]; }
        else if (F & F_CLAMP) { const uint32_t in2 = min(i + 256u, last); an = p0[in2]; bn = p1[in2]; cn = p2[in2]; dn = p3[in2]; }
        else if (r + 1 < 4 && i + 256 < n) { an = p0[i + 256]; bn = p1[i + 256]; cn = p2[i + 256]; dn = p3[i + 256]; }
        bool alive = i < lim;
        if (F & F_COMPACT) alive = alive && lane != 63u;
        uint32_t o = i;
        if (F & F_ROUNDB) {
            const unsigned long long m = __ballot(alive && a.w + 0.016f < d.w + 1e9f);
            if (lane == 0) s_c[r & 1][wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t wbase = run;
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint32_t cc = s_c[r & 1][w]; if ((uint32_t)w < wave) wbase += cc; run += cc; }
            o = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        } else if (F & F_COMPACT) {
            o = excl + (r * 4 + wave) * 63u + lane;
        }
        if (alive) {
            float4 e = make_float4(a.x + b.x, a.y * c.y, d.z, a.w);
            float4 f = make_float4(b.w, c.x, d.y, e.x);
            if (F & F_ALU) {  // 4 chains x 88 FMAs
                float x0 = a.x, x1 = b.y, x2 = c.z, x3 = d.w;
#pragma unroll
                for (int k = 0; k < 88; k++) { x0 = fmaf(x0, 1.0001f, x1); x1 = fmaf(x1, 0.9999f, x2); x2 = fmaf(x2, 1.0002f, x3); x3 = fmaf(x3, 0.9998f, x0); }
                e.x += x0 * 1e-30f; e.y += x1 * 1e-30f; f.x += x2 * 1e-30f; f.y += x3 * 1e-30f;
            }
            ((float4*)(out))[o] = a;
            ((float4*)(out + (size_t)16 * C))[o] = b;
            ((float4*)(out + (size_t)32 * C))[o] = c;
            ((float4*)(out + (size_t)48 * C))[o] = d;
            ((float4*)(out + (size_t)64 * C))[o] = e;
            ((float4*)(out + (size_t)80 * C))[o] = f;
            ((float*)(out + (size_t)96 * C))[o] = e.y;
            fa += (uint32_t)(e.y != 0.f);
        }
        if (F & F_DEEP) { a = a1; b = b1; c = c1; d = d1; a1 = an; b1 = bn; c1 = cn; d1 = dn; }
        else { a = an; b = bn; c = cn; d = dn; }
        if (stamps && r == 0) tsR0 = __builtin_amdgcn_s_memrealtime() + (fa & 0u);
    }
    if (F & F_TABLE) {
        for (int o2 = 32; o2; o2 >>= 1) fa += __shfl_xor(fa, o2);
        if (lane == 0) s_p[wave] = fa;
        __syncthreads();
        if (tid == 0) table_out[tile] = make_uint4(1008u + ((s_p[0] + s_p[1] + s_p[2] + s_p[3]) & 0u), 0u, (tile * 1008u) / 1024u, epoch);
    }
    if (tid == 0) {
        const unsigned long long tsE = __builtin_amdgcn_s_memrealtime() + (fa == 0xffffffffu ? 1 : 0);
        if (blockIdx.x == 0) ring[(epoch & 255u) * 2u] = ts0;
        if (blockIdx.x == gridDim.x - 1) ring[(epoch & 255u) * 2u + 1u] = tsE;
        if (stamps) {
            unsigned long long* d = stamps + (size_t)tile * 4;
            const unsigned xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20) & 15u;  // HW_REG_XCC_ID
            d[0] = ts0, d[1] = tsP, d[2] = (tsR0 << 4) | xcc, d[3] = tsE;
        }
    }
}

template <int F>
void run_twin(const char* label, hipStream_t s, char* p0, char* p1, uint32_t n, uint32_t C, unsigned long long* ring, int launches) {
    std::vector<unsigned long long> h(512, 0);
    CK(hipMemset(ring, 0, 512 * 8));
    const uint32_t tiles = (n + 1023) / 1024;
    TwinSeg hs[2] = {{p0, p1, C, n}, {p1, p0, C, n}};
    TwinSeg* segs; CK(hipMalloc(&segs, sizeof hs)); CK(hipMemcpy(segs, hs, sizeof hs, hipMemcpyHostToDevice));
    uint4 *t0, *t1; CK(hipMalloc(&t0, 4096 * 16)); CK(hipMalloc(&t1, 4096 * 16));
    CK(hipMemset(t0, 0, 4096 * 16)); CK(hipMemset(t1, 0, 4096 * 16));
    CK(hipStreamSynchronize(s));
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)tiles * 32));
    for (int i = 0; i < launches; i++)
        hipLaunchKernelGGL((k_twin<F>), dim3(tiles), dim3(256), 0, s, segs, (i & 1) ? t1 : t0, (i & 1) ? t0 : t1,
                           (i & 1) ? p1 : p0, (i & 1) ? p0 : p1, n, C, ring, (uint32_t)i, stamps);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> st((size_t)tiles * 4);
    CK(hipMemcpy(st.data(), stamps, (size_t)tiles * 32, hipMemcpyDeviceToHost));
    CK(hipFree(stamps));
    double pro = 0, r0 = 0, rest = 0, endmax = 0; unsigned long long tmin = ~0ull;
    for (uint32_t t = 0; t < tiles; t++) tmin = std::min(tmin, st[t * 4]);
    double xend[16] = {}; int xcnt[16] = {};
    for (uint32_t t = 0; t < tiles; t++) {
        const unsigned xcc = (unsigned)(st[t * 4 + 2] & 15u); st[t * 4 + 2] >>= 4;
        pro += (double)(st[t * 4 + 1] - st[t * 4]) / 100.0; r0 += (double)(st[t * 4 + 2] - st[t * 4 + 1]) / 100.0;
        rest += (double)(st[t * 4 + 3] - st[t * 4 + 2]) / 100.0; endmax = std::max(endmax, (double)(st[t * 4 + 3] - tmin) / 100.0);
        xend[xcc] += (double)(st[t * 4 + 3] - tmin) / 100.0; xcnt[xcc]++;
    }
    CK(hipMemcpy(h.data(), ring, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<double> per;
    for (int k = launches - 200; k < launches - 1; k++) {
        const int i = k & 255, j = (k + 1) & 255;
        per.push_back((double)((long long)h[2 * j] - (long long)h[2 * i]) / 100.0);
    }
    std::sort(per.begin(), per.end());
    printf("%-48s period p50 %6.2f p90 %6.2f | span %5.2f  prologue %5.2f round0 %5.2f rounds1-3 %5.2f\n", label, per[per.size() / 2],
           per[per.size() * 9 / 10], endmax, pro / tiles, r0 / tiles, rest / tiles);
    printf("    mean end per XCD:");
    for (int x = 0; x < 8; x++) printf(" %5.2f(%d)", xcnt[x] ? xend[x] / xcnt[x] : 0.0, xcnt[x]);
    printf("\n");
    CK(hipFree(segs)); CK(hipFree(t0)); CK(hipFree(t1));
}

int main(int argc, char** argv) {
    const bool only_twin = argc > 1 && std::string(argv[1]) == "twin";  // for counter collection: just the all-features twin
    const uint32_t n = 1000000, C = (n + 1023) / 1024 * 1024 + 262144;
    size_t pb = (size_t)100 * C;
    char *p0, *p1; CK(hipMalloc(&p0, pb)); CK(hipMalloc(&p1, pb));
    CK(hipMemset(p0, 0, pb)); CK(hipMemset(p1, 0, pb));
    unsigned long long* ring; CK(hipMalloc(&ring, 512 * 8));
    hipStream_t nb, blk; CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking)); CK(hipStreamCreate(&blk));
    for (int rep = 0; rep < 2 && !only_twin; rep++) {
        run<1>("null stream, 44 B kernarg", 0, p0, p1, n, C, ring, 600);
        run<1>("non-blocking stream, 44 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<1>("blocking stream, 44 B kernarg", blk, p0, p1, n, C, ring, 600);
        run<64>("non-blocking stream, 296 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<256>("non-blocking stream, 1064 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<512>("non-blocking stream, 2088 B kernarg", nb, p0, p1, n, C, ring, 600);
        run<960>("non-blocking stream, 3880 B kernarg", nb, p0, p1, n, C, ring, 600);
    }
    printf("feature twins (983040 particles, 960 tiles):\n");
    const uint32_t n2 = 983040;
    if (argc > 1 && std::string(argv[1]) == "alu") {
        for (int rep = 0; rep < 2; rep++) {
            run_twin<F_CLAMP | F_COMPACT | F_ROUNDB | F_TABLE | F_SCALAR>("all features", nb, p0, p1, n2, C, ring, 600);
            run_twin<F_CLAMP | F_COMPACT | F_ROUNDB | F_TABLE | F_SCALAR | F_ALU>("all features + 352 FMA/particle", nb, p0, p1, n2, C, ring, 600);
            run_twin<F_CLAMP | F_COMPACT | F_ROUNDB | F_TABLE | F_SCALAR | F_DEEP>("all features, prefetch 2 rounds ahead", nb, p0, p1, n2, C, ring, 600);
            run_twin<F_CLAMP | F_COMPACT | F_ROUNDB | F_TABLE | F_SCALAR | F_ALU | F_DEEP>("all + FMA + prefetch 2 ahead", nb, p0, p1, n2, C, ring, 600);
            run_twin<F_CLAMP | F_ALU>("clamped + FMA only", nb, p0, p1, n2, C, ring, 600);
            run_twin<F_CLAMP | F_ALU | F_DEEP>("clamped + FMA + prefetch 2 ahead", nb, p0, p1, n2, C, ring, 600);
        }
        return 0;
    }
    if (only_twin) {
        run_twin<F_CLAMP | F_COMPACT | F_ROUNDB | F_TABLE | F_SCALAR>("all features", nb, p0, p1, n2, C, ring, 300);
        return 0;
    }
    for (int rep = 0; rep < 2; rep++) {
        run_twin<0>("plain (predicated loads)", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP>("clamped unconditional loads", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_COMPACT>("+ compacted (misaligned) stores", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_ROUNDB>("+ per-round ballot/barrier", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_COMPACT | F_ROUNDB>("+ compacted + per-round barrier", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_TABLE>("+ forecast-table prologue", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_SCALAR>("+ dependent scalar chain", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_TABLE | F_SCALAR>("+ table + scalar chain", nb, p0, p1, n2, C, ring, 600);
        run_twin<F_CLAMP | F_COMPACT | F_ROUNDB | F_TABLE | F_SCALAR>("all features", nb, p0, p1, n2, C, ring, 600);
    }
    return 0;
}
