// fw_k_aux.hip -- readback / upload (AoS <-> planes), plane fills, render hand-off (fw_k_pack), AABB queries, live totals, the copy-bandwidth probe
// (gfx950 only; device helpers in fw_dev.h, launch interface in fw_kernels.h)
#include "fw_dev.h"

// ---------------------------------------------------------------------------------
// readback / upload / render hand-off helpers
// ---------------------------------------------------------------------------------

// SoA planes -> fw_particle records (26 x 4 B)
// (rot: the rotation of every particle of a type that cannot turn -- FW_TYPE_NOSPIN, its plane is not maintained -- or null)
__global__ void fw_k_gather(const char *buf, uint32_t C, uint32_t head, uint32_t n, int32_t pbr, float *out, bool nospin, float4 rot,
                            uint32_t life_plane, float life_const, const FwType *derived, const float *keys, bool cpl) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const uint32_t i = fw_ring_slot(head, li, C);
    const float4 q0 = fw_ld4(buf + FW_OFF_Q0(C), i), q1 = fw_ldq(buf + FW_OFF_Q1(C), C, i, cpl),
                 q2 = nospin ? rot : fw_ld4(buf + FW_OFF_Q2(C), i),
                 // (cannot turn: angular velocity 0; the lifetime from its plane, or -- a ring -- the type's one value)
                 q3 = !nospin ? fw_ldq(buf + FW_OFF_Q3(C), C, i, cpl)
                              : make_float4(0.0f, 0.0f, 0.0f, life_plane != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(C, life_plane), i) : life_const),
                 bc0 = fw_ld4(buf + FW_OFF_Q5(C), i), em0 = fw_ld4(buf + FW_OFF_Q6(C), i);
    float4 bc = bc0, em = em0;
    float sc = reinterpret_cast<const float *>(buf + FW_OFF_S4(C))[i];
    if (derived) fw_derived_values(*derived, keys + derived->keys_off, q0.w, q3.w, q1.w, &bc, &em, &sc);  // FW_TYPE_DERIVED
    float *r = out + (size_t)li * 26;
    r[0] = q0.x, r[1] = q0.y, r[2] = q0.z;
    r[3] = q1.x, r[4] = q1.y, r[5] = q1.z;
    r[6] = q2.x, r[7] = q2.y, r[8] = q2.z, r[9] = q2.w;
    r[10] = q3.x, r[11] = q3.y, r[12] = q3.z;
    r[13] = q1.w, r[14] = sc, r[15] = q0.w, r[16] = q3.w;
    r[17] = bc.x, r[18] = bc.y, r[19] = bc.z, r[20] = bc.w;
    r[21] = em.x, r[22] = em.y, r[23] = em.z, r[24] = em.w;
    reinterpret_cast<int32_t *>(r)[25] = pbr;
}

__global__ void fw_k_scatter(char *buf, uint32_t C, uint32_t n, uint32_t n_lplanes, const float *in) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = in + (size_t)i * 26;
    fw_st4(buf + FW_OFF_Q0(C), i, make_float4(r[0], r[1], r[2], r[15]));
    fw_st4(buf + FW_OFF_Q1(C), i, make_float4(r[3], r[4], r[5], r[13]));  // (the caller's particles: always a segment of the compacting path -- float4 planes)
    fw_st4(buf + FW_OFF_Q2(C), i, make_float4(r[6], r[7], r[8], r[9]));
    fw_st4(buf + FW_OFF_Q3(C), i, make_float4(r[10], r[11], r[12], r[16]));
    fw_st4(buf + FW_OFF_Q5(C), i, make_float4(r[17], r[18], r[19], r[20]));
    fw_st4(buf + FW_OFF_Q6(C), i, make_float4(r[21], r[22], r[23], r[24]));
    reinterpret_cast<float *>(buf + FW_OFF_S4(C))[i] = r[14];
    for (uint32_t k = 0; k < n_lplanes; k++) reinterpret_cast<float *>(buf + FW_OFF_L(C, k))[i] = FW_F32_MIN;
}

// both colour planes of a fresh buffer pair filled with the type's colours at age 0: for a constant gradient that is
// the colour of every particle for ever (FwOutWin::wr5 / wr6), for any other it is simply overwritten
__global__ void fw_k_fill_colors(char *buf0, char *buf1, uint32_t C, float4 bc, float4 em) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    fw_st4(buf0 + FW_OFF_Q5(C), i, bc), fw_st4(buf0 + FW_OFF_Q6(C), i, em);
    if (buf1) fw_st4(buf1 + FW_OFF_Q5(C), i, bc), fw_st4(buf1 + FW_OFF_Q6(C), i, em);
}

// a type leaves FW_TYPE_DERIVED (its instance buffer is detached): scale and colour planes of every slot, evaluated from
// the slot's age / lifetime / initial_scale -- what the updates would have stored
__global__ void fw_k_rederive(char *buf, uint32_t C, const FwType *T, const float *keys, bool nospin, uint32_t life_plane, float life_const, bool cpl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const float4 q0 = fw_ld4(buf + FW_OFF_Q0(C), i);
    const float life = !nospin ? fw_ldq_w(buf + FW_OFF_Q3(C), C, i, cpl)
                               : (life_plane != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(C, life_plane), i) : life_const);
    float4 bc, em;
    float sc;
    fw_derived_values(*T, keys + T->keys_off, q0.w, life, fw_ldq_w(buf + FW_OFF_Q1(C), C, i, cpl), &bc, &em, &sc);
    fw_st4(buf + FW_OFF_Q5(C), i, bc), fw_st4(buf + FW_OFF_Q6(C), i, em), fw_st1(buf + FW_OFF_S4(C), i, sc);
}
hipError_t fw_launch_rederive(hipStream_t s, char *buf, uint32_t capacity, const FwType *d_type, const float *d_keys, bool nospin,
                              uint32_t life_plane, float life_const, bool cpl) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_rederive, dim3((capacity + 255) / 256), dim3(256), 0, s, buf, capacity, d_type, d_keys, nospin, life_plane, life_const, cpl);
    return hipGetLastError();
}

// a type leaves FW_TYPE_NOSPIN: its rotation plane, which nobody maintained, gets the constant rotation in every slot
__global__ void fw_k_fill_rotation(char *buf0, char *buf1, uint32_t C, float4 rot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    fw_st4(buf0 + FW_OFF_Q2(C), i, rot);
    if (buf1) fw_st4(buf1 + FW_OFF_Q2(C), i, rot);
}

// a 4-byte plane filled with one value (the lifetime plane of a ring that becomes a compacting segment)
__global__ void fw_k_fill_plane1(char *buf0, char *buf1, size_t plane_off, uint32_t C, float v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    fw_st1(buf0 + plane_off, i, v);
    if (buf1) fw_st1(buf1 + plane_off, i, v);
}
// a type leaves FW_TYPE_NOSPIN: Q3 = {0, 0, 0, lifetime} again, the lifetime from its plane (or one value: a ring)
__global__ void fw_k_restore_q3(char *buf0, char *buf1, uint32_t C, uint32_t life_plane, float life_const, bool cpl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const bool pl = life_plane != 0xFFFFFFFFu;
    fw_stq(buf0 + FW_OFF_Q3(C), C, i, make_float4(0.f, 0.f, 0.f, pl ? fw_ld1(buf0 + FW_OFF_L(C, life_plane), i) : life_const), cpl);
    if (buf1) fw_stq(buf1 + FW_OFF_Q3(C), C, i, make_float4(0.f, 0.f, 0.f, pl ? fw_ld1(buf1 + FW_OFF_L(C, life_plane), i) : life_const), cpl);
}

// ParticleInstance packing (reference src/render.rs:95-115): {pos, scale, rot, base, emissive}
// SoA planes -> ParticleInstance records (render.rs:95-115).  Loads are plane-wise coalesced; the 64-byte records are
// transposed through LDS so that every store instruction of a wave writes 1 KiB of consecutive bytes (a lane writing
// its own record with four float4 stores would touch 64 lines a quarter at a time).
__global__ __launch_bounds__(256) void fw_k_pack(const char *buf, uint32_t C, uint32_t head, const uint32_t *d_count,
                                                 uint32_t n_upper, float4 *out, bool nospin, float4 rot, const uint32_t *d_rold,
                                                 const FwType *derived, const float *keys, uint32_t life_plane, float life_const, bool cpl) {
    __shared__ float4 s_rec[256 * 4];
    // a range ring (d_rold: the size of its old part, FwGlobals::rold): `head` is the slot of the first young particle
    if (d_rold) head = fw_range_head(head, *d_rold, C);
    const uint32_t n = min(*d_count, n_upper);
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = blockIdx.x * 256u; b < n; b += gridDim.x * 256u) {
        const uint32_t i = fw_ring_slot(head, min(b + tid, n - 1u), C);
        const float4 q0 = fw_ld4(buf + FW_OFF_Q0(C), i);
        float sc = fw_ld1(buf + FW_OFF_S4(C), i);
        const float4 q2 = nospin ? rot : fw_ld4(buf + FW_OFF_Q2(C), i);
        float4 q5 = fw_ld4(buf + FW_OFF_Q5(C), i);
        float4 q6 = fw_ld4(buf + FW_OFF_Q6(C), i);
        if (derived) {  // FW_TYPE_DERIVED: the three planes are not maintained -- what the last update computed, again
            const float life = !nospin ? fw_ldq_w(buf + FW_OFF_Q3(C), C, i, cpl)
                                       : (life_plane != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(C, life_plane), i) : life_const);
            fw_derived_values(*derived, keys + derived->keys_off, q0.w, life, fw_ldq_w(buf + FW_OFF_Q1(C), C, i, cpl), &q5, &q6, &sc);
        }
        s_rec[tid * 4 + 0] = make_float4(q0.x, q0.y, q0.z, sc);
        s_rec[tid * 4 + 1] = q2;
        s_rec[tid * 4 + 2] = q5;
        s_rec[tid * 4 + 3] = q6;
        __syncthreads();
        const uint32_t cnt4 = min(256u, n - b) * 4u;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t e = k * 256u + tid;
            if (e < cnt4) out[(size_t)b * 4 + e] = s_rec[e];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void fw_atomic_minf(float *addr, float v) {
    int *ia = reinterpret_cast<int *>(addr);
    int old = __hip_atomic_load(ia, RLX, AGENT);
    while (v < __int_as_float(old)) {
        const int assumed = old;
        old = atomicCAS(ia, assumed, __float_as_int(v));
        if (old == assumed) break;
    }
}
__device__ __forceinline__ void fw_atomic_maxf(float *addr, float v) {
    int *ia = reinterpret_cast<int *>(addr);
    int old = __hip_atomic_load(ia, RLX, AGENT);
    while (v > __int_as_float(old)) {
        const int assumed = old;
        old = atomicCAS(ia, assumed, __float_as_int(v));
        if (old == assumed) break;
    }
}

// update_aabbs reduction (reference src/render.rs:677-703): min/max over position -/+ scale.
// blockIdx.y = segment of the spawner; out6 = {min xyz, max xyz}, pre-set to {+MAX, -MAX}.
// update_aabbs (render.rs:677-703) in two launches and no atomics: workgroups reduce position -/+ scale over their slice
// of every particle type of the spawner into one partial box each; a single workgroup folds the partials and leaves
// {min.xyz, any, max.xyz, -} in PINNED host memory, so the query costs one stream synchronisation and no copies.
struct FwSegList {
    uint32_t n;
    uint32_t id[8];    // FW_MAX_TYPES
    uint32_t head[8];  // slot of each segment's particle 0 (FIFO rings; 0 otherwise)
    uint32_t range_y[8];  // 0xFFFFFFFF, or -- a range ring -- its young count: head[] is the slot of its first young particle
    uint32_t life_plane[8];  // FW_TYPE_DERIVED types (scale evaluated from age / lifetime): where a type that cannot turn keeps
    float life_const[8];     // its lifetimes -- a plane behind the last_emitted_age planes, or (0xFFFFFFFF) one value
};
#define FW_AABB_BLOCKS 256u
__global__ __launch_bounds__(FW_BLOCK) void fw_k_aabb(FwGlobals g, FwSegList L, uint32_t parity, float *part8) {
    __shared__ float s_m[4][6];
    float mn[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
    float mx[3] = {FW_F32_MIN, FW_F32_MIN, FW_F32_MIN};
    for (uint32_t k = 0; k < L.n; k++) {
        const uint32_t seg = L.id[k];
        const FwSeg &S = g.segs[seg];
        const uint32_t n = g.count[parity * g.max_seg + seg];
        const char *buf = S.buf[parity];
        const FwType &TT = g.types[S.type_idx];
        uint32_t head = L.head[k];
        if (L.range_y[k] != 0xFFFFFFFFu) head = fw_range_head(head, g.rold[parity * g.max_seg + seg], S.capacity);  // (a range ring)
        for (uint32_t li = blockIdx.x * FW_BLOCK + threadIdx.x; li < n; li += gridDim.x * FW_BLOCK) {
            const uint32_t i = fw_ring_slot(head, li, S.capacity);
            const float4 q0 = fw_ld4(buf + FW_OFF_Q0(S.capacity), i);
            float sc = fw_ld1(buf + FW_OFF_S4(S.capacity), i);
            if (TT.flags & FW_TYPE_DERIVED) {  // the scale plane is not maintained: what the last update computed, again
                const float life = !(TT.flags & FW_TYPE_NOSPIN) ? fw_ldq_w(buf + FW_OFF_Q3(S.capacity), S.capacity, i, S.cpl != 0u)
                                   : (L.life_plane[k] != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(S.capacity, L.life_plane[k]), i) : L.life_const[k]);
                const float *keys = g.keys + TT.keys_off;
                sc = fw_ldq_w(buf + FW_OFF_Q1(S.capacity), S.capacity, i, S.cpl != 0u) * fw_curve_sample(TT.sc_kind, TT.sc_n, keys, keys + TT.o_sc_v, q0.w / life);
            }
            mn[0] = fminf(mn[0], q0.x - sc), mn[1] = fminf(mn[1], q0.y - sc), mn[2] = fminf(mn[2], q0.z - sc);
            mx[0] = fmaxf(mx[0], q0.x + sc), mx[1] = fmaxf(mx[1], q0.y + sc), mx[2] = fmaxf(mx[2], q0.z + sc);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int c = 0; c < 3; c++) s_m[wave][c] = mn[c], s_m[wave][3 + c] = mx[c];
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t c = threadIdx.x;
        float v = s_m[0][c];
        for (int w = 1; w < 4; w++) v = c < 3 ? fminf(v, s_m[w][c]) : fmaxf(v, s_m[w][c]);
        part8[blockIdx.x * 8u + c] = v;
    }
}
__global__ __launch_bounds__(FW_AABB_BLOCKS) void fw_k_aabb_fold(FwGlobals g, FwSegList L, uint32_t parity,
                                                                 const float *part8, float *host8) {
    __shared__ float s_m[FW_AABB_BLOCKS / 64][6];
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        v[c] = part8[threadIdx.x * 8u + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(v[c], o, 64);
            v[c] = c < 3 ? fminf(v[c], other) : fmaxf(v[c], other);
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int c = 0; c < 6; c++) s_m[wave][c] = v[c];
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t c = threadIdx.x;
        float r = s_m[0][c];
        for (uint32_t w = 1; w < FW_AABB_BLOCKS / 64; w++) r = c < 3 ? fminf(r, s_m[w][c]) : fmaxf(r, s_m[w][c]);
        host8[c < 3 ? c : c + 1u] = r;
    }
    if (threadIdx.x == 0) {
        uint32_t any = 0;
        for (uint32_t k = 0; k < L.n; k++) any |= g.count[parity * g.max_seg + L.id[k]];
        host8[3] = any ? 1.0f : 0.0f;
    }
}

// fw_spawner_aabb from the per-tile boxes the last update left (fw_tile_box_flush): one workgroup folds the boxes of the
// spawner's segments -- a few hundred 32-byte records -- and leaves {min.xyz, any, max.xyz, -} in pinned host memory.
__global__ __launch_bounds__(FW_BLOCK) void fw_k_aabb_from_tiles(FwGlobals g, FwSegList L, uint32_t parity, uint32_t epoch,
                                                                 const uint32_t *seg_tile_first, float *host8) {
    __shared__ float s_m[FW_BLOCK / 64][6];
    float v[6] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f, FW_F32_MIN, FW_F32_MIN, FW_F32_MIN};
    const float4 *boxes = reinterpret_cast<const float4 *>(g.tile_box);
    for (uint32_t k = 0; k < L.n; k++) {
        const uint32_t seg = L.id[k];
        const uint32_t t0 = seg_tile_first[seg], t1 = seg_tile_first[seg + 1];
        for (uint32_t t = t0 + threadIdx.x; t < t1; t += FW_BLOCK) {
            const float4 lo = boxes[(size_t)t * 2], hi = boxes[(size_t)t * 2 + 1];
            if (__float_as_uint(lo.w) != epoch) continue;  // a tile that held no particles in the last update
            v[0] = fminf(v[0], lo.x), v[1] = fminf(v[1], lo.y), v[2] = fminf(v[2], lo.z);
            v[3] = fmaxf(v[3], hi.x), v[4] = fmaxf(v[4], hi.y), v[5] = fmaxf(v[5], hi.z);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(v[c], o, 64);
            v[c] = c < 3 ? fminf(v[c], other) : fmaxf(v[c], other);
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int c = 0; c < 6; c++) s_m[wave][c] = v[c];
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t c = threadIdx.x;
        float r = s_m[0][c];
        for (uint32_t w = 1; w < FW_BLOCK / 64; w++) r = c < 3 ? fminf(r, s_m[w][c]) : fmaxf(r, s_m[w][c]);
        host8[c < 3 ? c : c + 1u] = r;
    }
    if (threadIdx.x == 0) {
        uint32_t any = 0;
        for (uint32_t k = 0; k < L.n; k++) any |= g.count[parity * g.max_seg + L.id[k]];
        host8[3] = any ? 1.0f : 0.0f;
    }
}

__global__ void fw_k_total(const uint32_t *counts, uint32_t n_seg, unsigned long long *out) {
    __shared__ unsigned long long s[4];
    unsigned long long t = 0;
    for (uint32_t i = threadIdx.x; i < n_seg; i += blockDim.x) t += counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) *out = s[0] + s[1] + s[2] + s[3];
}

// float4 streaming copy: the measured-roofline probe (bytes read + written per second).  Shape chosen by a sweep on
// MI355X at 1 GiB -> 1 GiB (tools/membw <MiB> copy, profiles/r02/copy_sweep.txt): grid-strided, four float4 per lane in
// flight, non-temporal loads and stores, 16384 workgroups: 6.34 TB/s (plain one-float4 grid-stride: 4.9; hipMemcpy D2D: 5.0).
__global__ __launch_bounds__(FW_BLOCK) void fw_k_copy(const float4 *src, float4 *dst, size_t n4) {
    constexpr int U = 4;
    const size_t stride = (size_t)gridDim.x * FW_BLOCK * U;
    const FW_GLOBAL fw_f4 *s = reinterpret_cast<const FW_GLOBAL fw_f4 *>(reinterpret_cast<uintptr_t>(src));
    FW_GLOBAL fw_f4 *d = reinterpret_cast<FW_GLOBAL fw_f4 *>(reinterpret_cast<uintptr_t>(dst));
    for (size_t i = (size_t)blockIdx.x * FW_BLOCK * U + threadIdx.x; i < n4; i += stride) {
        fw_f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + (size_t)u * FW_BLOCK;
            if (j < n4) v[u] = __builtin_nontemporal_load(&s[j]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + (size_t)u * FW_BLOCK;
            if (j < n4) __builtin_nontemporal_store(v[u], &d[j]);
        }
    }
}

// ---- launch wrappers

hipError_t fw_launch_gather(hipStream_t s, const char *buf, uint32_t capacity, uint32_t head, uint32_t n, int32_t pbr, void *d_out,
                            const float *const_rot, uint32_t life_plane, float life_const, const FwType *derived, const float *keys, bool cpl) {
    if (!n) return hipSuccess;
    const float4 rot = const_rot ? make_float4(const_rot[0], const_rot[1], const_rot[2], const_rot[3]) : make_float4(0.f, 0.f, 0.f, 1.f);
    hipLaunchKernelGGL(fw_k_gather, dim3((n + 255) / 256), dim3(256), 0, s, buf, capacity, head, n, pbr, (float *)d_out,
                       const_rot != nullptr, rot, life_plane, life_const, derived, keys, cpl);
    return hipGetLastError();
}

hipError_t fw_launch_scatter(hipStream_t s, char *buf, uint32_t capacity, uint32_t n, uint32_t n_lplanes,
                             const void *d_in) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fw_k_scatter, dim3((n + 255) / 256), dim3(256), 0, s, buf, capacity, n, n_lplanes,
                       (const float *)d_in);
    return hipGetLastError();
}

hipError_t fw_launch_fill_colors(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, const float bc[4], const float em[4]) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_fill_colors, dim3((capacity + 255) / 256), dim3(256), 0, s, buf0, buf1, capacity,
                       make_float4(bc[0], bc[1], bc[2], bc[3]), make_float4(em[0], em[1], em[2], em[3]));
    return hipGetLastError();
}

hipError_t fw_launch_fill_plane1(hipStream_t s, char *buf0, char *buf1, size_t plane_off, uint32_t capacity, float v) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_fill_plane1, dim3((capacity + 255) / 256), dim3(256), 0, s, buf0, buf1, plane_off, capacity, v);
    return hipGetLastError();
}
hipError_t fw_launch_restore_q3(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, uint32_t life_plane, float life_const, bool cpl) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_restore_q3, dim3((capacity + 255) / 256), dim3(256), 0, s, buf0, buf1, capacity, life_plane, life_const, cpl);
    return hipGetLastError();
}

hipError_t fw_launch_fill_rotation(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, const float rot[4]) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_fill_rotation, dim3((capacity + 255) / 256), dim3(256), 0, s, buf0, buf1, capacity,
                       make_float4(rot[0], rot[1], rot[2], rot[3]));
    return hipGetLastError();
}

hipError_t fw_launch_pack_instances(hipStream_t s, const char *buf, uint32_t capacity, uint32_t head, const uint32_t *d_count,
                                    uint32_t n_upper, void *d_out, const float *const_rot, const uint32_t *d_rold, const FwType *derived,
                                    const float *keys, uint32_t life_plane, float life_const, bool cpl) {
    if (!n_upper) return hipSuccess;
    uint32_t blocks = (n_upper + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const float4 rot = const_rot ? make_float4(const_rot[0], const_rot[1], const_rot[2], const_rot[3]) : make_float4(0.f, 0.f, 0.f, 1.f);
    hipLaunchKernelGGL(fw_k_pack, dim3(blocks), dim3(256), 0, s, buf, capacity, head, d_count, n_upper, (float4 *)d_out,
                       const_rot != nullptr, rot, d_rold, derived, keys, life_plane, life_const, cpl);
    return hipGetLastError();
}

hipError_t fw_launch_aabb(hipStream_t s, const FwGlobals &g, const uint32_t *seg_ids, const uint32_t *seg_heads, uint32_t n_segs,
                          uint32_t parity, float *d_part, float *h_out8, const uint32_t *seg_range_y, const uint32_t *seg_life_plane,
                          const float *seg_life_const) {
    if (!n_segs || n_segs > 8u) return hipErrorInvalidValue;  // FW_MAX_TYPES
    FwSegList L{};
    L.n = n_segs;
    for (uint32_t i = 0; i < n_segs; i++)
        L.id[i] = seg_ids[i], L.head[i] = seg_heads ? seg_heads[i] : 0u, L.range_y[i] = seg_range_y ? seg_range_y[i] : 0xFFFFFFFFu,
        L.life_plane[i] = seg_life_plane ? seg_life_plane[i] : 0xFFFFFFFFu, L.life_const[i] = seg_life_const ? seg_life_const[i] : 0.0f;
    hipLaunchKernelGGL(fw_k_aabb, dim3(FW_AABB_BLOCKS), dim3(FW_BLOCK), 0, s, g, L, parity, d_part);
    hipLaunchKernelGGL(fw_k_aabb_fold, dim3(1), dim3(FW_AABB_BLOCKS), 0, s, g, L, parity, (const float *)d_part, h_out8);
    return hipGetLastError();
}

hipError_t fw_launch_aabb_from_tiles(hipStream_t s, const FwGlobals &g, const uint32_t *seg_ids, uint32_t n_segs,
                                     uint32_t parity, uint32_t epoch, const uint32_t *d_seg_tile_first, float *h_out8) {
    if (!n_segs || n_segs > 8u) return hipErrorInvalidValue;  // FW_MAX_TYPES
    FwSegList L{};
    L.n = n_segs;
    for (uint32_t i = 0; i < n_segs; i++) L.id[i] = seg_ids[i];
    hipLaunchKernelGGL(fw_k_aabb_from_tiles, dim3(1), dim3(FW_BLOCK), 0, s, g, L, parity, epoch, d_seg_tile_first, h_out8);
    return hipGetLastError();
}

hipError_t fw_launch_total(hipStream_t s, const uint32_t *counts, uint32_t n_seg, unsigned long long *d_out) {
    hipLaunchKernelGGL(fw_k_total, dim3(1), dim3(256), 0, s, counts, n_seg, d_out);
    return hipGetLastError();
}

hipError_t fw_launch_copy_probe(hipStream_t s, const void *src, void *dst, size_t bytes) {
    hipLaunchKernelGGL(fw_k_copy, dim3(16384), dim3(FW_BLOCK), 0, s, (const float4 *)src, (float4 *)dst, bytes / 16);
    return hipGetLastError();
}

