// fw_dev.h -- device-side helpers shared by the kernel files of the firework backend (gfx950): plane loads / stores,
// the spawn formula, the integrate-and-store step, destroyed records, decoupled look-back, survivor-forecast sums.
// Everything here is __device__ __forceinline__ (or a type / macro): every kernel family is a translation unit of its own
//   fw_k_general.hip  the compacting update (fw_k_update, fw_k_update_stream) + the count / scan / collision feature path
//   fw_k_rings.hip    the in-place ring updates (fw_k_update_fifo, fw_k_update_range)
//   fw_k_nested.hip   fw_k_spawn, fw_k_nest
//   fw_k_aux.hip      readback / upload / render hand-off / AABB / probes
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>

#include "fw_kernels.h"
#include "fw_math.h"

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

// ---------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------

// an internal error (a check of the host's bookkeeping against the particles failed, a look-back wait ran out): the device
// flags for whoever synchronises next, and the pinned word the next fw_step looks at (FwGlobals::err_host)
// sets bits of the device's error flags (FwGlobals::err) and notes in pinned memory that there is something to read there:
// the host half fetches the flags only then (a blocking 32-byte copy is ~20 us of every synchronising call otherwise)
__device__ __forceinline__ void fw_flag(const FwGlobals &g, uint32_t bits) {
    atomicOr(g.err, bits);
    if (g.err_host) g.err_host[1] = 1ull;
}

__device__ __forceinline__ void fw_raise(const FwGlobals &g, uint32_t check, uint32_t seg, uint32_t x) {
    fw_flag(g, FW_ERR_FORECAST);
    g.err[5] = check, g.err[6] = seg, g.err[7] = x;
    if (g.err_host) *g.err_host = (1ull << 63) | ((unsigned long long)check << 32) | seg;
}

__device__ __forceinline__ uint32_t fw_lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t fw_wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// largest i in [0, n) with first[i] <= x   (first[] ascending, first[0] == 0)
__device__ __forceinline__ uint32_t fw_upper_slot(const uint32_t *first, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (first[mid] <= x)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Particle buffers are reached through pointers that were themselves loaded from memory, so the compiler
// cannot prove they are global and would emit FLAT loads/stores (which also tick lgkmcnt and so serialise
// against every LDS / scalar-memory wait).  Casting to address space 1 gives global_load/store_dwordx4.
typedef float fw_f4 __attribute__((ext_vector_type(4)));
#define FW_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 fw_ld4(const char *plane, uint32_t i) {
    const fw_f4 v = reinterpret_cast<const FW_GLOBAL fw_f4 *>(reinterpret_cast<uintptr_t>(plane))[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 fw_ld4_nt(const char *plane, uint32_t i) {  // bypasses the CU's L1
    const fw_f4 v = __builtin_nontemporal_load(
        &reinterpret_cast<const FW_GLOBAL fw_f4 *>(reinterpret_cast<uintptr_t>(plane))[i]);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void fw_st4(char *plane, uint32_t i, float4 v) {
    const fw_f4 x = {v.x, v.y, v.z, v.w};
    reinterpret_cast<FW_GLOBAL fw_f4 *>(reinterpret_cast<uintptr_t>(plane))[i] = x;
}
__device__ __forceinline__ float fw_ld1(const char *plane, uint32_t i) {
    return reinterpret_cast<const FW_GLOBAL float *>(reinterpret_cast<uintptr_t>(plane))[i];
}
__device__ __forceinline__ void fw_st1(char *plane, uint32_t i, float v) {
    reinterpret_cast<FW_GLOBAL float *>(reinterpret_cast<uintptr_t>(plane))[i] = v;
}

// Window addressing: a workgroup reads one contiguous window of each input plane and writes one contiguous window of
// each output plane.  With the plane pointer advanced to the window start on the scalar unit and a 32-bit byte offset
// per lane, the access is "SGPR pair + VGPR offset" (the saddr form of global_load / global_store): no 64-bit vector
// address arithmetic and no address register pairs kept alive per plane.
// NT: non-temporal accesses (the `nt` bit of global_load / global_store).  The kernels that update rings in place exist in
// three forms, the host picks one per launch from what the launch streams (fw_ctx::nt_bytes / nt_wo_bytes):
//   0  plain: everything may stay in the 256 MiB Infinity Cache (configs[1]: 164 MB, and it does);
//   1  the planes no update ever reads back -- scale, base colour, emissive colour: 36 of a particle's 100-164 bytes --
//      are stored non-temporally, so what the cache keeps is what the next frame reads (configs[4]'s share, 425 MB: 92.8 ->
//      87.8 us; a 4M-particle ring, 645 MB: 115 -> 100 us; at 164 MB: nothing either way; profiles/r03/nt_wo.txt);
//   2  every plane access non-temporal: a launch several times the cache gains another 4-8 % (configs[2] 332 -> 317 us, one
//      16M ring 497 -> 464 us); one that fits would lose up to 25 % (configs[1] 24.0 -> 30.8 us; profiles/r03/nt_ab.txt,
//      nt_sweep.txt).
template <bool NT = false>
__device__ __forceinline__ float4 fw_ld4w(const char *win, uint32_t byte_off) {
    const FW_GLOBAL fw_f4 *p = reinterpret_cast<const FW_GLOBAL fw_f4 *>(
        reinterpret_cast<const FW_GLOBAL char *>(reinterpret_cast<uintptr_t>(win)) + byte_off);
    fw_f4 v;
    if constexpr (NT) v = __builtin_nontemporal_load(p);
    else v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// (a plane no particle type of the launch has -- rotation / angular velocity in an all-FW_TYPE_NOSPIN launch: not even a dummy load)
template <bool SKIP, bool NT>
__device__ __forceinline__ float4 fw_ld4w_opt(const char *win, uint32_t byte_off) {
    if constexpr (SKIP) return make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    else return fw_ld4w<NT>(win, byte_off);
}
template <bool NT = false>
__device__ __forceinline__ void fw_st4w(char *win, uint32_t byte_off, float4 v) {
    const fw_f4 x = {v.x, v.y, v.z, v.w};
    FW_GLOBAL fw_f4 *p = reinterpret_cast<FW_GLOBAL fw_f4 *>(reinterpret_cast<FW_GLOBAL char *>(reinterpret_cast<uintptr_t>(win)) + byte_off);
    if constexpr (NT) __builtin_nontemporal_store(x, p);
    else *p = x;
}
template <bool NT = false>
__device__ __forceinline__ float fw_ld1w(const char *win, uint32_t byte_off) {
    const FW_GLOBAL float *p = reinterpret_cast<const FW_GLOBAL float *>(reinterpret_cast<const FW_GLOBAL char *>(reinterpret_cast<uintptr_t>(win)) + byte_off);
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT = false>
__device__ __forceinline__ void fw_st1w(char *win, uint32_t byte_off, float v) {
    FW_GLOBAL float *p = reinterpret_cast<FW_GLOBAL float *>(reinterpret_cast<FW_GLOBAL char *>(reinterpret_cast<uintptr_t>(win)) + byte_off);
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ uint4 fw_ld4u(const char *win, uint32_t byte_off) {
    typedef uint32_t fw_u4v __attribute__((ext_vector_type(4)));
    const fw_u4v v = *reinterpret_cast<const FW_GLOBAL fw_u4v *>(
        reinterpret_cast<const FW_GLOBAL char *>(reinterpret_cast<uintptr_t>(win)) + byte_off);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t fw_ld1u(const uint32_t *base, uint32_t idx) {
    return reinterpret_cast<const FW_GLOBAL uint32_t *>(reinterpret_cast<uintptr_t>(base))[idx];
}
// ---- component planes (round 6, fw_device.h; RING segments only -- FwSeg::cpl): the Q1 and Q3 regions of a ring's buffer hold their
// four components as four 4-byte planes of C slots each -- x at +0, y at +4C, z at +8C, w at +12C bytes -- because `.w` of both (initial_scale, lifetime) never changes:
// an in-place update loads and stores three dwords per lane and plane instead of a dwordx4 (tools/inplace.hip: the shapes follow the
// bytes with scalar planes, 1M particles 16.6 -> 14.7 us, 16M 167.5 -> 147.5; packed float3 planes -- dwordx3 -- get SLOWER).
// Segments on the compacting paths keep float4 planes: out of place everything moves anyway, and four dword accesses where one dwordx4
// did cost the streaming kernel 5-9 % (profiles/r06/component_planes.txt); a ring that leaves for those paths is transposed where it
// is copied (realloc_segment).  Readers take the layout as a flag (fw_ldq).
// `reg`: the region's base (buf + FW_OFF_Q1 / Q3); `win`: the x plane advanced to a window's first slot; off4 = 4 * (slot - first).
#define FW_CP(C) ((size_t)4 * (C))  // bytes between two component planes of a region
__device__ __forceinline__ float4 fw_ldc4(const char *reg, uint32_t C, uint32_t i) {
    const size_t cp = FW_CP(C);
    return make_float4(fw_ld1(reg, i), fw_ld1(reg + cp, i), fw_ld1(reg + 2 * cp, i), fw_ld1(reg + 3 * cp, i));
}
__device__ __forceinline__ float4 fw_ldc3(const char *reg, uint32_t C, uint32_t i, float w) {
    const size_t cp = FW_CP(C);
    return make_float4(fw_ld1(reg, i), fw_ld1(reg + cp, i), fw_ld1(reg + 2 * cp, i), w);
}
__device__ __forceinline__ void fw_stc4(char *reg, uint32_t C, uint32_t i, float4 v) {
    const size_t cp = FW_CP(C);
    fw_st1(reg, i, v.x), fw_st1(reg + cp, i, v.y), fw_st1(reg + 2 * cp, i, v.z), fw_st1(reg + 3 * cp, i, v.w);
}
// element i of a Q1 / Q3 region in either layout (cpl: component planes) -- the readers
__device__ __forceinline__ float4 fw_ldq(const char *reg, uint32_t C, uint32_t i, bool cpl) {
    return cpl ? fw_ldc4(reg, C, i) : fw_ld4(reg, i);
}
__device__ __forceinline__ float fw_ldq_w(const char *reg, uint32_t C, uint32_t i, bool cpl) {
    return cpl ? fw_ld1(reg + 3 * FW_CP(C), i) : fw_ld4(reg, i).w;
}
__device__ __forceinline__ void fw_stq(char *reg, uint32_t C, uint32_t i, float4 v, bool cpl) {
    if (cpl) fw_stc4(reg, C, i, v);
    else fw_st4(reg, i, v);
}
template <bool NT = false>
__device__ __forceinline__ float4 fw_ldc4w(const char *win, size_t cp, uint32_t off4) {
    return make_float4(fw_ld1w<NT>(win, off4), fw_ld1w<NT>(win + cp, off4), fw_ld1w<NT>(win + 2 * cp, off4), fw_ld1w<NT>(win + 3 * cp, off4));
}
template <bool NT = false>
__device__ __forceinline__ float4 fw_ldc3w(const char *win, size_t cp, uint32_t off4, float w) {
    return make_float4(fw_ld1w<NT>(win, off4), fw_ld1w<NT>(win + cp, off4), fw_ld1w<NT>(win + 2 * cp, off4), w);
}
template <bool NT = false>
__device__ __forceinline__ void fw_stc3w(char *win, size_t cp, uint32_t off4, float x, float y, float z) {
    fw_st1w<NT>(win, off4, x), fw_st1w<NT>(win + cp, off4, y), fw_st1w<NT>(win + 2 * cp, off4, z);
}
template <bool NT = false>
__device__ __forceinline__ void fw_stc4w(char *win, size_t cp, uint32_t off4, float4 v) {
    fw_st1w<NT>(win, off4, v.x), fw_st1w<NT>(win + cp, off4, v.y), fw_st1w<NT>(win + 2 * cp, off4, v.z), fw_st1w<NT>(win + 3 * cp, off4, v.w);
}

// Bounds-checked window loads (buffer_load through a 128-bit resource descriptor): a lane whose offset falls outside
// [0, bytes) gets zeros and costs NO memory traffic -- a negative offset wraps to a huge one, so one descriptor clips a tile at
// both ends.  Used where a tile of a ring only partly holds the particles it is dispatched for (the ends of a range ring's
// young part: 7-8 tiles for the 6.4 tiles of data of a configs[4] emitter -- unconditional loads fetched every slot of them).
// The descriptor is built from workgroup-uniform values only.
typedef __amdgpu_buffer_rsrc_t fw_rsrc;
__device__ __forceinline__ fw_rsrc fw_make_rsrc(const char *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), (short)0, (int)bytes, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ float4 fw_ldb4(fw_rsrc r, uint32_t byte_off) {
    typedef uint32_t fw_u4b __attribute__((ext_vector_type(4)));
    const fw_u4b v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, NT ? 2 : 0);  // aux bit 1 = nt
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <bool SKIP, bool NT>
__device__ __forceinline__ float4 fw_ldb4_opt(fw_rsrc r, uint32_t byte_off) {
    if constexpr (SKIP) return make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    else return fw_ldb4<NT>(r, byte_off);
}
template <bool NT>
__device__ __forceinline__ float fw_ldb1(fw_rsrc r, uint32_t byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, NT ? 2 : 0));
}

struct FwOutWin {  // output planes advanced to slot `first` (workgroup-uniform)
    char *q0, *q1, *q2, *q3, *q5, *q6, *s4;  // (q1 / q3 of a ring -- component planes: the x plane advanced by 4 * first, `cp` bytes apart)
    size_t cp;
    uint32_t first;
    // A gradient with a single key (the reference's default emissive colour, core.rs:205) gives every particle of the
    // type the same colour for ever: both buffers of the segment are filled with it once (fw_k_fill_colors) and the
    // update does not write that plane again -- 16 of its 164 bytes per particle per constant gradient.  wr5 / wr6:
    // this launch writes base_color / emissive_color (always true for a few frames after the caller rewrote particles).
    bool wr5, wr6;
    bool wr2;  // rotation plane (false for a type that cannot turn: FW_TYPE_NOSPIN)
    // ... and such a type keeps its lifetimes in a 4-byte plane of their own behind the last_emitted_age planes instead of
    // Q3 (angular velocity 0 + lifetime): wr3 false -> the lifetime goes to `lf`, Q3 is not written
    bool wr3;
    char *lf;
    bool wr4;  // scale plane (false, like wr5 / wr6, for a type whose instance records carry it: FW_TYPE_DERIVED)
};
__device__ __forceinline__ FwOutWin fw_out_window(char *ob, uint32_t C, uint32_t first, const FwType &T, uint32_t force_colors,
                                                  uint32_t n_lplanes = 0u, bool cpl = false) {
    const size_t f16 = (size_t)first * 16u, f4 = (size_t)first * 4u, fq = cpl ? f4 : f16;
    return FwOutWin{ob + FW_OFF_Q0(C) + f16, ob + FW_OFF_Q1(C) + fq, ob + FW_OFF_Q2(C) + f16, ob + FW_OFF_Q3(C) + fq,
                    ob + FW_OFF_Q5(C) + f16, ob + FW_OFF_Q6(C) + f16, ob + FW_OFF_S4(C) + f4, FW_CP(C), first,
                    (T.bc_kind != 0 || force_colors != 0u) && !(T.flags & FW_TYPE_DERIVED),
                    (T.em_kind != 0 || force_colors != 0u) && !(T.flags & FW_TYPE_DERIVED), !(T.flags & FW_TYPE_NOSPIN),
                    !(T.flags & FW_TYPE_NOSPIN), ob + FW_OFF_L(C, n_lplanes) + (size_t)first * 4u, !(T.flags & FW_TYPE_DERIVED)};
}

// slot of logical particle i of a segment whose particle 0 sits in slot `head` (0 unless the segment is a FIFO ring)
__device__ __forceinline__ uint32_t fw_ring_slot(uint32_t head, uint32_t i, uint32_t C) {
    const uint32_t s = head + i;  // head < C, i < C <= 0xFFFF0000 / 2 ... no overflow: capacities stay below 2^31
    return s >= C ? s - C : s;
}

// slot of particle 0 of a RANGE ring whose first young particle sits in slot b: the old part (`rold` survivors, FwGlobals::rold)
// lies right before it
__device__ __forceinline__ uint32_t fw_range_head(uint32_t b, uint32_t rold, uint32_t C) {
    const uint32_t back = rold >= C ? 0u : rold;  // (rold < C always: a guard against a wild value, not a case)
    return b >= back ? b - back : b + C - back;
}

// Q3 (angular velocity, lifetime) of particle `idx`: from the plane, or -- a type that cannot turn -- zero and the lifetime
// plane (FwOutWin::lf)
__device__ __forceinline__ float4 fw_load_q3(const char *buf, uint32_t C, uint32_t n_lplanes, uint32_t idx, bool nospin, bool cpl = false) {
    if (nospin) return make_float4(0.0f, 0.0f, 0.0f, fw_ld1(buf + FW_OFF_L(C, n_lplanes), idx));
    return fw_ldq(buf + FW_OFF_Q3(C), C, idx, cpl);
}
// ... and the lifetime alone
__device__ __forceinline__ float fw_load_lifetime(const char *buf, uint32_t C, uint32_t n_lplanes, uint32_t idx, bool nospin, bool cpl) {
    if (nospin) return fw_ld1(buf + FW_OFF_L(C, n_lplanes), idx);
    return fw_ldq_w(buf + FW_OFF_Q3(C), C, idx, cpl);
}

// alive test of update_particles: `if particle.age >= particle.lifetime { destroyed }` (core.rs:594-599)
__device__ __forceinline__ bool fw_survives(float age, float dt, float lifetime, float *age_new) {
    float a = age + dt;
    *age_new = a;
    return !(a >= lifetime);
}

__device__ __forceinline__ void fw_stage_keys(float *s_keys, const FwGlobals &g, const FwType &T) {
    for (uint32_t i = threadIdx.x; i < T.keys_len; i += FW_BLOCK) s_keys[i] = g.keys[T.keys_off + i];
}

// ---------------------------------------------------------------------------------
// spawn: one new ParticleData (reference src/core.rs:437-469 Global, 506-544 Nested)
// ---------------------------------------------------------------------------------

struct FwSpawnOut {
    float4 q0, q1, q2, q3;
};

__device__ __forceinline__ fw_v3 fw_randvec3(float mag_min, float mag_max, float spread, const float dir[4],
                                             const float arc[4], float u_angle, float u_radius, float u_mag) {
    fw_v3 d;
    if (spread > 0.0f) {  // cone of half-angle `spread` around `direction` (bevy_utilitarian RandVec3)
        float spread_angle = u_angle * 2.0f * FW_PI;
        float spread_radius = u_radius * spread;
        float sr, cr, sa, ca;
        sincosf(spread_radius, &sr, &cr);
        sincosf(spread_angle, &sa, &ca);
        fw_v3 local{sr * ca, cr, sr * sa};
        d = fw_quat_mul_vec3(fw_q4{arc[0], arc[1], arc[2], arc[3]}, local);
    } else {
        d = fw_v3{dir[0], dir[1], dir[2]};
    }
    float m = u_mag * (mag_max - mag_min) + mag_min;  // RandF32::generate
    return fw_v3{d.x * m, d.y * m, d.z * m};
}

__device__ __forceinline__ FwSpawnOut fw_spawn_one(const FwEmit &e, uint32_t seed, unsigned long long serial,
                                                   fw_v3 origin_pos, fw_q4 origin_rot, fw_v3 inherit_vel, float speed,
                                                   float scale_mod) {
    float u[12];
#pragma unroll
    for (uint32_t b = 0; b < 3; b++) {
        fw_u4 o = fw_philox4x32_10(fw_u4{(uint32_t)serial, (uint32_t)(serial >> 32), e.emission_index, b}, seed, e.uid);
        u[4 * b + 0] = fw_unit_f32(o.x);
        u[4 * b + 1] = fw_unit_f32(o.y);
        u[4 * b + 2] = fw_unit_f32(o.z);
        u[4 * b + 3] = fw_unit_f32(o.w);
    }
    // EmissionShape::generate_point (emission_shape.rs:18-39)
    fw_v3 off{0.0f, 0.0f, 0.0f};
    if (e.shape_kind == 1) {
        float pitch = u[0] * 2.0f * FW_PI, yaw = u[1] * FW_PI, r = u[2];
        float sp, cp, sy, cy;
        sincosf(pitch, &sp, &cp);
        sincosf(yaw, &sy, &cy);
        fw_v3 unit{cp * sy, sp, cp * cy};
        off = fw_v3{unit.x * r * e.shape_radius, unit.y * r * e.shape_radius, unit.z * r * e.shape_radius};
    } else if (e.shape_kind == 2) {
        float ang = u[0] * 2.0f * FW_PI, r = u[1];
        float h = ang * 0.5f, sh, ch;
        sincosf(h, &sh, &ch);
        fw_q4 q2{0.0f, sh, 0.0f, ch};  // Quat::from_rotation_y
        fw_q4 q = fw_quat_mul(fw_q4{e.shape_arc[0], e.shape_arc[1], e.shape_arc[2], e.shape_arc[3]}, q2);
        off = fw_quat_mul_vec3(q, fw_v3{r * e.shape_radius, 0.0f, 0.0f});
    }
    // velocity (core.rs:440-448)
    fw_v3 vr = fw_randvec3(e.v_mag_min, e.v_mag_max, e.v_spread, e.v_dir, e.v_arc, u[3], u[4], u[5]);
    fw_v3 rv = fw_quat_mul_vec3(origin_rot, vr);
    fw_v3 n = fw_normalize_or_zero(off);
    float radial = u[6] * (e.radial_max - e.radial_min) + e.radial_min;
    float ix = e.inherit ? inherit_vel.x : 0.0f, iy = e.inherit ? inherit_vel.y : 0.0f,
          iz = e.inherit ? inherit_vel.z : 0.0f;
    float vx = speed * (rv.x + n.x * radial) + ix;
    float vy = speed * (rv.y + n.y * radial) + iy;
    float vz = speed * (rv.z + n.z * radial) + iz;
    float iscale = (u[7] * (e.iscale_max - e.iscale_min) + e.iscale_min) * scale_mod;  // core.rs:450-451
    float life = u[8] * (e.life_max - e.life_min) + e.life_min;                        // core.rs:455
    fw_v3 w = fw_randvec3(e.w_mag_min, e.w_mag_max, e.w_spread, e.w_dir, e.w_arc, u[9], u[10], u[11]);
    FwSpawnOut o;
    o.q0 = make_float4(origin_pos.x + off.x, origin_pos.y + off.y, origin_pos.z + off.z, 0.0f);
    o.q1 = make_float4(vx, vy, vz, iscale);
    o.q2 = make_float4(e.init_rot[0], e.init_rot[1], e.init_rot[2], e.init_rot[3]);
    o.q3 = make_float4(w.x, w.y, w.z, life);
    return o;
}

__device__ __forceinline__ void fw_store_new(const FwGlobals &g, const FwSeg &S, char *buf, uint32_t slot,
                                             const FwSpawnOut &o) {
    const uint32_t C = S.capacity;
    const FwType &T = g.types[S.type_idx];
    const float *keys = g.keys + T.keys_off;
    float bc[4], em[4];  // gradient.sample_clamped(0.) (core.rs:460-461)
    fw_gradient_sample(T.bc_kind, T.bc_n, keys + T.o_bc_t, keys + T.o_bc_v, 0.0f, bc);
    fw_gradient_sample(T.em_kind, T.em_n, keys + T.o_em_t, keys + T.o_em_v, 0.0f, em);
    fw_st4(buf + FW_OFF_Q0(C), slot, o.q0);
    fw_stq(buf + FW_OFF_Q1(C), C, slot, o.q1, S.cpl != 0u);  // (a ring's Q1 / Q3: component planes)
    fw_st4(buf + FW_OFF_Q2(C), slot, o.q2);
    fw_stq(buf + FW_OFF_Q3(C), C, slot, o.q3, S.cpl != 0u);
    if (T.flags & FW_TYPE_NOSPIN) fw_st1(buf + FW_OFF_L(C, S.n_lplanes), slot, o.q3.w);  // the lifetime plane (FwOutWin::lf)
    fw_st4(buf + FW_OFF_Q5(C), slot, make_float4(bc[0], bc[1], bc[2], bc[3]));
    fw_st4(buf + FW_OFF_Q6(C), slot, make_float4(em[0], em[1], em[2], em[3]));
    fw_st1(buf + FW_OFF_S4(C), slot, o.q1.w);  // scale = initial_scale
    for (uint32_t k = 0; k < S.n_lplanes; k++) fw_st1(buf + FW_OFF_L(C, k), slot, FW_F32_MIN);  // core.rs:467
}

// last_emitted_age planes of a particle spawned inside a ring's update kernel (FwSeg::lplane_emit): f32::MIN (core.rs:467), or --
// the frame's Nested pass would have visited the new particle, entry order permitting (core.rs:377-428: entries run in index
// order, the pass sees what earlier entries pushed) -- what that visit leaves behind: compute_emission_count(0, f32::MIN, ..)
// emits nothing for offsets >= 0 and returns `next` (core.rs:490-500), evaluated here with the same function
__device__ __forceinline__ void fw_init_last_emitted(const FwGlobals &g, const FwSeg &S, char *buf, uint32_t slot,
                                                     uint32_t new_emission_index, float lifetime) {
    const uint32_t C = S.capacity;
    for (uint32_t k = 0; k < S.n_lplanes; k++) {
        float v = FW_F32_MIN;
        const uint32_t ei = k < 2u ? S.lplane_emit[k] : 0xFFFFFFFFu;
        if (ei != 0xFFFFFFFFu) {
            const FwEmit &e = g.emits[ei];
            if (new_emission_index < e.emission_index) fw_emission_count(0.0f, FW_F32_MIN, lifetime, e.n_start, e.n_end, e.n_count, &v);
        }
        fw_st1(buf + FW_OFF_L(C, k), slot, v);
    }
}

// ---------------------------------------------------------------------------------
// update_particles (reference src/core.rs:577-670) with fused stable compaction
// ---------------------------------------------------------------------------------

__device__ __forceinline__ unsigned long long fw_pack_status(uint32_t epoch, uint32_t state, uint32_t value) {
    return ((unsigned long long)epoch << 34) | ((unsigned long long)state << 32) | value;
}

// integrate one surviving particle and store it at `o` of the output buffer (core.rs:601-657)
// Quat::from_scaled_axis(w * dt) for the per-frame rotation step (core.rs:645-647).  glam evaluates
// (v / |v|) * sin(|v| / 2), cos(|v| / 2); with h = |v| / 2 that is v * (sin(h) / 2h) and cos(h), both even functions
// of h, so for the small angles of a frame step (h^2 < 0.6, i.e. |w| dt < 89 degrees) two short polynomials in
// h^2 = |v|^2 / 4 give the quaternion without the square root, the three divisions and the sin/cos range reduction
// (truncation error < 3e-8 relative, below fp32 rounding; the zero axis comes out as the identity by itself).
// Rotation is one of the fields compared at 1e-5 (the CPU oracle's libm sin/cos already differs from the device's in
// the last bit).  Larger angles, NaN and infinities take the reference formula.
__device__ __forceinline__ fw_q4 fw_quat_step(fw_v3 v) {
    const float h2 = 0.25f * ((v.x * v.x) + (v.y * v.y) + (v.z * v.z));
    if (__builtin_expect(__ballot(!(h2 < 0.6f)) == 0ull, 1)) {  // wave-uniform choice
        float sh = __builtin_fmaf(h2, 2.7557319e-6f, -1.9841270e-4f);   // 1/9!, -1/7!
        sh = __builtin_fmaf(h2, sh, 8.3333333e-3f);                     // 1/5!
        sh = __builtin_fmaf(h2, sh, -1.6666667e-1f);                    // -1/3!
        sh = __builtin_fmaf(h2, sh, 1.0f) * 0.5f;                       // sin(h) / (2 h)
        float c = __builtin_fmaf(h2, -2.7557319e-7f, 2.4801587e-5f);    // -1/10!, 1/8!
        c = __builtin_fmaf(h2, c, -1.3888889e-3f);                      // -1/6!
        c = __builtin_fmaf(h2, c, 4.1666667e-2f);                       // 1/4!
        c = __builtin_fmaf(h2, c, -0.5f);
        c = __builtin_fmaf(h2, c, 1.0f);                                // cos(h)
        return fw_q4{v.x * sh, v.y * sh, v.z * sh, c};
    }
    return fw_quat_from_scaled_axis(v);
}

// INPLACE (FIFO segments, fw_k_update_fifo): the output slot is the input slot, so a plane whose new value is
// bit-identical to the loaded one for every lane of the wave is not written (rotation and angular velocity of particles
// that do not spin, the scale under a constant curve); `full` marks a lane whose slot holds nothing yet (a particle
// spawned this frame): it writes everything.
// WM >= 0: which of the optional planes the launch writes is a compile-time fact (bit 0 base colour, 1 emissive, 2 scale)
// wmode (INPLACE only): where the particle's two constants -- initial_scale in q1.w, lifetime in q3.w -- are.  FW_W_REGS: in the
// arguments, and a `full` lane (generated in this kernel) stores them; FW_W_MEM: in the slot already (a particle of an earlier frame, or
// one another kernel materialised) and q1.w is valid; FW_W_MEM_LAZY: in the slot, q1.w NOT loaded -- the streaming loops leave it there
// unless somebody needs the scale (a type whose planes are stored, an instance record, the boxes): then it is read here.
enum { FW_W_REGS = 0, FW_W_MEM = 1, FW_W_MEM_LAZY = 2 };
// CPL: the output segment is a ring -- Q1 / Q3 are component planes (the ring kernels; everybody else writes float4 planes)
template <bool INPLACE = false, int WM = -1, int NT = 0, bool CPL = false>
__device__ __forceinline__ void fw_integrate_store(const FwType &T, const float *s_keys, float dt, float4 q0, float4 q1,
                                                   float4 q2, float4 q3, float age_new, const FwOutWin &W, uint32_t o,
                                                   float4 *rec = nullptr, const fw_v3 *cpos = nullptr,
                                                   const fw_v3 *cvel = nullptr, float *box = nullptr,
                                                   bool box_on = false, bool full = false, bool use_c = true, int wmode = FW_W_REGS) {
    if (T.flags & FW_TYPE_NOSPIN) q2 = make_float4(T.const_rot[0], T.const_rot[1], T.const_rot[2], T.const_rot[3]);
    const float lifetime = q3.w;
    // scale and colours (core.rs:601-605, 652-655).  A FW_TYPE_DERIVED type stores none of them: unless this launch writes an
    // instance record or tracks the boxes, nobody asks for them here -- the division, the curve and both gradients are skipped
    // (a workgroup-uniform branch on the type record: round 6, when that became the state of EVERY type; their readers evaluate
    // fw_derived_values from the stored age)
    const bool need_cs = !(T.flags & FW_TYPE_DERIVED) || rec != nullptr || box_on;
    float scale = 0.0f;
    float bc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, em[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (need_cs) {
        float iscale = q1.w;
        if (INPLACE && CPL && wmode == FW_W_MEM_LAZY) iscale = fw_ld1w<NT == 2>(W.q1 + 3 * W.cp, (o - W.first) * 4u);  // (workgroup-uniform branch)
        const float age_percent = age_new / lifetime;
        const float scale_factor = fw_curve_sample(T.sc_kind, T.sc_n, s_keys, s_keys + T.o_sc_v, age_percent);
        scale = iscale * scale_factor;
        fw_gradient_sample(T.bc_kind, T.bc_n, s_keys + T.o_bc_t, s_keys + T.o_bc_v, age_percent, bc);
        fw_gradient_sample(T.em_kind, T.em_n, s_keys + T.o_em_t, s_keys + T.o_em_v, age_percent, em);
    }
    // explicit Euler with the OLD velocity (core.rs:626-631, 641-643); cpos / cvel: what particle_collision returned
    // for a type with collision settings (core.rs:607-624) -- the velocity update then starts from the new velocity
    // (use_c: a runtime "this type collides" next to pointers that are null or not at compile time -- a pointer SELECTED at
    // run time between a local's address and null would force the local into scratch memory)
    const bool uc = cpos != nullptr && cvel != nullptr && use_c;
    const float ux = uc ? cvel->x : q1.x, uy = uc ? cvel->y : q1.y, uz = uc ? cvel->z : q1.z;
    const float px = uc ? cpos->x : q0.x + q1.x * dt, py = uc ? cpos->y : q0.y + q1.y * dt,
                pz = uc ? cpos->z : q0.z + q1.z * dt;
    const float vx = ux + (T.acc[0] - ux * T.lin_drag) * dt;
    const float vy = uy + (T.acc[1] - uy * T.lin_drag) * dt;
    const float vz = uz + (T.acc[2] - uz * T.lin_drag) * dt;
    // rotation = from_scaled_axis(angvel * dt) * rotation, no renormalisation (core.rs:645-647)
    // (a type that cannot turn, FW_TYPE_NOSPIN: angular velocity 0, from_scaled_axis(0) * const_rot = const_rot bit for bit -- its
    // components carry no negative zero, fw_engine_build.cpp -- and neither plane is stored: nothing to evaluate; workgroup-uniform)
    fw_q4 nr{q2.x, q2.y, q2.z, q2.w};
    float wx = 0.0f, wy = 0.0f, wz = 0.0f;
    if (!(T.flags & FW_TYPE_NOSPIN)) {
        const fw_q4 dq = fw_quat_step(fw_v3{q3.x * dt, q3.y * dt, q3.z * dt});
        nr = fw_quat_mul(dq, fw_q4{q2.x, q2.y, q2.z, q2.w});
        wx = q3.x + (T.angacc[0] - T.ang_drag * q3.x) * dt;  // core.rs:648-650
        wy = q3.y + (T.angacc[1] - T.ang_drag * q3.y) * dt;
        wz = q3.z + (T.angacc[2] - T.ang_drag * q3.z) * dt;
    }
    const uint32_t b16 = (o - W.first) * 16u;  // < 16 KiB + a tile: the window starts at the tile's first output slot
    fw_st4w<NT == 2>(W.q0, b16, make_float4(px, py, pz, age_new));
    const uint32_t b4 = (o - W.first) * 4u;
    if constexpr (CPL) fw_stc3w<NT == 2>(W.q1, W.cp, b4, vx, vy, vz);
    else fw_st4w<NT == 2>(W.q1, b16, make_float4(vx, vy, vz, q1.w));
    if (INPLACE) {
        static_assert(!INPLACE || CPL, "in place = a ring = component planes");
        // (initial_scale and lifetime never change: in place only a slot that holds nothing yet writes them -- wave-uniform branch)
        if (wmode == FW_W_REGS && __any(full)) {
            if (full) fw_st1w<NT == 2>(W.q1 + 3 * W.cp, b4, q1.w);
            if (full && W.wr3) fw_st1w<NT == 2>(W.q3 + 3 * W.cp, b4, lifetime);
        }
        const uint32_t d2 = (__float_as_uint(nr.x) ^ __float_as_uint(q2.x)) | (__float_as_uint(nr.y) ^ __float_as_uint(q2.y)) |
                            (__float_as_uint(nr.z) ^ __float_as_uint(q2.z)) | (__float_as_uint(nr.w) ^ __float_as_uint(q2.w));
        const uint32_t d3 = (__float_as_uint(wx) ^ __float_as_uint(q3.x)) | (__float_as_uint(wy) ^ __float_as_uint(q3.y)) |
                            (__float_as_uint(wz) ^ __float_as_uint(q3.z));
        if (W.wr2 && __any(full || d2 != 0u)) fw_st4w<NT == 2>(W.q2, b16, make_float4(nr.x, nr.y, nr.z, nr.w));  // wave-uniform branches
        if (W.wr3 && __any(full || d3 != 0u)) fw_stc3w<NT == 2>(W.q3, W.cp, b4, wx, wy, wz);
        // (`full` lanes -- slots that hold nothing yet -- write every plane the type MAINTAINS: W.wr4 is false exactly for FW_TYPE_DERIVED)
        const bool fullk = full && W.wr4;
        if ((WM >= 0 ? (WM & 1) != 0 : W.wr5) || fullk) fw_st4w<NT != 0>(W.q5, b16, make_float4(bc[0], bc[1], bc[2], bc[3]));
        if ((WM >= 0 ? (WM & 2) != 0 : W.wr6) || fullk) fw_st4w<NT != 0>(W.q6, b16, make_float4(em[0], em[1], em[2], em[3]));
        if ((WM >= 0 ? (WM & 4) != 0 : (T.sc_kind != 0 && W.wr4)) || fullk) fw_st1w<NT != 0>(W.s4, (o - W.first) * 4u, scale);
    } else {
        if (W.wr2) fw_st4w<NT == 2>(W.q2, b16, make_float4(nr.x, nr.y, nr.z, nr.w));
        if constexpr (CPL) {
            fw_st1w<NT == 2>(W.q1 + 3 * W.cp, b4, q1.w);  // (the old part of a range ring, its new particles: the constants are stored too)
            if (W.wr3) fw_stc4w<NT == 2>(W.q3, W.cp, b4, make_float4(wx, wy, wz, lifetime));
        } else {
            if (W.wr3) fw_st4w<NT == 2>(W.q3, b16, make_float4(wx, wy, wz, lifetime));
        }
        if (!W.wr3) fw_st1w<NT == 2>(W.lf, b4, lifetime);
        if (W.wr5) fw_st4w<NT != 0>(W.q5, b16, make_float4(bc[0], bc[1], bc[2], bc[3]));  // workgroup-uniform branches
        if (W.wr6) fw_st4w<NT != 0>(W.q6, b16, make_float4(em[0], em[1], em[2], em[3]));
        if (W.wr4) fw_st1w<NT != 0>(W.s4, (o - W.first) * 4u, scale);
    }
    if (box_on) {  // update_aabbs (render.rs:677-703): running min / max of position -/+ scale, per lane
        // (`box` always points at the caller's local array when box_on can be true: never selected against null, so it
        // stays in registers)
        box[0] = fminf(box[0], px - scale), box[1] = fminf(box[1], py - scale), box[2] = fminf(box[2], pz - scale);
        box[3] = fmaxf(box[3], px + scale), box[4] = fmaxf(box[4], py + scale), box[5] = fmaxf(box[5], pz + scale);
    }
    if (rec) {  // ParticleInstance {pos.xyz, scale, rot, base_color, emissive} (render.rs:95-103); `rec` may be in LDS
        rec[0] = make_float4(px, py, pz, scale), rec[1] = make_float4(nr.x, nr.y, nr.z, nr.w);
        rec[2] = make_float4(bc[0], bc[1], bc[2], bc[3]), rec[3] = make_float4(em[0], em[1], em[2], em[3]);
    }
}

// Render hand-off fused into the update: the ParticleInstance records of a wave's survivors of one round occupy
// consecutive slots [wbase, wbase + cnt), i.e. one contiguous run of cnt * 64 bytes.  The lanes park their records in
// a wave-private LDS area at their rank and the wave then stores the run with fully coalesced float4 stores.
template <bool NT = false>
__device__ __forceinline__ void fw_inst_flush(char *inst, uint32_t inst_cap, const float4 *s_wave, uint32_t lane,
                                              unsigned long long m, uint32_t wbase) {
    const uint32_t cnt = (uint32_t)__popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t wb = __builtin_amdgcn_readfirstlane(wbase);
    const uint32_t room4 = (wb < inst_cap ? min(cnt, inst_cap - wb) : 0u) * 4u;
    char *dst = inst + (size_t)wb * 64u;
#pragma unroll 1  // one float4 in registers at a time: the kernel sits at the 128-VGPR occupancy step
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t e = k * 64u + lane;
        if (e < room4) fw_st4w<NT>(dst, e * 16u, s_wave[e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


// the rotation a record of a particle carries: the plane's value, or -- FW_TYPE_NOSPIN: the plane is neither read nor
// maintained -- the type's one rotation.  Shared by every writer of destroyed records.
__device__ __forceinline__ float4 fw_record_rotation(const FwType &T, float4 q2) {
    if (T.flags & FW_TYPE_NOSPIN) return make_float4(T.const_rot[0], T.const_rot[1], T.const_rot[2], T.const_rot[3]);
    return q2;
}

// scale, base colour and emissive colour of a particle as the update that produced its stored `age` computed them
// (core.rs:601-605, 652-655): what the S4 / Q5 / Q6 planes hold -- or would hold, for a FW_TYPE_DERIVED type
__device__ __forceinline__ void fw_derived_values(const FwType &T, const float *keys, float age, float lifetime, float initial_scale,
                                                  float4 *bc, float4 *em, float *sc) {
    const float ap = age / lifetime;
    float b4[4], e4[4];
    fw_gradient_sample(T.bc_kind, T.bc_n, keys + T.o_bc_t, keys + T.o_bc_v, ap, b4);
    fw_gradient_sample(T.em_kind, T.em_n, keys + T.o_em_t, keys + T.o_em_v, ap, e4);
    *bc = make_float4(b4[0], b4[1], b4[2], b4[3]), *em = make_float4(e4[0], e4[1], e4[2], e4[3]);
    *sc = initial_scale * fw_curve_sample(T.sc_kind, T.sc_n, keys, keys + T.o_sc_v, ap);
}

// destroyed record = the clone with age already advanced, pose of the previous frame (core.rs:596-599)
__device__ __forceinline__ void fw_store_destroyed(char *dbuf, const char *ib, uint32_t C, uint32_t idx, bool loaded,
                                                   const FwType &T, const float *s_keys, float4 q0, float4 q1,
                                                   float4 q2, float4 q3, float age_new, uint32_t d) {
    float *rec = reinterpret_cast<float *>(dbuf) + (size_t)d * 26;
    q2 = fw_record_rotation(T, q2);
    const int32_t pbr = T.pbr;
    float4 bc, em;
    float sc;
    if (loaded && (T.flags & FW_TYPE_DERIVED)) {  // the planes are not maintained: what the previous update computed, again
        fw_derived_values(T, s_keys, q0.w, q3.w, q1.w, &bc, &em, &sc);
    } else if (loaded) {
        bc = fw_ld4(ib + FW_OFF_Q5(C), idx), em = fw_ld4(ib + FW_OFF_Q6(C), idx);
        sc = reinterpret_cast<const float *>(ib + FW_OFF_S4(C))[idx];
    } else {  // born and destroyed in the same frame: spawn-time colours and scale (core.rs:457-461)
        float b4[4], e4[4];
        fw_gradient_sample(T.bc_kind, T.bc_n, s_keys + T.o_bc_t, s_keys + T.o_bc_v, 0.0f, b4);
        fw_gradient_sample(T.em_kind, T.em_n, s_keys + T.o_em_t, s_keys + T.o_em_v, 0.0f, e4);
        bc = make_float4(b4[0], b4[1], b4[2], b4[3]), em = make_float4(e4[0], e4[1], e4[2], e4[3]);
        sc = q1.w;
    }
    rec[0] = q0.x, rec[1] = q0.y, rec[2] = q0.z;
    rec[3] = q1.x, rec[4] = q1.y, rec[5] = q1.z;
    rec[6] = q2.x, rec[7] = q2.y, rec[8] = q2.z, rec[9] = q2.w;
    rec[10] = q3.x, rec[11] = q3.y, rec[12] = q3.z;
    rec[13] = q1.w, rec[14] = sc, rec[15] = age_new, rec[16] = q3.w;
    rec[17] = bc.x, rec[18] = bc.y, rec[19] = bc.z, rec[20] = bc.w;
    rec[21] = em.x, rec[22] = em.y, rec[23] = em.z, rec[24] = em.w;
    reinterpret_cast<int32_t *>(rec)[25] = pbr;
}

// AABB fused into the update (render.rs:677-703 reads every particle twice on the CPU each frame): the
// lanes keep a running box of position -/+ scale over the survivors they store, the workgroup folds the lane boxes once
// at the end of the tile and leaves {min.xyz, epoch, max.xyz, -} in its slot of a per-tile array.  fw_spawner_aabb then
// folds a few hundred 32-byte tile boxes instead of re-reading 20 bytes of every particle.  min / max are exact and
// order-independent: the result is bit-identical to the two-pass query.  `s_box`: NW x 6 floats of LDS.
template <int NW>
__device__ __forceinline__ void fw_tile_box_flush(float *tile_box, uint32_t tile, uint32_t epoch, const float (&box)[6],
                                                  float (*s_box)[6]) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        v[c] = box[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(v[c], o, 64);
            v[c] = c < 3 ? fminf(v[c], other) : fmaxf(v[c], other);
        }
    }
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 6; c++) s_box[wave][c] = v[c];
    __syncthreads();
    if (tid == 0) {
        float r[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            r[c] = s_box[0][c];
#pragma unroll
            for (int w = 1; w < NW; w++) r[c] = c < 3 ? fminf(r[c], s_box[w][c]) : fmaxf(r[c], s_box[w][c]);
        }
        float4 *dst = reinterpret_cast<float4 *>(tile_box) + (size_t)tile * 2;
        dst[0] = make_float4(r[0], r[1], r[2], __uint_as_float(epoch));
        dst[1] = make_float4(r[3], r[4], r[5], 0.0f);
    }
}

// SPAWN selects where this frame's Global spawn ops come from: none (already materialised by
// fw_k_spawn), the kernel arguments (small frames) or a device table (many emitters).  Spawned
// particles are "virtual" inputs with index >= the live count: generated in registers from the
// counter RNG, then integrated, compacted and stored like loaded ones (spawn runs before update
// in the same frame, reference src/plugin.rs:46-60) -- they never cost an extra HBM round trip.
#define FW_OP(i) (SPAWN == FW_SPAWN_INLINE ? inl.ops[i] : a.ops[i])

// ---- decoupled look-back over the tiles [lo, tile) of the status array ----------------------------
// Every lane fetches LBW status words with all loads in flight at once, so a step costs one memory
// round trip and covers LBW * BLK tiles.  Returns the exclusive sum; sets *timed_out (block-uniform)
// when a predecessor did not publish within the spin limit.
template <int BLK, int NW, int LBW>
__device__ __forceinline__ uint32_t fw_lookback(const unsigned long long *status, uint32_t lo, uint32_t tile,
                                                uint32_t epoch, uint32_t spin_limit, uint32_t *s_lb, bool *timed_out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t excl = 0;
    uint32_t pos = tile - 1u;
    bool to = false;
    for (;;) {
        unsigned long long wd[LBW];
        bool has[LBW];
#pragma unroll
        for (int j = 0; j < LBW; j++) {
            has[j] = pos >= lo + tid + (uint32_t)j * BLK;
            wd[j] = has[j] ? __hip_atomic_load(&status[pos - tid - (uint32_t)j * BLK], RLX, AGENT) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < LBW; j++) {
            uint32_t st = 0, val = 0;
            if (has[j]) {
                uint32_t spins = 0;
                while ((uint32_t)(wd[j] >> 34) != epoch) {
                    if (++spins > spin_limit) {
                        to = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    wd[j] = __hip_atomic_load(&status[pos - tid - (uint32_t)j * BLK], RLX, AGENT);
                }
                st = (uint32_t)(wd[j] >> 32) & 3u;
                val = (uint32_t)wd[j];
            }
            const unsigned long long incl = __ballot(has[j] && st == FW_ST_INCL);
            bool use = has[j];
            if (incl) use = has[j] && lane <= (uint32_t)(__ffsll((long long)incl) - 1);
            const uint32_t wsum = fw_wave_sum(use ? val : 0u);
            if (lane == 0) {
                s_lb[j * NW + wave] = wsum;
                s_lb[LBW * NW + j * NW + wave] = incl ? 1u : 0u;
            }
        }
        if (__syncthreads_or(to ? 1 : 0)) {
            to = true;
            break;
        }
        bool found = false;
#pragma unroll
        for (int w = 0; w < LBW * NW; w++) {  // nearest sub-window first, nearest wave first
            if (!found) {
                excl += s_lb[w];
                found = s_lb[LBW * NW + w] != 0u;
            }
        }
        __syncthreads();
        if (found || pos < lo + LBW * BLK) break;
        pos -= LBW * BLK;
    }
    *timed_out = to;
    return excl;
}

// ---- survivor forecast sums (FwUpdateArgs::fc_*) ----------------------------------------------------
// A tile's survivors-of-the-next-step land in output tile A (sa of them) and A + 1 (sb).  Both go into ONE 64-bit word
// P[A] = {lo += sa, hi += sb} with a single atomic, so the next frame's count of input tile t is lo(P[t]) + hi(P[t-1])
// and the prefix a tile needs is   sum_{t < tis} (lo + hi)(P[t])  -  hi(P[tis - 1]).
// Device-scope atomics execute at the memory side: each costs the kernel about 0.3 ns of wall time at 1M particles
// (measured by doubling them), and counters sharing a cache line serialise, so the per-group sums P2 (one per 64
// tiles, large segments only) have a 64-byte line each; segments of up to FW_FC_DIRECT tiles sum P directly.
typedef unsigned long long fw_u64;
__device__ __forceinline__ uint2 fw_ld2u(const fw_u64 *base, uint32_t idx) {
    typedef uint32_t fw_u2v __attribute__((ext_vector_type(2)));
    const fw_u2v v = reinterpret_cast<const FW_GLOBAL fw_u2v *>(reinterpret_cast<uintptr_t>(base))[idx];
    return make_uint2(v.x, v.y);
}
// this lane's share of the prefix over tiles [lo, hi) (lo = the segment's first tile); the hi(P[hi-1]) correction is
// applied by the lane that holds it; all loads are issued unconditionally at a clamped index
template <int BLK>
__device__ __forceinline__ uint32_t fw_fc_prefix_part(const fw_u64 *fc, uint32_t s2, uint32_t lo, uint32_t hi,
                                                      uint32_t seg_tiles) {
    const uint32_t tid = threadIdx.x;
    uint32_t nL, nG = 0u, nR = 0u, gl = 0u, gh = 0u;
    if (seg_tiles <= FW_FC_DIRECT) {
        nL = hi - lo;
    } else {
        gl = (lo + 63u) >> 6, gh = hi >> 6;
        if (gl >= gh) nL = hi - lo;
        else nL = gl * 64u - lo, nG = gh - gl, nR = hi - gh * 64u;
    }
    const uint32_t n = nL + nG + nR;
    const uint2 last = fw_ld2u(fc, hi > lo ? hi - 1u : lo);  // every lane loads it (one line), lane 0 uses it
    uint32_t part = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += BLK) {
        const uint32_t i = i0 + tid;
        const uint32_t idx = i < nL ? lo + i : (i < nL + nG ? s2 + (gl + (i - nL)) * FW_FC_S2_STRIDE : gh * 64u + (i - nL - nG));
        const uint2 v = fw_ld2u(fc, i < n ? idx : lo);
        part += i < n ? v.x + v.y : 0u;
    }
    if (tid == 0 && hi > lo) part -= last.y;
    return part;
}
// a tile's contribution (global tile index A; P2 only for large segments)
__device__ __forceinline__ void fw_fc_add(fw_u64 *fc, uint32_t s2, uint32_t A, uint32_t sa, uint32_t sb, uint32_t seg_tiles) {
    if (sa | sb) {
        const fw_u64 v = (fw_u64)sa | ((fw_u64)sb << 32);
        atomicAdd(&fc[A], v);
        if (seg_tiles > FW_FC_DIRECT) atomicAdd(&fc[s2 + (A >> 6) * FW_FC_S2_STRIDE], v);
    }
}
// small segments: the forecast is one entry per tile {into A, into A + 1, A, epoch}; a tile adds what its predecessors
// put into the tiles before it.  FW_FC_DIRECT / BLK entries per lane, requested up front (fw_fce_request) at a clamped
// index, consumed here.
constexpr int FW_FCE_U = 8;
template <int BLK>
__device__ __forceinline__ void fw_fce_request(const uint4 *fce_in, uint32_t first, uint32_t seg_tiles, uint4 (&e)[FW_FCE_U]) {
    static_assert(FW_FCE_U * BLK >= (int)FW_FC_DIRECT, "entries per lane must cover a small segment");
#pragma unroll
    for (int j = 0; j < FW_FCE_U; j++) {
        const uint32_t t = threadIdx.x + (uint32_t)j * BLK;
        e[j] = fw_ld4u(reinterpret_cast<const char *>(fce_in + first), min(t, seg_tiles - 1u) * 16u);
    }
}
template <int BLK>
__device__ __forceinline__ uint32_t fw_fce_prefix_part(const uint4 (&e)[FW_FCE_U], uint32_t seg_tiles, uint32_t tis,
                                                       uint32_t epoch, bool *bad) {
    uint32_t part = 0;
    bool b = false;
#pragma unroll
    for (int j = 0; j < FW_FCE_U; j++) {
        const bool in = threadIdx.x + (uint32_t)j * BLK < seg_tiles;  // beyond the table: a clamped duplicate, ignored
        b |= in && e[j].w != epoch - 1u;
        part += in ? (e[j].z + 1u < tis ? e[j].x + e[j].y : (e[j].z < tis ? e[j].x : 0u)) : 0u;
    }
    *bad = b;
    return part;
}

// Threshold forecast, producer side (FwUpdateArgs::fc_theta): a stored survivor that a step of `theta` would destroy leaves its
// (age, lifetime) in the tile's list -- from the front when it was stored into output tile A (o < fc_bnd), from the back otherwise.
// s_tf: two LDS counters of the workgroup (zeroed before the first round).  Wave-uniform fast exit: nobody is risky in most rounds.
__device__ __forceinline__ void fw_tf_note(float theta, float2 *list, uint32_t *s_tf, bool alive, float age_new, float lifetime,
                                           uint32_t o, uint32_t fc_bnd) {
    const bool risky = alive && (age_new + theta >= lifetime);  // == !fw_survives(age_new, theta, lifetime)
    const unsigned long long mr = __ballot(risky);
    if (mr == 0ull) return;
    const bool ra = risky && o < fc_bnd, rb = risky && !(o < fc_bnd);
    const unsigned long long ma = __ballot(ra), mb = __ballot(rb);
    uint32_t ba = 0, bb = 0;
    if ((threadIdx.x & 63u) == 0u) {
        if (ma) ba = atomicAdd(&s_tf[0], (uint32_t)__popcll(ma));
        if (mb) bb = atomicAdd(&s_tf[1], (uint32_t)__popcll(mb));
    }
    ba = __builtin_amdgcn_readfirstlane(ba), bb = __builtin_amdgcn_readfirstlane(bb);
    if (ra) {
        const uint32_t i = ba + fw_lane_prefix(ma);
        if (i < FW_TF_K) list[i] = make_float2(age_new, lifetime);
    } else if (rb) {
        const uint32_t j = bb + fw_lane_prefix(mb);
        if (j < FW_TF_K) list[FW_TF_K - 1u - j] = make_float2(age_new, lifetime);
    }
}
// ... and the tile's header once all its rounds are done (one thread; the counters are final: a barrier lies in between)
__device__ __forceinline__ void fw_tf_header(uint4 *fct, uint32_t tile, const uint32_t *s_tf, uint32_t excl, uint32_t run, uint32_t fc_bnd) {
    const uint32_t ta = min(run, fc_bnd) - min(excl, fc_bnd), tb = (run - excl) - ta;
    const uint32_t na = s_tf[0], nb = s_tf[1];
    fct[tile] = make_uint4(ta, tb, (na + nb > FW_TF_K) ? 0xFFFFFFFFu : (na | (nb << 16)), excl);
}

// every workgroup (active or not) clears its own slot of the buffer the frame after the next will accumulate into
__device__ __forceinline__ void fw_fc_housekeeping(const FwUpdateArgs &a) {
    if (threadIdx.x == 0 && a.fc_zero) {
        a.fc_zero[blockIdx.x] = 0ull;
        if ((blockIdx.x & 63u) == 0u) a.fc_zero[a.fc_s2 + (blockIdx.x >> 6) * FW_FC_S2_STRIDE] = 0ull;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0 && a.fc_out) a.fc_out[a.fc_tag] = (fw_u64)a.epoch;
}



// Launch with optional timing events attached to the dispatch itself (hipExtLaunchKernel): the events take the
// packet's own begin / end timestamps, which is what rocprofv3 --kernel-trace reports for the kernel.
// (fw_dyn_lds: experiment knob FW_DYN_LDS -- unused dynamic LDS per workgroup lowers the number of resident workgroups
// per CU; measured on the HBM-resident configurations)
static unsigned fw_dyn_lds = (getenv("FW_ENABLE_KNOBS") && atoi(getenv("FW_ENABLE_KNOBS")) && getenv("FW_DYN_LDS")) ? (unsigned)atoi(getenv("FW_DYN_LDS")) : 0u;
#define FW_LAUNCH_T(kern, grid, block, s, e0, e1, ...)                                          \
    do {                                                                                          \
        if ((e0) || (e1))                                                                         \
            hipExtLaunchKernelGGL(kern, grid, block, fw_dyn_lds, s, e0, e1, 0, __VA_ARGS__);      \
        else                                                                                      \
            hipLaunchKernelGGL(kern, grid, block, fw_dyn_lds, s, __VA_ARGS__);                    \
    } while (0)

