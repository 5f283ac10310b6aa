// fw_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the firework particle backend.
//
// The path is a streaming fp32 update: ~0.5 flop/byte, far below the CDNA4 ridge, so
// there is no MFMA here; every kernel is designed around HBM traffic:
//   * one particle per lane, float4 planes (fw_device.h) -> every global access is a
//     full-width dwordx4, 1 KiB per wave instruction, and stays 16-B aligned after
//     stable compaction;
//   * per-type constants arrive as scalar loads (block-uniform), curve / gradient keys
//     are staged in LDS once per workgroup;
//   * dead-particle compaction is order preserving (the reference's filter_map().collect()
//     src/core.rs:589-659): wave64 ballot + mbcnt inside a wave, LDS across the four
//     waves, and a single-pass decoupled look-back across workgroups whose status word
//     carries its own epoch tag (8-byte agent-scope granule: no fences, no per-frame
//     memset).  A bounded spin falls back to recomputing the prefix locally, so the
//     kernel cannot deadlock whatever the dispatch order is.
//
// Arithmetic order is the reference's (fw_math.h); built with -ffp-contract=off.
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
__device__ __forceinline__ void fw_raise(const FwGlobals &g, uint32_t check, uint32_t seg, uint32_t x) {
    atomicOr(g.err, FW_ERR_FORECAST);
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
    char *q0, *q1, *q2, *q3, *q5, *q6, *s4;
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
                                                  uint32_t n_lplanes = 0u) {
    const size_t f16 = (size_t)first * 16u;
    return FwOutWin{ob + FW_OFF_Q0(C) + f16, ob + FW_OFF_Q1(C) + f16, ob + FW_OFF_Q2(C) + f16, ob + FW_OFF_Q3(C) + f16,
                    ob + FW_OFF_Q5(C) + f16, ob + FW_OFF_Q6(C) + f16, ob + FW_OFF_S4(C) + (size_t)first * 4u, first,
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
__device__ __forceinline__ float4 fw_load_q3(const char *buf, uint32_t C, uint32_t n_lplanes, uint32_t idx, bool nospin) {
    if (nospin) return make_float4(0.0f, 0.0f, 0.0f, fw_ld1(buf + FW_OFF_L(C, n_lplanes), idx));
    return fw_ld4(buf + FW_OFF_Q3(C), idx);
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
    fw_st4(buf + FW_OFF_Q1(C), slot, o.q1);
    fw_st4(buf + FW_OFF_Q2(C), slot, o.q2);
    fw_st4(buf + FW_OFF_Q3(C), slot, o.q3);
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

// Global emission: ops[] lists this frame's (segment, entry, count) triples; op i owns
// workgroups [first_block_i, first_block_{i+1}).
// (`ops` = device table, or null: the ops ride in the kernel arguments -- no staging copy, no event in the stream)
__global__ __launch_bounds__(FW_BLOCK) void fw_k_spawn(FwGlobals g, FwInlineOps inl, const FwOp *ops, uint32_t n_ops,
                                                      uint32_t parity) {
    uint32_t lo = 0, hi = n_ops;  // op lookup (block-uniform)
    if (ops) {
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (ops[mid].first_block <= blockIdx.x)
                lo = mid;
            else
                hi = mid;
        }
    } else {
        for (uint32_t i = 1; i < n_ops; i++)
            if (inl.ops[i].first_block <= blockIdx.x) lo = i;
    }
    const FwOp &op = ops ? ops[lo] : inl.ops[lo];
    const uint32_t k = (blockIdx.x - op.first_block) * FW_BLOCK + threadIdx.x;
    const uint32_t sidx = parity * g.max_seg + op.seg;
    const FwSeg &S = g.segs[op.seg];
    const uint32_t base = g.count[sidx] + g.appended[sidx] + op.rel_base;
    // Only what fits is counted: the update sizes its input from count + spawned + appended and must never see more
    // than `capacity` particles (types that also receive Nested children cannot be grown by the host: their count is
    // only known on the device).  Ops of one launch own disjoint slot ranges [base, base + n), so the clamps add up.
    const uint32_t room = base < S.capacity ? S.capacity - base : 0u;
    if (k == 0) {
        atomicAdd(&g.spawned[sidx], min(op.n, room));
        if (op.n > room) atomicOr(g.err, FW_ERR_CAPACITY);
    }
    if (k >= op.n || k >= room) return;
    const uint32_t head = op.range_ring ? fw_range_head(op.head, g.rold[sidx], S.capacity) : op.head;
    const uint32_t slot = fw_ring_slot(head, base + k, S.capacity);  // (base + k < capacity)
    const FwEmit &e = g.emits[op.emit];
    FwSpawnOut o = fw_spawn_one(e, g.seed, op.serial_base + k, fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                                fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                                fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
    fw_store_new(g, S, S.buf[parity], slot, o);
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
template <bool INPLACE = false, int WM = -1, int NT = 0>
__device__ __forceinline__ void fw_integrate_store(const FwType &T, const float *s_keys, float dt, float4 q0, float4 q1,
                                                   float4 q2, float4 q3, float age_new, const FwOutWin &W, uint32_t o,
                                                   float4 *rec = nullptr, const fw_v3 *cpos = nullptr,
                                                   const fw_v3 *cvel = nullptr, float *box = nullptr,
                                                   bool box_on = false, bool full = false) {
    if (T.flags & FW_TYPE_NOSPIN) q2 = make_float4(T.const_rot[0], T.const_rot[1], T.const_rot[2], T.const_rot[3]);
    const float lifetime = q3.w;
    const float age_percent = age_new / lifetime;
    const float scale_factor = fw_curve_sample(T.sc_kind, T.sc_n, s_keys, s_keys + T.o_sc_v, age_percent);
    const float scale = q1.w * scale_factor;
    // explicit Euler with the OLD velocity (core.rs:626-631, 641-643); cpos / cvel: what particle_collision returned
    // for a type with collision settings (core.rs:607-624) -- the velocity update then starts from the new velocity
    const float ux = cvel ? cvel->x : q1.x, uy = cvel ? cvel->y : q1.y, uz = cvel ? cvel->z : q1.z;
    const float px = cpos ? cpos->x : q0.x + q1.x * dt, py = cpos ? cpos->y : q0.y + q1.y * dt,
                pz = cpos ? cpos->z : q0.z + q1.z * dt;
    const float vx = ux + (T.acc[0] - ux * T.lin_drag) * dt;
    const float vy = uy + (T.acc[1] - uy * T.lin_drag) * dt;
    const float vz = uz + (T.acc[2] - uz * T.lin_drag) * dt;
    // rotation = from_scaled_axis(angvel * dt) * rotation, no renormalisation (core.rs:645-647)
    const fw_q4 dq = fw_quat_step(fw_v3{q3.x * dt, q3.y * dt, q3.z * dt});
    const fw_q4 nr = fw_quat_mul(dq, fw_q4{q2.x, q2.y, q2.z, q2.w});
    const float wx = q3.x + (T.angacc[0] - T.ang_drag * q3.x) * dt;  // core.rs:648-650
    const float wy = q3.y + (T.angacc[1] - T.ang_drag * q3.y) * dt;
    const float wz = q3.z + (T.angacc[2] - T.ang_drag * q3.z) * dt;
    float bc[4], em[4];
    fw_gradient_sample(T.bc_kind, T.bc_n, s_keys + T.o_bc_t, s_keys + T.o_bc_v, age_percent, bc);
    fw_gradient_sample(T.em_kind, T.em_n, s_keys + T.o_em_t, s_keys + T.o_em_v, age_percent, em);
    const uint32_t b16 = (o - W.first) * 16u;  // < 16 KiB + a tile: the window starts at the tile's first output slot
    fw_st4w<NT == 2>(W.q0, b16, make_float4(px, py, pz, age_new));
    fw_st4w<NT == 2>(W.q1, b16, make_float4(vx, vy, vz, q1.w));
    if (INPLACE) {
        const uint32_t d2 = (__float_as_uint(nr.x) ^ __float_as_uint(q2.x)) | (__float_as_uint(nr.y) ^ __float_as_uint(q2.y)) |
                            (__float_as_uint(nr.z) ^ __float_as_uint(q2.z)) | (__float_as_uint(nr.w) ^ __float_as_uint(q2.w));
        const uint32_t d3 = (__float_as_uint(wx) ^ __float_as_uint(q3.x)) | (__float_as_uint(wy) ^ __float_as_uint(q3.y)) |
                            (__float_as_uint(wz) ^ __float_as_uint(q3.z));
        if (W.wr2 && __any(full || d2 != 0u)) fw_st4w<NT == 2>(W.q2, b16, make_float4(nr.x, nr.y, nr.z, nr.w));  // wave-uniform branches
        if (W.wr3 && __any(full || d3 != 0u)) fw_st4w<NT == 2>(W.q3, b16, make_float4(wx, wy, wz, lifetime));
        if ((WM >= 0 ? (WM & 1) != 0 : W.wr5) || full) fw_st4w<NT != 0>(W.q5, b16, make_float4(bc[0], bc[1], bc[2], bc[3]));
        if ((WM >= 0 ? (WM & 2) != 0 : W.wr6) || full) fw_st4w<NT != 0>(W.q6, b16, make_float4(em[0], em[1], em[2], em[3]));
        if ((WM >= 0 ? (WM & 4) != 0 : (T.sc_kind != 0 && W.wr4)) || full) fw_st1w<NT != 0>(W.s4, (o - W.first) * 4u, scale);
    } else {
        if (W.wr2) fw_st4w<NT == 2>(W.q2, b16, make_float4(nr.x, nr.y, nr.z, nr.w));
        if (W.wr3) fw_st4w<NT == 2>(W.q3, b16, make_float4(wx, wy, wz, lifetime));
        else fw_st1w<NT == 2>(W.lf, (o - W.first) * 4u, lifetime);
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

// every workgroup (active or not) clears its own slot of the buffer the frame after the next will accumulate into
__device__ __forceinline__ void fw_fc_housekeeping(const FwUpdateArgs &a) {
    if (threadIdx.x == 0 && a.fc_zero) {
        a.fc_zero[blockIdx.x] = 0ull;
        if ((blockIdx.x & 63u) == 0u) a.fc_zero[a.fc_s2 + (blockIdx.x >> 6) * FW_FC_S2_STRIDE] = 0ull;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0 && a.fc_out) a.fc_out[a.fc_tag] = (fw_u64)a.epoch;
}


// R = rounds per tile; the workgroup has FW_TILE / R threads, so a tile is always FW_TILE particles.
// R = 4 (256 threads) is the measured optimum on MI355X.
//
// Register diet.  A frame at 1M particles is ~1k tiles; what bounds the kernel there is not bandwidth but
// how many tiles are resident at once (tile lifetime x number of "rounds" of workgroups).  Holding a tile's
// 64 B/particle of input in VGPRs for all four rounds costs 165 VGPRs = 3 workgroups per CU = 768 slots, i.e.
// two rounds.  So the survival planes (Q0: position+age, Q3: angular velocity+lifetime) are loaded once,
// used for the survivor count, and parked in LDS (32 KiB per workgroup); the rounds then run as a rolled
// loop that re-derives each lane's flags from LDS and prefetches Q1/Q2 one round ahead.  That is ~90 VGPRs:
// four workgroups per CU (LDS-limited), 1024 slots, one round.
//
// Where a tile's output offset (exclusive survivor prefix) comes from:
//   * FORECAST (a.fc_in != null): the previous frame's kernel already evaluated, for every survivor it
//     stored, whether it survives one more step of the same dt, and left per-tile sums in the forecast
//     table.  The host enables this only when dt repeats bit-for-bit and nothing touched the state in
//     between, so the sums are exact: the tile adds up its predecessors' entries (plain L2 reads of data
//     finished a kernel ago) and never waits for a co-resident workgroup.  Only tiles that hold freshly
//     spawned particles look back -- among themselves -- for the survivors of the new particles.
//   * otherwise: single-pass decoupled look-back over all earlier tiles of the segment.
template <bool FUSED, int SPAWN, int R, bool INST, bool SUMS>
__global__ __launch_bounds__(FW_TILE / R) void fw_k_update(FwGlobals g, FwUpdateArgs a, FwInlineOps inl) {
    constexpr int BLK = FW_TILE / R;
    constexpr int NW = BLK / 64;
    constexpr int LBW = 4;  // status words per lane per look-back step
    __shared__ __attribute__((aligned(16))) float4 s_q0[FW_TILE];  // Q0 / Q3 of the tile (virtual particles included)
    __shared__ __attribute__((aligned(16))) float4 s_q3[FW_TILE];
    __shared__ __attribute__((aligned(16))) float s_keys[FW_KEYS_MAX];
    __shared__ uint32_t s_wcnt[R + 1][NW];  // row R: the extra round of a tile that carries its segment's few new particles
    __shared__ uint32_t s_lb[2 * LBW * NW];
    __shared__ uint32_t s_part[4][NW];  // per-wave partials: forecast prefix, new survivors, next-frame sums A / B

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long ts0 = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // workgroup -> (segment, tile in segment): one table read instead of a dependent binary search
    uint32_t seg, first, seg_tiles, type_idx;
    if (a.n_seg == 1u) {  // a lone segment: everything is in the kernel arguments, no table read on the critical path
        seg = 0, first = 0, seg_tiles = a.total_tiles, type_idx = a.seg0_type;
    } else if (a.tile_desc) {
        const uint4 d = a.tile_desc[blockIdx.x];
        seg = d.x, first = d.y, seg_tiles = d.z, type_idx = d.w;
    } else {
        seg = fw_upper_slot(a.seg_tile_first, a.n_seg, blockIdx.x);
        first = a.seg_tile_first[seg];
        seg_tiles = a.seg_tile_first[seg + 1] - first;
        type_idx = g.segs[seg].type_idx;
    }
    // (a type that cannot turn: the rotation plane is not read -- every lane asks for the same slot instead, one line per
    // wave, and fw_integrate_store takes FwType::const_rot; an unconditional load keeps the prefetch structure)
    const uint32_t m2 = (type_idx & FW_TYPE_IDX_NOSPIN) ? 0u : 0xFFFFFFFFu;
    type_idx &= ~FW_TYPE_IDX_NOSPIN;
    uint32_t tis = blockIdx.x - first;
    const bool use_fc = FUSED && a.fc_in != nullptr;
    constexpr bool fc_small = !SUMS;  // one plain entry per tile (every segment small) instead of atomic sums
    // (fw_fc_housekeeping runs at the end: a store this early would sit in front of every load in the vmcnt queue)
    uint4 fce[FW_FCE_U];
    if (use_fc && fc_small) fw_fce_request<BLK>(a.fce_in, first, seg_tiles, fce);
    const uint32_t p = a.parity;
    const uint32_t sidx = p * g.max_seg + seg, oidx = (p ^ 1u) * g.max_seg + seg;
    // particles that existed before this frame; [n_before, n_in) were materialised this frame (fw_k_spawn / fw_k_nest) and have
    // never been updated: their slots hold the spawn-time scale and colours, whatever FW_TYPE_DERIVED says about the planes
    const uint32_t n_before = g.count[sidx];
    const uint32_t n_in = n_before + g.spawned[sidx] + g.appended[sidx];
    uint32_t o0 = 0, o1 = 0, n_spawn = 0;  // this segment's ops (contiguous: ops are sorted by segment)
    if (SPAWN == FW_SPAWN_INLINE) {
        for (uint32_t i = 0; i < a.n_ops; i++) {
            if (inl.ops[i].seg == seg) {
                if (o1 == 0) o0 = i;
                o1 = i + 1;
                n_spawn += inl.ops[i].n;
            }
        }
    } else if (SPAWN == FW_SPAWN_TABLE) {
        o0 = a.seg_op_first[seg], o1 = a.seg_op_first[seg + 1];
        for (uint32_t i = o0; i < o1; i++) n_spawn += a.ops[i].n;
    }
    // virtual spawns beyond the segment's capacity are dropped (and reported): nothing may be read or written past it
    const uint32_t seg_cap = g.segs[seg].capacity;
    if (SPAWN != FW_SPAWN_NONE) {
        const uint32_t spawn_room = seg_cap - min(n_in, seg_cap);
        if (n_spawn > spawn_room) {  // (reported here: nothing about it has to stay live through the kernel)
            n_spawn = spawn_room;
            if (blockIdx.x == first && threadIdx.x == 0) atomicOr(g.err, FW_ERR_CAPACITY);
        }
    }
    const uint32_t n_tot = n_in + n_spawn;
    // Tiling of the index space [0, n_tot): the live particles [0, n_in) in tiles of FW_TILE, then the new
    // ones [n_in, n_tot) in SMALL tiles (1, 2 or 4 rounds, the smallest that keeps all active tiles of the frame
    // resident at once; chosen by the host).  A new particle costs ~1-3k VALU instructions
    // before its tile can even count survivors; small tiles spread that over 4x more workgroups, and those
    // workgroups are dispatched FIRST (they are the last tiles) so the compute overlaps everybody else's
    // streaming instead of forming the kernel's tail.  At most FW_VFRONT tiles are front-loaded: they wait for
    // all earlier tiles while holding a slot, and the rest of the grid, still dispatched in tile order,
    // always makes progress.
    const uint32_t t_spawn = (n_in + FW_TILE - 1u) / FW_TILE;  // number of live tiles = first new-particle tile
    // new-particle tile size: one round, or two when that is what keeps every active tile resident at once
    // (a.resident_slots workgroups).  The host can only bound the live count, so a lone segment decides from
    // the exact device count; with several segments the host's choice (from its bounds) is used.
    const uint32_t vt_rounds = a.vt_rounds;  // the host's choice (from its bounds), for every segment
    const uint32_t vtile = vt_rounds * BLK;
    // a handful of new particles ride along in the segment's last live tile when it has room (same rule as in
    // fw_k_update_stream: the host sizes the grid counting on it)
    const bool merge_new = SPAWN != FW_SPAWN_NONE && n_spawn != 0u && n_spawn <= BLK && t_spawn != 0u &&
                           n_tot <= t_spawn * FW_TILE;
    const uint32_t n_vt = merge_new ? 0u : (n_spawn + vtile - 1u) / vtile;
    const uint32_t n_act = t_spawn + n_vt;                      // active tiles of this segment
    if (SPAWN != FW_SPAWN_NONE && n_vt != 0 && n_vt <= FW_VFRONT && t_spawn != 0 && tis < n_act)
        tis = tis < n_vt ? t_spawn + tis : tis - n_vt;
    const uint32_t tile = first + tis;
    const bool has_new = tis >= t_spawn;  // block-uniform: a tile is either all live or all new (tail_new: see above)
    const bool tail_new = merge_new && tis + 1u == t_spawn;
    const uint32_t base = has_new ? n_in + (tis - t_spawn) * vtile : tis * FW_TILE;
    const uint32_t lim = has_new ? min(base + vtile, n_tot) : min(base + FW_TILE, n_in);
    fw_u64 *fc_out = FUSED ? a.fc_out : nullptr;

    if (n_tot == 0 || tis >= n_act) {
        if (tid == 0) {
            if (fc_out && fc_small) a.fce_out[tile] = make_uint4(0u, 0u, 0u, a.epoch);  // contributes nothing next frame
            if (n_tot == 0 && tis == 0) {  // empty segment: its first tile still owns the bookkeeping
                g.count[oidx] = 0;
                g.spawned[oidx] = 0;
                g.appended[oidx] = 0;
                g.ndestroyed[seg] = 0;
                if (a.host_counts) a.host_counts[seg] = (unsigned long long)a.epoch << 32;
            }
            if (blockIdx.x == 0 && a.live_next) *a.live_next = 0ull;
            if (blockIdx.x == 0 && a.done_tag) *a.done_tag = a.done_value;
        }
        if (SUMS) fw_fc_housekeeping(a);
        return;
    }
    const bool is_last = tis + 1u == n_act;
    if (tis == 0 && tid == 0 && n_act > seg_tiles) {
        atomicOr(g.err, FW_ERR_CAPACITY);
        g.err[1] = seg, g.err[2] = n_tot, g.err[3] = seg_tiles, g.err[4] = n_in;  // diagnostics
    }
    if (blockIdx.x == 0 && tid == 0 && a.live_next) *a.live_next = 0ull;
    if (blockIdx.x == 0 && tid == 0 && a.done_tag) *a.done_tag = a.done_value;

    // field-wise reads (block-uniform -> scalar loads)
    const FwSeg *Sp = &g.segs[seg];
    const uint32_t C = seg_cap;
    const uint32_t n_lplanes = Sp->n_lplanes;
    char *ib = Sp->buf[p];  // written only at the slots of this frame's new particles
    char *ob = Sp->buf[p ^ 1u];
    char *destroyed = Sp->destroyed;
    char *inst = INST ? Sp->inst : nullptr;
    const uint32_t inst_cap = INST ? Sp->inst_cap : 0u;

    const unsigned long long tsA = FW_DBG(a.dbg, 8u) ? (__builtin_amdgcn_s_memrealtime() + (C & 0u)) : 0ull;
    // ---- phase 1: the planes that decide survival: Q0 (age in .w) and Q3 (lifetime in .w); all R loads of
    // both planes are in flight together, then parked in LDS
    {
        float4 t0[R], t3[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            // unconditional, clamped into the tile (a predicated load would be waited for in its own basic block)
            const uint32_t idx = has_new ? 0u : min(base + r * BLK + tid, lim - 1u);
            t0[r] = fw_ld4(ib + FW_OFF_Q0(C), idx);
            t3[r] = fw_ld4(ib + FW_OFF_Q3(C), idx & m2);
            // a type that cannot turn: Q3 is not kept, the lifetime comes from its own plane (both loads unconditional: the
            // one that is not needed asks for one and the same slot of a plane that exists)
            const float lf = fw_ld1((m2 ? ib + FW_OFF_Q0(C) : ib + FW_OFF_L(C, n_lplanes)), m2 ? 0u : idx);
            if (!m2) t3[r] = make_float4(0.0f, 0.0f, 0.0f, lf);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t idx = base + r * BLK + tid;
            if (!has_new && idx < lim) s_q0[r * BLK + tid] = t0[r], s_q3[r * BLK + tid] = t3[r];
        }
    }
    const unsigned long long tsB = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // first round's Q1 / Q2 go out now; later rounds are prefetched one round ahead
    float4 q1c, q2c;
    {
        const uint32_t i0 = has_new ? 0u : min(base + tid, lim - 1u);
        q1c = fw_ld4(ib + FW_OFF_Q1(C), i0);
        q2c = fw_ld4(ib + FW_OFF_Q2(C), i0 & m2);
    }

    // per-type constants (scalar loads) and curve / gradient keys (staged in LDS) arrive under the loads
    const FwType T = g.types[type_idx];
    for (uint32_t i = tid; i < T.keys_len; i += BLK) s_keys[i] = g.keys[T.keys_off + i];

    // ---- forecast prefix: survivors sitting in the input tiles before this one (all live tiles for a new-particle tile)
    uint32_t fc_part = 0;
    bool fc_bad = false;
    if (use_fc && fc_small) {
        fc_part = fw_fce_prefix_part<BLK>(fce, seg_tiles, tis, a.epoch, &fc_bad);
    } else if (use_fc) {
        fc_part = fw_fc_prefix_part<BLK>(a.fc_in, a.fc_s2, first, first + min(tis, t_spawn), seg_tiles);
        fc_bad = tid == 0 && fw_ld2u(a.fc_in, a.fc_tag).x != a.epoch - 1u;
    }

    const unsigned long long tsC = FW_DBG(a.dbg, 8u) ? (__builtin_amdgcn_s_memrealtime() + (fc_part & 0u)) : 0ull;
    if (SPAWN != FW_SPAWN_NONE && has_new) {
        // New particles (src/core.rs:437-469), generated from the counter RNG straight into LDS: Q0/Q3 where a
        // loaded tile parks them, Q1/Q2 in the upper half of the same planes (a new-particle tile is at most
        // FW_TILE / 2 particles).  From here on they are ordinary inputs: spawn runs before update in the same
        // frame (src/plugin.rs:46-60).  One rolled instance of the (large) spawn code, in its own loop so its
        // registers do not add to the round loop's.
#pragma unroll 1
        for (int r = 0; r < R / 2; r++) {
            const uint32_t idx = base + r * BLK + tid;
            if (idx < lim) {
                const uint32_t k = idx - n_in;
                uint32_t oi = o0;
                for (uint32_t i = o0; i < o1; i++)
                    if (k >= FW_OP(i).rel_base && k - FW_OP(i).rel_base < FW_OP(i).n) oi = i;
                const FwOp &op = FW_OP(oi);
                const FwSpawnOut so = fw_spawn_one(
                    g.emits[op.emit], g.seed, op.serial_base + (k - op.rel_base),
                    fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                    fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                    fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
                s_q0[r * BLK + tid] = so.q0, s_q3[r * BLK + tid] = so.q3;
                s_q0[FW_TILE / 2 + r * BLK + tid] = so.q1, s_q3[FW_TILE / 2 + r * BLK + tid] = so.q2;
            }
        }
    }

    // survivor count of the tile: each lane re-reads what it parked (same lane, no barrier needed yet)
    uint32_t new_alive = 0;  // survivors among this tile's new particles (wave-uniform partial)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t idx = base + r * BLK + tid;
        float an;
        const bool al = idx < lim && fw_survives(s_q0[r * BLK + tid].w, a.dt, s_q3[r * BLK + tid].w, &an);
        const unsigned long long m = __ballot(al);
        if (lane == 0) s_wcnt[r][wave] = (uint32_t)__popcll(m);
        if (has_new) new_alive += (uint32_t)__popcll(m);
    }
    {  // row R: survivors among the new particles this (live) tile carries: age 0, lifetime = RNG block 2 word 0
        bool al = false;
        if (SPAWN != FW_SPAWN_NONE && tail_new && tid < n_spawn) {
            const uint32_t k = tid;
            uint32_t oi = o0;
            for (uint32_t i = o0; i < o1; i++)
                if (k >= FW_OP(i).rel_base && k - FW_OP(i).rel_base < FW_OP(i).n) oi = i;
            const FwOp &op = FW_OP(oi);
            const FwEmit &e = g.emits[op.emit];
            const unsigned long long serial = op.serial_base + (k - op.rel_base);
            const fw_u4 o = fw_philox4x32_10(fw_u4{(uint32_t)serial, (uint32_t)(serial >> 32), e.emission_index, 2u}, g.seed,
                                             e.uid);
            float an;
            al = fw_survives(0.0f, a.dt, fw_unit_f32(o.x) * (e.life_max - e.life_min) + e.life_min, &an);
        }
        const unsigned long long m = __ballot(al);
        if (lane == 0) s_wcnt[R][wave] = (uint32_t)__popcll(m);
    }
    if (use_fc) {
        fc_part = fw_wave_sum(fc_part);
        if (lane == 0) s_part[0][wave] = fc_part, s_part[1][wave] = new_alive;
        if (__any(fc_bad) && lane == 0) fw_raise(g, 1u, 0xFFFFFFFFu, tile);
    }
    __syncthreads();
    const unsigned long long ts1 = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r <= R; r++) {
#pragma unroll
        for (int w = 0; w < NW; w++) cnt += s_wcnt[r][w];
    }
    uint32_t fc_excl = 0, new_cnt = 0;
    if (use_fc) {
#pragma unroll
        for (int w = 0; w < NW; w++) fc_excl += s_part[0][w], new_cnt += s_part[1][w];
    }

    // ---- phase 2: exclusive prefix of survivors over the earlier tiles of this segment
    uint32_t excl = 0;
    if (FUSED) {
        // what this tile publishes / where its look-back starts
        const bool lb_needed = use_fc ? (has_new && tis > t_spawn) : tis > 0;
        const bool lb_publish = use_fc ? has_new : true;
        const uint32_t lb_lo = use_fc ? first + t_spawn : first;
        const uint32_t lb_val = use_fc ? new_cnt : cnt;
        if (lb_publish && lb_needed && tid == 0)
            __hip_atomic_store(&g.tile_status[tile], fw_pack_status(a.epoch, FW_ST_AGG, lb_val), RLX, AGENT);
        uint32_t lb_excl = 0;
        if (lb_needed && !FW_DBG(a.dbg, 1u)) {
            bool timed_out = false;
            lb_excl = fw_lookback<BLK, NW, LBW>(g.tile_status, lb_lo, tile, a.epoch, a.spin_limit, s_lb, &timed_out);
            if (timed_out) {
                // Fallback (never taken when workgroups are dispatched in order): recount the survivors of
                // the earlier particles of this segment (forecast mode: of the earlier NEW particles only).
                if (tid == 0) atomicOr(g.err, FW_ERR_LOOKBACK_TIMEOUT);
                uint32_t c = 0;
                for (uint32_t i = (use_fc ? n_in : 0u) + tid; i < base; i += BLK) {  // base is a particle index
                    float an, ag = 0.0f, lf;
                    if (i < n_in) {
                        ag = fw_ld4(ib + FW_OFF_Q0(C), i).w, lf = fw_load_q3(ib, C, n_lplanes, i, m2 == 0u).w;
                    } else {  // a spawned particle: only its lifetime draw matters (RNG block 2, word 0)
                        const uint32_t k = i - n_in;
                        uint32_t oi = o0;
                        for (uint32_t j = o0; j < o1; j++)
                            if (k >= FW_OP(j).rel_base && k - FW_OP(j).rel_base < FW_OP(j).n) oi = j;
                        const FwOp &op = FW_OP(oi);
                        const FwEmit &e = g.emits[op.emit];
                        const unsigned long long serial = op.serial_base + (k - op.rel_base);
                        const fw_u4 o = fw_philox4x32_10(
                            fw_u4{(uint32_t)serial, (uint32_t)(serial >> 32), e.emission_index, 2u}, g.seed, e.uid);
                        lf = fw_unit_f32(o.x) * (e.life_max - e.life_min) + e.life_min;
                    }
                    c += fw_survives(ag, a.dt, lf, &an) ? 1u : 0u;
                }
                c = fw_wave_sum(c);
                __syncthreads();
                if (lane == 0) s_lb[wave] = c;
                __syncthreads();
                lb_excl = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) lb_excl += s_lb[w];
            }
        }
        if (lb_publish && tid == 0)
            __hip_atomic_store(&g.tile_status[tile], fw_pack_status(a.epoch, FW_ST_INCL, lb_excl + lb_val), RLX, AGENT);
        excl = fc_excl + lb_excl;
    } else {
        excl = g.tile_off[tile];
    }

    const unsigned long long ts2 = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // ---- phase 3: round loop -- integrate survivors, store them at their compacted slot
    const bool want_destroyed = T.report_destroyed && destroyed != nullptr;
    excl = __builtin_amdgcn_readfirstlane(excl);  // workgroup-uniform: keep it on the scalar unit
    const FwOutWin W = fw_out_window(ob, C, excl, T, a.force_colors, n_lplanes);
    const uint32_t fcA = excl / FW_TILE, fc_bnd = (fcA + 1u) * FW_TILE;  // output tiles this workgroup feeds
    uint32_t fa = 0, fb = 0;
    float box[6] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f, FW_F32_MIN, FW_F32_MIN, FW_F32_MIN};
    const bool box_on = a.boxes != 0u;  // workgroup-uniform
    uint32_t run = excl;  // output slot of the first survivor of (round r, wave 0)
    const int n_rounds = (int)((lim - base + BLK - 1u) / BLK);  // a partial tile runs only the rounds that hold particles
#pragma unroll 1
    for (int r = 0; r < n_rounds; r++) {
        const uint32_t idx = base + r * BLK + tid;
        // prefetch the next round's Q1 / Q2 (new particles were materialised above, so idx < n_tot is enough)
        const uint32_t in_ = has_new ? 0u : min(idx + BLK, lim - 1u);  // clamped, unconditional
        const float4 q1n = fw_ld4(ib + FW_OFF_Q1(C), in_);
        const float4 q2n = fw_ld4(ib + FW_OFF_Q2(C), in_ & m2);
        const bool valid = idx < lim, loaded = !has_new;
        const bool updated_before = loaded && idx < n_before;  // (destroyed records: evaluate / read the planes vs spawn-time values)
        const float4 q0 = s_q0[r * BLK + tid], q3 = s_q3[r * BLK + tid];
        if (SPAWN != FW_SPAWN_NONE && has_new && valid)
            q1c = s_q0[FW_TILE / 2 + r * BLK + tid], q2c = s_q3[FW_TILE / 2 + r * BLK + tid];
        float age_new;
        const bool alive = valid && fw_survives(q0.w, a.dt, q3.w, &age_new);
        const unsigned long long m = __ballot(alive);
        uint32_t wbase = run;  // + survivors of the earlier waves of this round
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = s_wcnt[r][w];
            if ((uint32_t)w < wave) wbase += c;
            run += c;
        }
        const uint32_t o = wbase + fw_lane_prefix(m);
        if (fc_out) {  // will it survive one more step of the same dt?  (same expression as fw_survives)
            float an2;
            const bool nx = alive && fw_survives(age_new, a.dt, q3.w, &an2);
            fa += (uint32_t)__popcll(__ballot(nx && o < fc_bnd));
            fb += (uint32_t)__popcll(__ballot(nx && o >= fc_bnd));
        }
        if (alive && FW_DBG(a.dbg, 2u)) {  // profiling only: stream without arithmetic
            const uint32_t b16 = (o - W.first) * 16u;
            fw_st4w(W.q0, b16, make_float4(q0.x, q0.y, q0.z, age_new)), fw_st4w(W.q1, b16, q1c);
            fw_st4w(W.q2, b16, q2c), fw_st4w(W.q3, b16, q3);
            fw_st4w(W.q5, b16, q0), fw_st4w(W.q6, b16, q1c);
            fw_st1w(W.s4, (o - W.first) * 4u, q1c.w);
        } else if (alive) {
            float4 rec[4];
            fw_integrate_store(T, s_keys, a.dt, q0, q1c, q2c, q3, age_new, W, o, INST ? rec : nullptr, nullptr, nullptr, box, box_on);
            if (INST && inst != nullptr && o < inst_cap) {  // this schedule is the rare one: plain per-lane records
                fw_st4(inst, o * 4u + 0u, rec[0]), fw_st4(inst, o * 4u + 1u, rec[1]);
                fw_st4(inst, o * 4u + 2u, rec[2]), fw_st4(inst, o * 4u + 3u, rec[3]);
            }
            for (uint32_t k = 0; k < n_lplanes; k++)  // new particles: vec![f32::MIN; n] (core.rs:467)
                fw_st1(ob + FW_OFF_L(C, k), o, loaded ? fw_ld1(ib + FW_OFF_L(C, k), idx) : FW_F32_MIN);
        } else if (valid && want_destroyed) {
            fw_store_destroyed(destroyed, ib, C, idx, updated_before, T, s_keys, q0, q1c, q2c, q3, age_new, idx - o);
        }
        q1c = q1n, q2c = q2n;
    }
    if (SPAWN != FW_SPAWN_NONE && tail_new) {
        // ---- the extra round: this segment's few new particles, spawned (src/core.rs:437-469) and updated right
        // behind the tile's live survivors
        const uint32_t idx = n_in + tid;
        const bool valid = tid < n_spawn;
        FwSpawnOut so;
        so.q0 = so.q1 = so.q2 = so.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            const uint32_t k = tid;
            uint32_t oi = o0;
            for (uint32_t i = o0; i < o1; i++)
                if (k >= FW_OP(i).rel_base && k - FW_OP(i).rel_base < FW_OP(i).n) oi = i;
            const FwOp &op = FW_OP(oi);
            so = fw_spawn_one(g.emits[op.emit], g.seed, op.serial_base + (k - op.rel_base),
                              fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                              fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                              fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
        }
        float age_new;
        const bool alive = valid && fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);
        const unsigned long long m = __ballot(alive);
        uint32_t wbase = run;
#pragma unroll
        for (int w = 0; w < NW; w++)
            if ((uint32_t)w < wave) wbase += s_wcnt[R][w];
        const uint32_t o = wbase + fw_lane_prefix(m);
        if (fc_out) {
            float an2;
            const bool nx = alive && fw_survives(age_new, a.dt, so.q3.w, &an2);
            fa += (uint32_t)__popcll(__ballot(nx && o < fc_bnd));
            fb += (uint32_t)__popcll(__ballot(nx && o >= fc_bnd));
        }
        if (alive) {
            float4 rec[4];
            fw_integrate_store(T, s_keys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, W, o, INST ? rec : nullptr, nullptr, nullptr,
                               box, box_on);
            if (INST && inst != nullptr && o < inst_cap) {
                fw_st4(inst, o * 4u + 0u, rec[0]), fw_st4(inst, o * 4u + 1u, rec[1]);
                fw_st4(inst, o * 4u + 2u, rec[2]), fw_st4(inst, o * 4u + 3u, rec[3]);
            }
            for (uint32_t k = 0; k < n_lplanes; k++) fw_st1(ob + FW_OFF_L(C, k), o, FW_F32_MIN);  // core.rs:467
        } else if (valid && want_destroyed) {
            fw_store_destroyed(destroyed, ib, C, idx, false, T, s_keys, so.q0, so.q1, so.q2, so.q3, age_new, idx - o);
        }
    }
    if (fc_out) {
        if (lane == 0) s_part[2][wave] = fa, s_part[3][wave] = fb;
        __syncthreads();
        if (tid == 0) {
            uint32_t sa = 0, sb = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) sa += s_part[2][w], sb += s_part[3][w];
            if (fc_small) a.fce_out[tile] = make_uint4(sa, sb, fcA, a.epoch);
            else fw_fc_add(fc_out, a.fc_s2, first + fcA, sa, sb, seg_tiles);
        }
    }
    if (a.boxes) {
        __syncthreads();  // (s_lb doubles as the exchange area of the box fold)
        fw_tile_box_flush<NW>(g.tile_box, tile, a.epoch, box, reinterpret_cast<float (*)[6]>(s_lb));
    }

    if (FW_DBG(a.dbg, 8u) && g.dbg_ts && tid == 0) {
        unsigned long long *d = g.dbg_ts + 32768 + ((size_t)(a.epoch & 1u) * gridDim.x + tile) * 8;  // two launches kept
        {  // ring of the last 256 launches: earliest start / latest end, spread over 64 words each to keep the atomics apart
            unsigned long long *rg = g.dbg_ts + (size_t)(a.epoch & 255u) * 128u;
            atomicMax(&rg[blockIdx.x & 63u], ~ts0);  // max of the complement = min (slots are recycled with 0)
            atomicMax(&rg[64u + (blockIdx.x & 63u)], __builtin_amdgcn_s_memrealtime());
            if (blockIdx.x < 128u) g.dbg_ts[(size_t)((a.epoch + 128u) & 255u) * 128u + blockIdx.x] = 0ull;
        }
        const unsigned long long tsE = __builtin_amdgcn_s_memrealtime();
        d[0] = ts0, d[1] = ts1, d[2] = ts2, d[3] = tsE;
        d[4] = tsA, d[5] = tsB, d[6] = tsC, d[7] = 0;
    }
    if (SUMS) fw_fc_housekeeping(a);
    if (is_last && tid == 0) {
        const uint32_t nc = excl + cnt;
        g.count[oidx] = nc;
        g.spawned[oidx] = 0;
        g.appended[oidx] = 0;
        g.ndestroyed[seg] = n_tot - nc;
        if (a.host_counts) a.host_counts[seg] = ((unsigned long long)a.epoch << 32) | nc;  // one 8-byte store: tag + count
        if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, (unsigned long long)n_tot);  // (FW_DEBUG 128: profiling, no statistics)
    }
}

// ---------------------------------------------------------------------------------
// fw_k_update_stream: the update kernel of FORECAST frames (dt repeated, state untouched: the steady state).
//
// Same tiling, tables and results as fw_k_update, different schedule.  With the forecast a live tile knows its
// output offset before it has seen a single particle, so nothing has to be counted ahead of time and nothing
// has to be parked: each round loads its four input planes (prefetched one round ahead), ranks its survivors
// (ballot + one LDS exchange per round), integrates and stores.  Loads of round r+1 and stores of round r are
// in flight together in every workgroup, which is what the memory system wants: the microbenchmark of this
// exact schedule (tools/membw.hip, "stream twin") moves the 164 B/particle at 8.2 TB/s out of the Infinity
// Cache at 1M particles and 5.6-5.8 TB/s from HBM at 4M-16M, against 5.8 / 4.8 TB/s for load-count-park-store.
// A new-particle tile counts its survivors from the lifetime draws alone (one Philox block per particle),
// looks back among the new-particle tiles only, then generates each particle right before integrating it.
// ---------------------------------------------------------------------------------
struct FwRoundOut {
    uint32_t fa, fb;
};

// everything a round does once a lane has its particle (q0..q3) and its output slot `o`
__device__ __forceinline__ void fw_round_finish(const FwType &T, const float *s_keys, float dt, uint32_t dbg, float4 q0,
                                                float4 q1, float4 q2, float4 q3, bool valid, bool alive, bool loaded,
                                                float age_new, uint32_t idx, uint32_t o, const char *ib, char *ob,
                                                const FwOutWin &W, char *destroyed, bool want_destroyed, uint32_t C,
                                                uint32_t n_lplanes, bool forecast, uint32_t fc_bnd, FwRoundOut &acc,
                                                float4 *rec = nullptr, float *box = nullptr, bool box_on = false,
                                                bool fresh = false) {  // fresh: materialised this frame, never updated
    if (forecast) {  // will it survive one more step of the same dt?  (same expression as fw_survives)
        float an2;
        const bool nx = alive && fw_survives(age_new, dt, q3.w, &an2);
        acc.fa += (uint32_t)__popcll(__ballot(nx && o < fc_bnd));
        acc.fb += (uint32_t)__popcll(__ballot(nx && o >= fc_bnd));
    }
    if (alive && FW_DBG(dbg, 2u)) {  // profiling only: stream without arithmetic
        const uint32_t b16 = (o - W.first) * 16u;
        fw_st4w(W.q0, b16, make_float4(q0.x, q0.y, q0.z, age_new)), fw_st4w(W.q1, b16, q1);
        fw_st4w(W.q2, b16, q2), fw_st4w(W.q3, b16, q3);
        fw_st4w(W.q5, b16, q0), fw_st4w(W.q6, b16, q1);
        fw_st1w(W.s4, (o - W.first) * 4u, q1.w);
    } else if (alive) {
        fw_integrate_store(T, s_keys, dt, q0, q1, q2, q3, age_new, W, o, rec, nullptr, nullptr, box, box_on);
        for (uint32_t k = 0; k < n_lplanes; k++)  // new particles: vec![f32::MIN; n] (core.rs:467)
            fw_st1(ob + FW_OFF_L(C, k), o, loaded ? fw_ld1(ib + FW_OFF_L(C, k), idx) : FW_F32_MIN);
    } else if (valid && want_destroyed) {
        // (a never-updated particle carries its spawn-time scale and colours: evaluated, which is also what its slot holds)
        fw_store_destroyed(destroyed, ib, C, idx, loaded && !fresh, T, s_keys, q0, q1, q2, q3, age_new, idx - o);
    }
}

// LONE: the context holds a single segment (the 1M-particle headline case).  Everything a tile needs to ADDRESS its
// input then comes with the kernel arguments (buffers, capacity: FwUpdateArgs::seg0_*) and the tile -> particle-range
// mapping of a live tile does not depend on the live count (new-particle tiles are dispatched first and their number
// follows from the spawn ops in the arguments), so round 0 of the four input planes is requested SPECULATIVELY right
// after the arguments arrive -- in parallel with the counters, the forecast entries and the per-type constants
// instead of one dependent memory round trip (~1 us at launch, when every workgroup asks at once) after them.  Lanes
// past the live count read stale slots of the buffer and are masked; a tile whose role turns out different (a
// new-particle tile of materialised children, a clamped spawn) reloads.
template <int SPAWN, bool INST, bool SUMS, bool LONE>
__global__ __launch_bounds__(FW_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void fw_k_update_stream(FwGlobals g, FwUpdateArgs a, FwInlineOps inl) {
    static_assert(!LONE || SPAWN != FW_SPAWN_TABLE, "a lone segment with a table of ops takes the general kernel");
    constexpr int BLK = FW_BLOCK;
    constexpr int NW = BLK / 64;
    constexpr int LBW = 4;
    __shared__ __attribute__((aligned(16))) float s_keys[FW_KEYS_MAX];
    __shared__ __attribute__((aligned(16))) float4 s_inst[INST ? NW * 256 : 1];  // per wave: 64 records of 4 float4
    __shared__ uint32_t s_c[2][NW];     // survivors per wave of the current round (double-buffered)
    __shared__ uint32_t s_lb[2 * LBW * NW];
    __shared__ uint32_t s_part[4][NW];  // per-wave partials: forecast prefix, new survivors, next-frame sums A / B

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long ts0 = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t seg, first, seg_tiles, type_idx, keys_off, keys_len;
    if (LONE) {
        seg = 0, first = 0, seg_tiles = a.total_tiles, type_idx = a.seg0_type;
        keys_off = a.seg0_keys_off, keys_len = a.seg0_keys_len;
    } else {
        const uint4 d = a.tile_desc[blockIdx.x];
        seg = d.x, first = d.y, seg_tiles = d.z, type_idx = d.w;
        const uint2 kd = a.tile_keys[seg];
        keys_off = kd.x, keys_len = kd.y;
    }
    const uint32_t m2 = (type_idx & FW_TYPE_IDX_NOSPIN) ? 0u : 0xFFFFFFFFu;  // FW_TYPE_NOSPIN: no rotation-plane traffic
    type_idx &= ~FW_TYPE_IDX_NOSPIN;
    // curve / gradient keys: requested first (into a register; they are moved to LDS after the other requests
    // are out, so nothing waits for them here)
    const float key0 = tid < keys_len ? g.keys[keys_off + tid] : 0.0f;
    uint32_t tis = blockIdx.x - first;
    // forecast entries of a small segment: requested before anything else (they depend on the descriptor only)
    constexpr bool fc_small = !SUMS;  // one plain entry per tile (every segment small) instead of atomic sums
    uint4 fce[FW_FCE_U];
    if (fc_small) fw_fce_request<BLK>(a.fce_in, first, seg_tiles, fce);
    // (fw_fc_housekeeping runs at the end: a store this early would sit in front of every load in the vmcnt queue)
    const uint32_t p = a.parity;
    // the segment record: kernel arguments for a lone segment, memory otherwise
    const FwSeg *Sp = &g.segs[seg];
    const uint32_t seg_cap = LONE ? a.seg0_capacity : Sp->capacity;
    const uint32_t C = seg_cap;
    const uint32_t n_lplanes = LONE ? a.seg0_n_lplanes : Sp->n_lplanes;
    const char *ib = LONE ? a.seg0_ib : Sp->buf[p];
    char *ob = LONE ? a.seg0_ob : Sp->buf[p ^ 1u];
    char *destroyed = LONE ? a.seg0_destroyed : Sp->destroyed;
    char *inst = INST ? (LONE ? a.seg0_inst : Sp->inst) : nullptr;  // attached ParticleInstance output (or null)
    const uint32_t inst_cap = INST ? (LONE ? a.seg0_inst_cap : Sp->inst_cap) : 0u;
    const uint32_t vt_rounds = a.vt_rounds;  // new-particle tile size: the host's choice (from its bounds)
    // ---- LONE: round 0 requested now, for the particle range this workgroup has if it is a live tile
    float4 q0c, q1c, q2c, q3c;
    float lfc = 0.0f;  // lifetime of a particle that cannot turn (its Q3 plane is not kept: FwOutWin::lf)
    uint32_t spec_base = 0xFFFFFFFFu;  // first particle of the speculative request (none: 0xFFFFFFFF)
    if (LONE) {
        uint32_t ks = 0;  // spawns of this frame, known from the arguments (every inline op belongs to the lone segment)
        if (SPAWN == FW_SPAWN_INLINE)
            for (uint32_t i = 0; i < a.n_ops; i++) ks += inl.ops[i].n;
        const uint32_t kvt = SPAWN == FW_SPAWN_INLINE ? (ks + vt_rounds * BLK - 1u) / (vt_rounds * BLK) : 0u;
        const bool kfront = SPAWN == FW_SPAWN_INLINE && kvt != 0u && kvt <= FW_VFRONT;  // new-particle tiles go first
        const bool is_front = kfront && blockIdx.x < kvt;
        const uint32_t sb = (blockIdx.x - (kfront && !is_front ? kvt : 0u)) * FW_TILE;
        // (capacity and sb are multiples of FW_TILE: sb < C leaves room for a whole tile)
        if (!is_front && sb < C) spec_base = sb;
        const size_t sfirst = spec_base != 0xFFFFFFFFu ? (size_t)spec_base * 16u : (size_t)0;
        const uint32_t i0 = tid * 16u;
        q0c = fw_ld4w(ib + FW_OFF_Q0(C) + sfirst, i0);
        q3c = fw_ld4w(ib + FW_OFF_Q3(C) + sfirst, i0 & m2);
        lfc = fw_ld1w((m2 ? ib + FW_OFF_Q0(C) : ib + FW_OFF_L(C, n_lplanes)) + (m2 ? (size_t)0 : sfirst / 4u), m2 ? 0u : i0 / 4u);
        q1c = fw_ld4w(ib + FW_OFF_Q1(C) + sfirst, i0);
        q2c = fw_ld4w(ib + FW_OFF_Q2(C) + sfirst, i0 & m2);
    }
    const uint32_t sidx = p * g.max_seg + seg, oidx = (p ^ 1u) * g.max_seg + seg;
    // SPAWN_NONE frames may have had this frame's new particles MATERIALISED behind the live ones (Global ops of a frame
    // with Nested entries by fw_k_spawn, Nested children by fw_k_nest_spawn): they form the new-particle tiles here
    // too, loaded instead of generated; the forecast only ever describes the live part [0, count).
    const uint32_t n_in = SPAWN == FW_SPAWN_NONE ? g.count[sidx] : g.count[sidx] + g.spawned[sidx] + g.appended[sidx];
    uint32_t o0 = 0, o1 = 0, n_spawn = SPAWN == FW_SPAWN_NONE ? g.spawned[sidx] + g.appended[sidx] : 0u;
    if (SPAWN == FW_SPAWN_INLINE) {
        for (uint32_t i = 0; i < a.n_ops; i++) {
            if (inl.ops[i].seg == seg) {
                if (o1 == 0) o0 = i;
                o1 = i + 1;
                n_spawn += inl.ops[i].n;
            }
        }
    } else if (SPAWN == FW_SPAWN_TABLE) {
        o0 = a.seg_op_first[seg], o1 = a.seg_op_first[seg + 1];
        for (uint32_t i = o0; i < o1; i++) n_spawn += a.ops[i].n;
    }
    if (SPAWN != FW_SPAWN_NONE) {  // virtual spawns beyond the capacity are dropped (and reported)
        const uint32_t spawn_room = seg_cap - min(n_in, seg_cap);
        if (n_spawn > spawn_room) {
            n_spawn = spawn_room;
            if (blockIdx.x == first && threadIdx.x == 0) atomicOr(g.err, FW_ERR_CAPACITY);
        }
    }
    const uint32_t n_tot = n_in + n_spawn;
    // tiling of [0, n_tot): identical to fw_k_update (live tiles of FW_TILE, then small new-particle tiles).  With
    // front-loading (at most FW_VFRONT new-particle tiles) workgroup b < n_vt is new-particle tile b and workgroup
    // b >= n_vt is live tile b - n_vt, whatever the live count is.
    const uint32_t t_spawn = (n_in + FW_TILE - 1u) / FW_TILE;
    const uint32_t vtile = SPAWN == FW_SPAWN_NONE ? (uint32_t)FW_TILE : vt_rounds * BLK;  // materialised: full tiles
    // A handful of new particles (at most one round) whose segment's last live tile has room for them ride along in
    // THAT tile, as one more round after its live ones: their slots follow its live survivors by construction, so they
    // need no tile of their own (no counting, no look-back).  With thousands of small emitters this halves the number
    // of workgroups -- each of which pays the same ~5 us of launch-time latencies however few particles it holds.
    // (materialised new particles -- SPAWN_NONE -- sit right behind the live ones in the input buffer: for them
    // "riding along" just means that the last live tile's range extends over them)
    const bool merge_new = n_spawn != 0u && n_spawn <= BLK && t_spawn != 0u && n_tot <= t_spawn * FW_TILE;
    const uint32_t n_vt = merge_new ? 0u : (n_spawn + vtile - 1u) / vtile;
    const uint32_t n_act = t_spawn + n_vt;
    if (SPAWN != FW_SPAWN_NONE && n_vt != 0 && n_vt <= FW_VFRONT && t_spawn != 0 && tis < n_act)
        tis = tis < n_vt ? t_spawn + tis : tis - n_vt;
    const uint32_t tile = first + tis;
    const bool has_new = tis >= t_spawn;
    const bool tail_new = merge_new && tis + 1u == t_spawn;  // this live tile also spawns + updates the new particles
    const uint32_t base = has_new ? n_in + (tis - t_spawn) * vtile : tis * FW_TILE;
    const uint32_t lim = has_new ? min(base + vtile, n_tot)
                                 : min(base + FW_TILE, (SPAWN == FW_SPAWN_NONE && tail_new) ? n_tot : n_in);
    fw_u64 *fc_out = a.fc_out;

    if (n_tot == 0 || tis >= n_act) {
        if (tid == 0) {
            if (fc_small) a.fce_out[tile] = make_uint4(0u, 0u, 0u, a.epoch);
            if (n_tot == 0 && tis == 0) {
                g.count[oidx] = 0;
                g.spawned[oidx] = 0;
                g.appended[oidx] = 0;
                g.ndestroyed[seg] = 0;
                if (a.host_counts) a.host_counts[seg] = (unsigned long long)a.epoch << 32;
            }
            if (blockIdx.x == 0 && a.live_next) *a.live_next = 0ull;
            if (blockIdx.x == 0 && a.done_tag) *a.done_tag = a.done_value;
        }
        if (SUMS) fw_fc_housekeeping(a);
        return;
    }
    const bool is_last = tis + 1u == n_act;
    if (tis == 0 && tid == 0 && n_act > seg_tiles) {
        atomicOr(g.err, FW_ERR_CAPACITY);
        g.err[1] = seg, g.err[2] = n_tot, g.err[3] = seg_tiles, g.err[4] = n_in;  // diagnostics
    }
    if (blockIdx.x == 0 && tid == 0 && a.live_next) *a.live_next = 0ull;
    if (blockIdx.x == 0 && tid == 0 && a.done_tag) *a.done_tag = a.done_value;

    const unsigned long long tsA = FW_DBG(a.dbg, 8u) ? (__builtin_amdgcn_s_memrealtime() + (C & 0u)) : 0ull;
    // round 0 of a live tile goes out now (unless the speculative request above already covers it)
    // (Loads are issued UNCONDITIONALLY at an index clamped into the tile: a load under a lane predicate lives in
    // its own basic block, and the copy into the merged value at the end of that block makes the compiler wait for
    // it right there -- the "prefetch" would complete before anything else is issued.  A lane past the end simply
    // re-reads the tile's last particle and ignores it.)
    const uint32_t last = lim - 1u;  // lim > base for an active tile
    // input windows: the planes advanced to the tile's first slot (slot 0 for a new-particle tile, which loads nothing real)
    const bool loaded_tile = !has_new || SPAWN == FW_SPAWN_NONE;  // block-uniform
    const size_t ifirst = loaded_tile ? (size_t)base * 16u : (size_t)0;
    const char *iw0 = ib + FW_OFF_Q0(C) + ifirst, *iw1 = ib + FW_OFF_Q1(C) + ifirst;
    const char *iw2 = ib + FW_OFF_Q2(C) + ifirst, *iw3 = ib + FW_OFF_Q3(C) + ifirst;
    const char *iwl = m2 ? iw0 : ib + FW_OFF_L(C, n_lplanes) + ifirst / 4u;  // lifetime plane (or any valid address)
    if (!LONE || (loaded_tile && base != spec_base)) {  // LONE: only a tile whose role differs from the guess reloads
        const uint32_t i0 = loaded_tile ? min(tid, last - base) * 16u : 0u;
        q0c = fw_ld4w(iw0, i0);
        q3c = fw_ld4w(iw3, i0 & m2);
        lfc = fw_ld1w(iwl, m2 ? 0u : i0 / 4u);
        q1c = fw_ld4w(iw1, i0);
        q2c = fw_ld4w(iw2, i0 & m2);
    }
    const FwType T = g.types[type_idx];  // scalar loads; first needed in the round loop
    if (tid < keys_len) s_keys[tid] = key0;
    for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];

    // forecast prefix of this tile: survivors sitting in the input tiles before it (all live tiles for a new-particle tile)
    uint32_t fc_part;
    bool fc_bad;
    if (fc_small) {
        fc_part = fw_fce_prefix_part<BLK>(fce, seg_tiles, tis, a.epoch, &fc_bad);
    } else {
        fc_part = fw_fc_prefix_part<BLK>(a.fc_in, a.fc_s2, first, first + min(tis, t_spawn), seg_tiles);
        fc_bad = tid == 0 && fw_ld2u(a.fc_in, a.fc_tag).x != a.epoch - 1u;
    }
    // survivors among a new-particle tile's particles: age 0, lifetime = RNG block 2 word 0 (core.rs:455).
    // When the host has established that every particle spawned this frame outlives the step (dt below the smallest
    // lifetime any of this frame's emitters can draw: a.new_static), nothing has to be counted or looked up: new
    // particle k lands right after the live survivors, at slot +k.
    uint32_t new_alive = 0;
    if (SPAWN != FW_SPAWN_NONE && has_new && !a.new_static) {
#pragma unroll 1
        for (uint32_t r = 0; r < vt_rounds; r++) {
            const uint32_t idx = base + r * BLK + tid;
            bool al = false;
            if (idx < lim) {
                const uint32_t k = idx - n_in;
                uint32_t oi = o0;
                for (uint32_t i = o0; i < o1; i++)
                    if (k >= FW_OP(i).rel_base && k - FW_OP(i).rel_base < FW_OP(i).n) oi = i;
                const FwOp &op = FW_OP(oi);
                const FwEmit &e = g.emits[op.emit];
                const unsigned long long serial = op.serial_base + (k - op.rel_base);
                const fw_u4 o = fw_philox4x32_10(fw_u4{(uint32_t)serial, (uint32_t)(serial >> 32), e.emission_index, 2u},
                                                 g.seed, e.uid);
                float an;
                al = fw_survives(0.0f, a.dt, fw_unit_f32(o.x) * (e.life_max - e.life_min) + e.life_min, &an);
            }
            new_alive += (uint32_t)__popcll(__ballot(al));
        }
    }
    if (fc_small && FW_DBG(a.dbg, 16u)) {  // FW_DEBUG 16 (profiling): read the table a second time -- what does the read cost?
        uint4 e2[FW_FCE_U];
        fw_fce_request<BLK>(a.fce_out, first, seg_tiles, e2);
#pragma unroll
        for (int j = 0; j < FW_FCE_U; j++) asm volatile("" ::"v"(e2[j].x), "v"(e2[j].w));
    }
    fc_part = fw_wave_sum(fc_part);
    if (lane == 0) s_part[0][wave] = fc_part, s_part[1][wave] = new_alive;
    if (__any(fc_bad) && lane == 0) fw_raise(g, 2u, 0xFFFFFFFFu, tile);
    const unsigned long long tsB = FW_DBG(a.dbg, 8u) ? (__builtin_amdgcn_s_memrealtime() + (fc_part & 0u)) : 0ull;
    __syncthreads();
    const unsigned long long ts1 = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t excl = 0, new_cnt = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) excl += s_part[0][w], new_cnt += s_part[1][w];

    if (has_new && (a.new_static || SPAWN == FW_SPAWN_NONE)) {
        excl += base - n_in;  // every earlier new particle survives
        // (materialised new particles come here only when the host has shown that: it does not schedule this kernel
        // for such a frame otherwise)
        if (SPAWN == FW_SPAWN_NONE && !a.new_static && tid == 0) fw_raise(g, 3u, 0xFFFFFFFFu, tile);
    } else if (SPAWN != FW_SPAWN_NONE && has_new) {  // look back among the new-particle tiles only
        const bool lb_needed = tis > t_spawn;
        if (lb_needed && tid == 0)
            __hip_atomic_store(&g.tile_status[tile], fw_pack_status(a.epoch, FW_ST_AGG, new_cnt), RLX, AGENT);
        uint32_t lb_excl = 0;
        if (lb_needed) {
            bool timed_out = false;
            lb_excl = fw_lookback<BLK, NW, LBW>(g.tile_status, first + t_spawn, tile, a.epoch, a.spin_limit, s_lb, &timed_out);
            if (timed_out) {  // recount the survivors of the earlier NEW particles (never taken in practice)
                if (tid == 0) atomicOr(g.err, FW_ERR_LOOKBACK_TIMEOUT);
                uint32_t c = 0;
                for (uint32_t i = n_in + tid; i < base; i += BLK) {
                    const uint32_t k = i - n_in;
                    uint32_t oi = o0;
                    for (uint32_t j = o0; j < o1; j++)
                        if (k >= FW_OP(j).rel_base && k - FW_OP(j).rel_base < FW_OP(j).n) oi = j;
                    const FwOp &op = FW_OP(oi);
                    const FwEmit &e = g.emits[op.emit];
                    const unsigned long long serial = op.serial_base + (k - op.rel_base);
                    const fw_u4 o = fw_philox4x32_10(
                        fw_u4{(uint32_t)serial, (uint32_t)(serial >> 32), e.emission_index, 2u}, g.seed, e.uid);
                    float an;
                    c += fw_survives(0.0f, a.dt, fw_unit_f32(o.x) * (e.life_max - e.life_min) + e.life_min, &an) ? 1u : 0u;
                }
                c = fw_wave_sum(c);
                __syncthreads();
                if (lane == 0) s_lb[wave] = c;
                __syncthreads();
                lb_excl = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) lb_excl += s_lb[w];
            }
        }
        if (tid == 0)
            __hip_atomic_store(&g.tile_status[tile], fw_pack_status(a.epoch, FW_ST_INCL, lb_excl + new_cnt), RLX, AGENT);
        excl += lb_excl;
    }

    const unsigned long long ts2 = FW_DBG(a.dbg, 8u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long tsR1 = 0;
    const bool want_destroyed = T.report_destroyed && destroyed != nullptr;
    const uint32_t fcA = excl / FW_TILE, fc_bnd = (fcA + 1u) * FW_TILE;
    FwRoundOut acc{0u, 0u};
    float box[6] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f, FW_F32_MIN, FW_F32_MIN, FW_F32_MIN};
    const bool box_on = a.boxes != 0u;  // workgroup-uniform
    excl = __builtin_amdgcn_readfirstlane(excl);  // workgroup-uniform (summed from LDS): keep it on the scalar unit
    const FwOutWin W = fw_out_window(ob, C, excl, T, a.force_colors, n_lplanes);
    uint32_t run = excl;
    int rr = 0;  // rounds done so far (the wave-count exchange area is double-buffered by round parity)
    if (loaded_tile) {
        // ---- live tile (or a tile of materialised new particles): stream the rounds that hold particles (a segment's
        // last tile is partial; with thousands of small emitters that is every tile)
        const int n_rounds = (int)((lim - base + BLK - 1u) / BLK);
        rr = n_rounds;
#pragma unroll 1
        for (int r = 0; r < n_rounds; r++) {
            const uint32_t idx = base + r * BLK + tid;
            const uint32_t in_ = min((r + 1) * BLK + tid, last - base) * 16u;  // next round's slot (clamped: see above)
            const float4 q0n = fw_ld4w(iw0, in_);
            const float4 q3n = fw_ld4w(iw3, in_ & m2);
            const float lfn = fw_ld1w(iwl, m2 ? 0u : in_ / 4u);
            const float4 q1n = fw_ld4w(iw1, in_);
            const float4 q2n = fw_ld4w(iw2, in_ & m2);
            if (!m2) q3c = make_float4(0.0f, 0.0f, 0.0f, lfc);
            const bool valid = idx < lim;
            float age_new;
            const bool alive = valid && fw_survives(q0c.w, a.dt, q3c.w, &age_new);
            const unsigned long long m = __ballot(alive);
            if (lane == 0) s_c[r & 1][wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t wbase = run;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t c = s_c[r & 1][w];
                if ((uint32_t)w < wave) wbase += c;
                run += c;
            }
            const uint32_t o = wbase + fw_lane_prefix(m);
            // the lane's instance record goes to its rank in the wave's LDS area as soon as each part is computed
            float4 *rec = (INST && inst != nullptr) ? s_inst + wave * 256u + (o - wbase) * 4u : nullptr;
            fw_round_finish(T, s_keys, a.dt, a.dbg, q0c, q1c, q2c, q3c, valid, alive, true, age_new, idx, o, ib, ob, W,
                            destroyed, want_destroyed, C, n_lplanes, true, fc_bnd, acc, rec, box, box_on, idx >= n_in);
            if (INST && inst != nullptr && !FW_DBG(a.dbg, 2u)) fw_inst_flush(inst, inst_cap, s_inst + wave * 256u, lane, m, wbase);
            q0c = q0n, q1c = q1n, q2c = q2n, q3c = q3n, lfc = lfn;
            if (FW_DBG(a.dbg, 8u) && r == 0) tsR1 = __builtin_amdgcn_s_memrealtime() + (o & 0u);
        }
    }
    if (SPAWN != FW_SPAWN_NONE && (!loaded_tile || tail_new)) {
        // ---- new-particle tile (or the one extra round of a live tile that carries its segment's few new particles):
        // spawn_particles (src/core.rs:437-469) right before update_particles, per slot
        const uint32_t sbase = tail_new ? n_in : base, slim = tail_new ? n_tot : lim;
        const uint32_t srounds = tail_new ? 1u : vt_rounds;
#pragma unroll 1
        for (uint32_t r = 0; r < srounds; r++) {
            const uint32_t idx = sbase + r * BLK + tid;
            const bool valid = idx < slim;
            const uint32_t cb = (uint32_t)(rr + (int)r) & 1u;
            FwSpawnOut so;
            so.q0 = so.q1 = so.q2 = so.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                const uint32_t k = idx - n_in;
                uint32_t oi = o0;
                for (uint32_t i = o0; i < o1; i++)
                    if (k >= FW_OP(i).rel_base && k - FW_OP(i).rel_base < FW_OP(i).n) oi = i;
                const FwOp &op = FW_OP(oi);
                so = fw_spawn_one(g.emits[op.emit], g.seed, op.serial_base + (k - op.rel_base),
                                  fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                                  fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                                  fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
            }
            float age_new;
            const bool alive = valid && fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);
            const unsigned long long m = __ballot(alive);
            if (lane == 0) s_c[cb][wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t wbase = run;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t c = s_c[cb][w];
                if ((uint32_t)w < wave) wbase += c;
                run += c;
            }
            const uint32_t o = wbase + fw_lane_prefix(m);
            float4 *rec = (INST && inst != nullptr) ? s_inst + wave * 256u + (o - wbase) * 4u : nullptr;
            fw_round_finish(T, s_keys, a.dt, a.dbg, so.q0, so.q1, so.q2, so.q3, valid, alive, false, age_new, idx, o, ib,
                            ob, W, destroyed, want_destroyed, C, n_lplanes, true, fc_bnd, acc, rec, box, box_on);
            if (INST && inst != nullptr && !FW_DBG(a.dbg, 2u)) fw_inst_flush(inst, inst_cap, s_inst + wave * 256u, lane, m, wbase);
        }
    }
    if (lane == 0) s_part[2][wave] = acc.fa, s_part[3][wave] = acc.fb;
    __syncthreads();
    if (tid == 0) {
        uint32_t sa = 0, sb = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) sa += s_part[2][w], sb += s_part[3][w];
        if (fc_small) a.fce_out[tile] = make_uint4(sa, sb, fcA, a.epoch);
        else fw_fc_add(fc_out, a.fc_s2, first + fcA, sa, sb, seg_tiles);
    }
    if (a.boxes) fw_tile_box_flush<NW>(g.tile_box, tile, a.epoch, box, reinterpret_cast<float (*)[6]>(s_lb));
    if (FW_DBG(a.dbg, 8u) && g.dbg_ts && tid == 0) {
        unsigned long long *d = g.dbg_ts + 32768 + ((size_t)(a.epoch & 1u) * gridDim.x + tile) * 8;  // two launches kept
        {  // ring of the last 256 launches: earliest start / latest end, spread over 64 words each to keep the atomics apart
            unsigned long long *rg = g.dbg_ts + (size_t)(a.epoch & 255u) * 128u;
            atomicMax(&rg[blockIdx.x & 63u], ~ts0);  // max of the complement = min (slots are recycled with 0)
            atomicMax(&rg[64u + (blockIdx.x & 63u)], __builtin_amdgcn_s_memrealtime());
            if (blockIdx.x < 128u) g.dbg_ts[(size_t)((a.epoch + 128u) & 255u) * 128u + blockIdx.x] = 0ull;
        }
        const unsigned long long tsE = __builtin_amdgcn_s_memrealtime();
        d[0] = ts0, d[1] = ts1, d[2] = ts2, d[3] = tsE;
        // HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20): which CU / XCD ran this tile
        const unsigned hwid = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
        d[4] = tsA, d[5] = tsB, d[6] = ((unsigned long long)xcc << 32) | hwid, d[7] = tsR1;
    }
    if (SUMS) fw_fc_housekeeping(a);
    if (is_last && tid == 0) {
        const uint32_t nc = run;  // excl + survivors of this tile
        g.count[oidx] = nc;
        g.spawned[oidx] = 0;
        g.appended[oidx] = 0;
        g.ndestroyed[seg] = n_tot - nc;
        if (a.host_counts) a.host_counts[seg] = ((unsigned long long)a.epoch << 32) | nc;  // one 8-byte store: tag + count
        if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, (unsigned long long)n_tot);  // (FW_DEBUG 128: profiling, no statistics)
    }
}

// ---------------------------------------------------------------------------------
// FIFO (ring) segments: update_particles in place (fw_kernels.h: FwFifoSeg).  One workgroup per ring tile of FW_TILE
// slots that holds a live or a new particle; a lane owns the same slot from load to store, so there is no compaction,
// no cross-wave exchange and no barrier in the round loop -- the waves of a workgroup drift apart and the loads of one
// overlap the arithmetic and the stores of another.  What the host says about each slot (destroyed / live / spawned
// this frame) is re-derived from the particle itself and a disagreement raises FW_ERR_FORECAST.
// Per live particle the kernel reads the four state planes (64 B) and writes position+age and velocity (32 B), the
// colour planes whose gradient is not constant, the scale unless its curve is constant, and rotation / angular
// velocity only in waves where they changed: 132 B for the linear 2-key curves of configs[1] instead of 164.
// ---------------------------------------------------------------------------------
// the ParticleInstance records of a wave's survivors: consecutive in the output unless the wave straddles the ring's head
template <bool INST, bool NT = false>
__device__ __forceinline__ void fw_fifo_inst_out(const FwFifoSeg &F, char *inst, const float4 *s_inst_wave, const float4 *rec,
                                                 uint32_t lane, unsigned long long m, bool alive, uint32_t o) {
    if (!INST || inst == nullptr || m == 0ull) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    const uint32_t o_first = __builtin_amdgcn_readlane(o, __ffsll((long long)m) - 1);
    const uint32_t o_last = __builtin_amdgcn_readlane(o, 63 - __clzll((long long)m));
    if (o_last - o_first + 1u == cnt) {
        fw_inst_flush<NT>(inst, F.inst_cap, s_inst_wave, lane, m, o_first);
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (alive && o < F.inst_cap)
            for (uint32_t k = 0; k < 4; k++) fw_st4(inst + (size_t)o * 64u, k, rec[k]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The physics_avian arm of update_particles (core.rs:607-624) for particle types that live in rings: a type with collision
// settings and destroy_on_collision == false changes neither age, lifetime nor the order of its particles -- a bounce only
// replaces `position + velocity * dt` and the velocity the drag step starts from (core.rs:626-643) -- so the "deaths are a
// prefix" / "nobody young dies" arguments of the ring paths hold unchanged and the type keeps its in-place update
// (examples/stress_test_collision.rs:92-115 is such a type).  COLL instantiations of the ring kernels only: the plain ones
// never see a collider and keep their register budget.  destroy_on_collision types stay on the count -> scan ->
// fw_k_update_coll path (a collision that removes a particle changes the survivor count).
struct FwCollArm {
    bool on;  // the workgroup's particle type has collision settings (workgroup-uniform)
    float restitution, friction;
    uint32_t mask;
};
template <bool COLL>
__device__ __forceinline__ FwCollArm fw_coll_arm(const FwGlobals &g, uint32_t type_idx) {
    if constexpr (COLL) {
        const FwTypeColl TC = g.type_coll[type_idx];
        return FwCollArm{(TC.coll_flags & FW_COLL_ENABLED) != 0u, TC.coll_restitution, TC.coll_friction, TC.coll_mask};
    } else {
        return FwCollArm{false, 0.0f, 0.0f, 0u};
    }
}
// position / velocity after particle_collision (a particle that meets nothing comes out as position + velocity * dt, velocity)
template <bool COLL>
__device__ __forceinline__ void fw_coll_step(const FwGlobals &g, const FwCollArm &A, bool active, float dt, float4 q0, float4 q1,
                                             fw_v3 *cpos, fw_v3 *cvel) {
    *cpos = fw_v3{q0.x, q0.y, q0.z}, *cvel = fw_v3{q1.x, q1.y, q1.z};
    if constexpr (COLL) {
        if (A.on && active) fw_particle_collision(cpos, cvel, dt, A.restitution, A.friction, false, A.mask, g.colliders, g.n_colliders);
    }
}

// WM: the optional planes this launch's particle types write (fw_integrate_store), or -1 = read from the type
#ifndef FW_FIFO_UNROLL
#define FW_FIFO_UNROLL 4
#endif
template <bool INST, int WM, int NT = 0, bool COLL = false>
__global__ __launch_bounds__(FW_BLOCK) void fw_k_update_fifo(FwGlobals g, FwFifoArgs a, FwInlineOps inl) {
    constexpr int BLK = FW_BLOCK;
    constexpr int NW = BLK / 64;
    constexpr int R = FW_TILE / BLK;  // (smaller ring tiles were measured: 2 rounds 26.7 us, 1 round 26.2 us, 4 rounds 24.6 us)
    constexpr uint32_t TILE = FW_TILE;
    __shared__ __attribute__((aligned(16))) float s_keys[FW_KEYS_MAX];
    __shared__ __attribute__((aligned(16))) float4 s_inst[INST ? NW * 256 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t j = 0;  // this workgroup's segment (block-uniform; the records ride in the kernel arguments)
    for (uint32_t i = 1; i < a.n_segs; i++)
        if (a.s[i].tile_first <= blockIdx.x) j = i;
    const FwFifoSeg &F = a.s[j];
    const uint32_t C = F.capacity, ring_tiles = C / TILE;
    const uint32_t tis = blockIdx.x - F.tile_first;
    const uint32_t n_vt = F.n_vt_a + F.n_vt_b;
    const bool spawner = tis < n_vt;  // a workgroup of new particles (they come first: the longest job starts earliest)
    uint32_t pt = F.tile0 + (tis - n_vt);
    if (pt >= ring_tiles) pt -= ring_tiles;
    const uint32_t head = F.head, n_dead = F.dead;
    const uint32_t sidx = a.parity * g.max_seg + F.seg;
    // materialised new particles (FwFifoSeg::mat): the counters are requested now and used after the barrier below
    uint32_t c_cnt = 0, c_new = 0;
    if (F.mat) c_cnt = g.count[sidx], c_new = g.spawned[sidx] + g.appended[sidx];
    // (uniform values: kept on the scalar unit)
    const uint32_t n_in = F.mat ? __builtin_amdgcn_readfirstlane(c_cnt + c_new) : F.n_in;
    const uint32_t full_from = F.mat ? __builtin_amdgcn_readfirstlane(c_cnt) : 0xFFFFFFFFu;  // first-update particles
    const uint32_t n_tot = n_in + F.n_spawn;
    // first slot of the workgroup: a ring tile, or the slot of the first new particle of its group (the groups [0, a) and
    // [a, n_spawn) of the new particles each occupy consecutive slots: a is where the ring wraps, FwFifoSeg::spawn_a)
    const uint32_t k0 = spawner ? (tis < F.n_vt_a ? tis * BLK : F.spawn_a + (tis - F.n_vt_a) * BLK) : 0u;
    uint32_t sbase = spawner ? head + n_in + k0 : pt * TILE;
    if (sbase >= C) sbase -= C;
    if (spawner && sbase >= C) sbase -= C;  // (head + n_in + k0 < 3 C)
    const float key0 = tid < F.keys_len ? g.keys[F.keys_off + tid] : 0.0f;
    char *buf = F.buf;
    const size_t sfirst = (size_t)sbase * 16u;
    const char *iw0 = buf + FW_OFF_Q0(C) + sfirst, *iw1 = buf + FW_OFF_Q1(C) + sfirst;
    const char *iw2 = buf + FW_OFF_Q2(C) + sfirst, *iw3 = buf + FW_OFF_Q3(C) + sfirst;
    // round 0 of a live tile (every slot of a tile exists -- the capacity is a multiple of FW_TILE -- so the loads
    // need no bounds; a spawning workgroup loads nothing)
    float4 q0c, q1c, q2c, q3c, q0n, q1n, q2n, q3n;  // rounds 0 and 1: the loop keeps two rounds of loads in flight
    q0c = q1c = q2c = q3c = q0n = q1n = q2n = q3n = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a ring whose live count only the device knows is launched over its whole capacity: its tiles look at the counters
    // first and the empty ones leave without having asked for a byte of particle data)
    // a type that cannot turn (FW_TYPE_NOSPIN): rotation is FwType::const_rot, angular velocity 0, the lifetime the type's one
    // value -- neither the Q2 nor the Q3 plane is read (every lane asks for the tile's first slot instead: one line per
    // wave, loads stay unconditional)
    const bool nospin = (F.type_idx & FW_TYPE_IDX_NOSPIN) != 0u;
    const uint32_t m2 = nospin ? 0u : 0xFFFFFFFFu;
    const float4 q3s = make_float4(0.0f, 0.0f, 0.0f, F.life);
    const bool defer = F.mat != 0u && F.n_in == 0xFFFFFFFFu;
    const uint32_t i1 = (uint32_t)(min(1, R - 1) * BLK + (int)tid) * 16u;
    if (!spawner && !defer) {
        q0c = fw_ld4w<NT == 2>(iw0, tid * 16u), q3c = fw_ld4w<NT == 2>(iw3, (tid * 16u) & m2);
        q1c = fw_ld4w<NT == 2>(iw1, tid * 16u), q2c = fw_ld4w<NT == 2>(iw2, (tid * 16u) & m2);
        q0n = fw_ld4w<NT == 2>(iw0, i1), q3n = fw_ld4w<NT == 2>(iw3, i1 & m2);
        q1n = fw_ld4w<NT == 2>(iw1, i1), q2n = fw_ld4w<NT == 2>(iw2, i1 & m2);
    }
    if (defer) {
        uint32_t i0 = sbase - head;
        if (sbase < head) i0 += C;
        if (tis != 0u && !(i0 < n_tot || (i0 + TILE > C && n_tot != 0u))) return;
        q0c = fw_ld4w<NT == 2>(iw0, tid * 16u), q3c = fw_ld4w<NT == 2>(iw3, (tid * 16u) & m2);
        q1c = fw_ld4w<NT == 2>(iw1, tid * 16u), q2c = fw_ld4w<NT == 2>(iw2, (tid * 16u) & m2);
        q0n = fw_ld4w<NT == 2>(iw0, i1), q3n = fw_ld4w<NT == 2>(iw3, i1 & m2);
        q1n = fw_ld4w<NT == 2>(iw1, i1), q2n = fw_ld4w<NT == 2>(iw2, i1 & m2);
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (a.live_next) *a.live_next = 0ull;
        if (a.done_tag) *a.done_tag = a.done_value;
    }
    const FwType T = g.types[F.type_idx & ~FW_TYPE_IDX_NOSPIN];
    const FwCollArm CA = fw_coll_arm<COLL>(g, F.type_idx & ~FW_TYPE_IDX_NOSPIN);
    if (tid < F.keys_len) s_keys[tid] = key0;
    for (uint32_t i = tid + BLK; i < F.keys_len; i += BLK) s_keys[i] = g.keys[F.keys_off + i];
    __syncthreads();
    const bool want_destroyed = T.report_destroyed && F.destroyed != nullptr;
    if (!spawner && tis != 0u) {
        // a tile without a single particle (the grid of a segment whose count only the device knows covers its ring)
        uint32_t i0 = sbase - head;
        if (sbase < head) i0 += C;
        if (!(i0 < n_tot || (i0 + TILE > C && n_tot != 0u))) return;
    }
    char *inst = INST ? F.inst : nullptr;
    float4 *s_inst_wave = s_inst + (INST ? wave * 256u : 0u);
    const FwOutWin W = fw_out_window(buf, C, sbase, T, 0u);
    bool bad = false;
    if (spawner) {
        // ---- this frame's new particles: spawn_particles (core.rs:437-469) right before update_particles, each in the
        // slot it will live in.  One round per workgroup: spawning is ~5x the arithmetic of an update.
        const uint32_t k = k0 + tid;
        const uint32_t k_end = tis < F.n_vt_a ? F.spawn_a : F.n_spawn;
        const bool is_new = k < k_end;
        const uint32_t i = n_in + k, s = sbase + tid;
        FwSpawnOut so;
        so.q0 = so.q1 = so.q2 = so.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t new_ei = 0u;  // emission index of the entry that spawns this lane's particle
        if (is_new) {
            uint32_t oi = F.op0;
            for (uint32_t x = F.op0; x < F.op1; x++)
                if (k >= inl.ops[x].rel_base && k - inl.ops[x].rel_base < inl.ops[x].n) oi = x;
            const FwOp &op = inl.ops[oi];
            new_ei = g.emits[op.emit].emission_index;
            so = fw_spawn_one(g.emits[op.emit], g.seed, op.serial_base + (k - op.rel_base),
                              fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                              fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                              fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
        }
        float age_new;
        const bool surv = fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);
        const bool dead = i < n_dead;
        bad |= is_new && surv == dead;
        const bool alive = is_new && !dead;
        const unsigned long long m = __ballot(alive);
        float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(m) * 4u : nullptr;
        fw_v3 cpos, cvel;
        fw_coll_step<COLL>(g, CA, alive, a.dt, so.q0, so.q1, &cpos, &cvel);
        if (alive) {
            fw_integrate_store<true, -1, NT>(T, s_keys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, W, s, rec, (COLL && CA.on) ? &cpos : nullptr,
                                         (COLL && CA.on) ? &cvel : nullptr, nullptr, false, true);
            if (F.n_lplanes) fw_init_last_emitted(g, g.segs[F.seg], buf, s, new_ei, so.q3.w);  // (a type other particles' entries emit from)
        } else if (is_new && want_destroyed)  // born and destroyed in the same frame (dt >= lifetime)
            fw_store_destroyed(F.destroyed, buf, C, s, false, T, s_keys, so.q0, so.q1, so.q2, so.q3, age_new, i);
        fw_fifo_inst_out<INST, NT == 2>(F, inst, s_inst_wave, rec, lane, m, alive, i - n_dead);
    } else {
        // ---- (1) records of the particles this update destroys (core.rs:596-599).  Kept out of the streaming loop:
        // memory reads inside a divergent branch make the compiler drain every outstanding load -- the prefetch
        // included -- where the branches join.
        if (want_destroyed && n_dead != 0u) {
#pragma unroll 1
            for (int r = 0; r < R; r++) {
                const uint32_t s = sbase + r * BLK + tid;
                uint32_t i = s - head;
                if (s < head) i += C;
                if (i < n_dead && i < n_in) {
                    const uint32_t b16 = (uint32_t)(r * BLK + (int)tid) * 16u;
                    const float4 q0 = fw_ld4w<NT == 2>(iw0, b16), q1 = fw_ld4w<NT == 2>(iw1, b16), q2 = fw_ld4w<NT == 2>(iw2, b16 & m2);
                    const float4 q3 = nospin ? q3s : fw_ld4w<NT == 2>(iw3, b16);
                    // (a materialised particle that dies in its first update carries the spawn-time colours and scale, like any
                    // particle born and destroyed in one frame: evaluated, not read -- the planes of a FW_TYPE_DERIVED type
                    // are not maintained, and for everybody else they hold exactly these values)
                    fw_store_destroyed(F.destroyed, buf, C, s, i < full_from, T, s_keys, q0, q1, q2, q3, q0.w + a.dt, i);
                }
            }
        }
        // ---- (2) the particles that were here before this frame: a streaming loop, next round's loads in flight while
        // this one is integrated and stored.  No barrier, no exchange between lanes: the waves of a workgroup drift apart.
#pragma unroll FW_FIFO_UNROLL
        for (int r = 0; r < R; r++) {
            const uint32_t s = sbase + r * BLK + tid;
            const uint32_t in_ = (uint32_t)(min(r + 2, R - 1) * BLK + (int)tid) * 16u;  // two rounds ahead (the last re-read)
            const float4 q0f = fw_ld4w<NT == 2>(iw0, in_), q3f = fw_ld4w<NT == 2>(iw3, in_ & m2);
            const float4 q1f = fw_ld4w<NT == 2>(iw1, in_), q2f = fw_ld4w<NT == 2>(iw2, in_ & m2);
            if (nospin) q3c = q3s;
            uint32_t i = s - head;  // logical index of the slot
            if (s < head) i += C;
            float age_new;
            const bool surv = fw_survives(q0c.w, a.dt, q3c.w, &age_new);
            const bool mine = i < n_in, dead = i < n_dead;
            bad |= mine && surv == dead;  // the host's cohort ages and the particle disagree
            const bool alive = mine && !dead;
            const unsigned long long m = INST ? __ballot(alive) : 0ull;
            float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(m) * 4u : nullptr;
            fw_v3 cpos, cvel;
            fw_coll_step<COLL>(g, CA, alive, a.dt, q0c, q1c, &cpos, &cvel);
            if (alive) {
                if FW_DBG(a.dbg, 2u) {  // profiling only: stream without arithmetic
                    const uint32_t b16 = (s - W.first) * 16u;
                    fw_st4w<NT == 2>(W.q0, b16, make_float4(q0c.x, q0c.y, q0c.z, age_new)), fw_st4w<NT == 2>(W.q1, b16, q1c);
                    if (WM >= 0 ? (WM & 1) != 0 : W.wr5) fw_st4w<NT != 0>(W.q5, b16, q0c);
                    if (WM >= 0 ? (WM & 2) != 0 : W.wr6) fw_st4w<NT != 0>(W.q6, b16, q1c);
                    if (WM >= 0 ? (WM & 4) != 0 : T.sc_kind != 0) fw_st1w<NT != 0>(W.s4, (s - W.first) * 4u, q1c.w);
                } else {
                    fw_integrate_store<true, WM, NT>(T, s_keys, a.dt, q0c, q1c, q2c, q3c, age_new, W, s, rec, (COLL && CA.on) ? &cpos : nullptr,
                                                 (COLL && CA.on) ? &cvel : nullptr, nullptr, false, i >= full_from);
                }
            }
            fw_fifo_inst_out<INST, NT == 2>(F, inst, s_inst_wave, rec, lane, m, alive, i - n_dead);
            q0c = q0n, q1c = q1n, q2c = q2n, q3c = q3n;
            q0n = q0f, q1n = q1f, q2n = q2f, q3n = q3f;
        }
    }
    if (__any(bad) && lane == 0) fw_raise(g, 4u, F.seg, blockIdx.x);
    if (tis == 0 && tid == 0) {
        const uint32_t oidx = (a.parity ^ 1u) * g.max_seg + F.seg;
        if (F.mat ? (F.n_in != 0xFFFFFFFFu && F.n_in != n_in) : g.count[sidx] != n_in)
            fw_raise(g, 5u, F.seg, n_in);
        if (F.report) *F.report = ((unsigned long long)a.epoch << 32) | c_new;
        const uint32_t nc = n_tot - min(n_dead, n_tot);
        g.count[oidx] = nc;
        g.spawned[oidx] = 0;
        g.appended[oidx] = 0;
        g.ndestroyed[F.seg] = n_tot - nc;
        if (a.host_counts) a.host_counts[F.seg] = ((unsigned long long)a.epoch << 32) | nc;
        if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, (unsigned long long)n_tot);
    }
}


// ---------------------------------------------------------------------------------
// Range rings (fw_kernels.h: FwRangeRec): update_particles IN PLACE for particle types whose lifetime is a range.
// Three kinds of workgroups in one launch, dispatched in this order:
//   OLD    1024 particles of the part of the list that may lose particles this frame (age + dt >= lifetime.min),
//          counted from the young part downwards.  The whole input of the tile is held in registers before its survivor
//          count is published; the exclusive count of the tiles nearer to the young part (decoupled look-back, those
//          tiles have lower workgroup indices) is the tile's output offset; survivors are integrated and stored
//          packed against the young part, order kept (core.rs:589-659).  The last active OLD tile of a segment knows
//          the total and does the segment's bookkeeping.
//   NEW    256 of this frame's new particles: spawn_particles (core.rs:437-469) + their first update, each in the slot
//          it will live in (behind the young part).
//   YOUNG  a ring tile of 1024 slots: in place, like fw_k_update_fifo without the death test.
// Slots are addressed as 32-bit byte offsets from the plane base (capacity <= FW_RANGE_MAX_CAPACITY).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void fw_store_destroyed_vals(char *dbuf, size_t d, const FwType &T, float4 q0, float4 q1, float4 q2,
                                                        float4 q3, float age_new, const float bc[4], const float em[4], float sc) {
    float *rec = reinterpret_cast<float *>(dbuf) + d * 26;
    q2 = fw_record_rotation(T, q2);
    rec[0] = q0.x, rec[1] = q0.y, rec[2] = q0.z;
    rec[3] = q1.x, rec[4] = q1.y, rec[5] = q1.z;
    rec[6] = q2.x, rec[7] = q2.y, rec[8] = q2.z, rec[9] = q2.w;
    rec[10] = q3.x, rec[11] = q3.y, rec[12] = q3.z;
    rec[13] = q1.w, rec[14] = sc, rec[15] = age_new, rec[16] = q3.w;
    rec[17] = bc[0], rec[18] = bc[1], rec[19] = bc[2], rec[20] = bc[3];
    rec[21] = em[0], rec[22] = em[1], rec[23] = em[2], rec[24] = em[3];
    reinterpret_cast<int32_t *>(rec)[25] = T.pbr;
}

// ParticleInstance records of a wave's particles of one round (range rings, windowed hand-off: fw_kernels.h): the lanes in
// `m` hold records for consecutive indices -- ascending with the lane, or (OLD tiles: reversed) descending -- staged in the
// wave's LDS area in ascending index order; a wave whose run is broken (the young part wraps around the whole ring) stores
// lane by lane.
template <bool NT = false>
__device__ __forceinline__ void fw_range_inst_out(char *inst, uint32_t inst_cap, const float4 *s_inst_wave, const float4 *rec,
                                                  uint32_t lane, unsigned long long m, uint32_t idx, bool reversed) {
    if (inst == nullptr || m == 0ull) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    const uint32_t i_lo_lane = __builtin_amdgcn_readlane(idx, __ffsll((long long)m) - 1);
    const uint32_t i_hi_lane = __builtin_amdgcn_readlane(idx, 63 - __clzll((long long)m));
    const uint32_t lowest = reversed ? i_hi_lane : i_lo_lane, highest = reversed ? i_lo_lane : i_hi_lane;
    if (highest - lowest + 1u == cnt) {
        fw_inst_flush<NT>(inst, inst_cap, s_inst_wave, lane, m, lowest);
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (((m >> lane) & 1ull) && idx < inst_cap)
            for (uint32_t k = 0; k < 4; k++) fw_st4(inst + (size_t)idx * 64u, k, rec[k]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

#ifndef FW_RANGE_YR
#define FW_RANGE_YR 4  // rounds of a YOUNG workgroup: it covers FW_RANGE_YR * 256 ring slots
#endif
#ifndef FW_RANGE_PF_ALL
#define FW_RANGE_PF_ALL 1  // a YOUNG workgroup of an all-NOSPIN launch requests all its rounds up front (0: two rounds in flight;
                          // configs[2] 321 -> 316 us, one GPU's share of configs[4] 87.9 -> 86.6 us, profiles/r04/strip_pf_ab.txt)
#endif
uint32_t fw_range_young_tile(void) { return FW_RANGE_YR * FW_BLOCK; }

// INST: some segment of the launch has a WINDOWED instance buffer attached (fw_spawner_attach_instances_window): every
// survivor's ParticleInstance record goes to index  n_old_in + (its position in the young part)  /  n_old_in - 1 - (its
// new distance from the young part)  -- both known to a tile without waiting for anybody: the records of the frame are
// d_out[first, first + count) with first = the particles this update destroyed (n_old_in - n_old_out = ndestroyed), in list
// order.  (An index counted from 0 would need the old part's survivor total, which only its last tile knows.)
template <bool ALLNOSPIN, bool INST, int NT, bool COLL = false>
__global__ __launch_bounds__(FW_BLOCK) void fw_k_update_range(FwGlobals g, FwRangeArgs a) {
    constexpr int BLK = FW_BLOCK;
    constexpr int NW = BLK / 64;
    constexpr int R = FW_TILE / BLK;
    constexpr int LBW = 4;
    constexpr uint32_t TILE = FW_TILE;
    __shared__ __attribute__((aligned(16))) float s_keys[FW_KEYS_MAX];
    __shared__ __attribute__((aligned(16))) float4 s_inst[INST ? NW * 256 : 1];  // per wave: 64 records of 4 float4
    __shared__ uint32_t s_cnt[R][NW];
    __shared__ uint32_t s_lb[2 * LBW * NW];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const FwRangeDesc &D = a.desc[blockIdx.x];  // block-uniform: scalar loads
    const uint32_t seg = D.seg, role = D.role_k >> 30, k = D.role_k & 0x3FFFFFFFu;
#ifdef FW_RANGE_SLEEP  // (experiment: what a microsecond more of dead time per workgroup costs)
    __builtin_amdgcn_s_sleep(FW_RANGE_SLEEP);
#endif
#ifdef FW_RANGE_STAMP  // (an instrumented build only -- make timeline: the four scalar stores and the branch cost 6 % of configs[4]'s share even unused)
    struct Stamp {  // FW_DEBUG & 8 (tools/range_timeline.py): when the workgroup started and when its wave 0 left, by whichever return
        unsigned long long *p;
        __device__ ~Stamp() {
            if (p && threadIdx.x == 0) p[3] = __builtin_amdgcn_s_memrealtime();
        }
    } stamp{FW_DBG(a.dbg, 8u) && a.ts ? a.ts + (size_t)blockIdx.x * 8u : nullptr};
    if (stamp.p && tid == 0) stamp.p[0] = __builtin_amdgcn_s_memrealtime(), stamp.p[4] = D.role_k, stamp.p[5] = seg;
#define FW_STAMP(i, dep) do { if (stamp.p && tid == 0) stamp.p[i] = __builtin_amdgcn_s_memrealtime() + ((unsigned long long)(dep) & 0ull); } while (0)
#else
#define FW_STAMP(i, dep) do { } while (0)
#endif
    const bool nospin = ALLNOSPIN || (D.type_idx & FW_TYPE_IDX_NOSPIN) != 0u;
    const uint32_t m2 = nospin ? 0u : 0xFFFFFFFFu;
    const uint32_t keys_off = D.keys_off, keys_len = D.keys_len;
    const float key0 = tid < keys_len ? g.keys[keys_off + tid] : 0.0f;
    const FwRangeRec &Rc = a.recs[seg];  // pinned host memory
    const uint32_t b = Rc.b, n_spawn_h = Rc.n_spawn, rflags = Rc.flags;
    uint32_t y_exist = Rc.y_exist;
    const FwSeg *Sp = &g.segs[seg];
    const uint32_t C = Sp->capacity;
    char *buf = Sp->buf[0];
    const uint32_t sidx = a.parity * g.max_seg + seg, oidx = (a.parity ^ 1u) * g.max_seg + seg;
    // Rings of spawners with Nested entries (FW_RREC_MAT / FW_RREC_DEV): this frame's new particles were materialised behind the
    // young part before the update, and for a type that receives children only the device knows how many particles it holds.
    // The young part is then  everything  -  the old part (rold + grad: survivors of the last update + the cohorts that joined
    // this frame); the particles from index y_full on were born this frame and get their first update with every plane written.
    uint32_t y_full = 0xFFFFFFFFu, n_old_dev = 0u;
    if (rflags & (FW_RREC_MAT | FW_RREC_DEV)) {
        const uint32_t c0 = g.count[sidx], c1 = g.spawned[sidx] + g.appended[sidx];
        n_old_dev = g.rold[sidx] + Rc.grad;
        y_full = c0 - min(c0, n_old_dev);
        y_exist = y_full + c1;
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (a.live_next) *a.live_next = 0ull;
        if (a.done_tag) *a.done_tag = a.done_value;
    }
    const char *p0 = buf + FW_OFF_Q0(C), *p1 = buf + FW_OFF_Q1(C), *p2 = buf + FW_OFF_Q2(C), *p3 = buf + FW_OFF_Q3(C);
    const char *pl = buf + FW_OFF_L(C, Sp->n_lplanes);  // lifetimes of a type that cannot turn (FwOutWin::lf)
    char *inst = INST ? Sp->inst : nullptr;
    const uint32_t inst_cap = INST ? Sp->inst_cap : 0u;
    float4 *s_inst_wave = s_inst + (INST ? wave * 256u : 0u);

    if (role == FW_RANGE_YOUNG) {
        // ---- in place: a lane owns its slot from load to store
        constexpr int YR = FW_RANGE_YR;
        constexpr uint32_t YT = YR * BLK;  // (capacities are multiples of it: the host rounds them, fw_range_young_tile)
        const uint32_t ring_tiles = C / YT;
        const uint32_t need = min(ring_tiles, (b % YT + y_exist + YT - 1u) / YT);
        if (k >= need) return;
        const uint32_t cnt_y = (INST && inst != nullptr && !(rflags & (FW_RREC_MAT | FW_RREC_DEV))) ? g.count[sidx] : 0u;  // (requested now, used later)
        // record index of the first young particle (= n_old_in)
        const uint32_t rec0 = (rflags & (FW_RREC_MAT | FW_RREC_DEV)) ? n_old_dev : (cnt_y > y_exist ? cnt_y - y_exist : 0u);
        uint32_t pt = b / YT + k;
        if (pt >= ring_tiles) pt -= ring_tiles;
        const uint32_t sbase = pt * YT;
        FW_STAMP(1, sbase);  // the descriptor, the pinned record and the segment record have arrived
        // The slots of this tile that hold young particles: [w_lo, w_lo + w_n) -- from the tile's first slot when that lies
        // inside the young part, from b otherwise.  Loads go through descriptors clipped to that window: the partly filled tiles
        // at the two ends of the young part fetch only what they own.  (A young part that wraps around nearly the whole ring
        // could re-enter the tile at its end: such a tile loads all its slots, as before.)
        uint32_t yi0 = sbase - b;
        if (sbase < b) yi0 += C;
        uint32_t w_lo = sbase, w_n = YT;
        if (y_exist + YT <= C) {
            if (yi0 < y_exist) w_n = min(YT, y_exist - yi0);
            else w_lo = b, w_n = (b >= sbase && b - sbase < YT) ? min(sbase + YT - b, y_exist) : 0u;
        }
        const fw_rsrc r0 = fw_make_rsrc(p0 + (size_t)w_lo * 16u, w_n * 16u), r1 = fw_make_rsrc(p1 + (size_t)w_lo * 16u, w_n * 16u);
        const fw_rsrc rl = fw_make_rsrc(pl + (size_t)w_lo * 4u, w_n * 4u);
        // this lane's byte offset in the window, round r; a slot below the window gets an offset far beyond it (clipped like one
        // above it) -- not the wrapped negative one, whose last bytes would wrap back to offset 0 in the range check
        const int wd0 = (int)(sbase - w_lo) + (int)tid;
        auto woff = [&](int r) -> uint32_t {
            const int d = wd0 + r * BLK;
            return d < 0 ? 0x7FFFFFF0u : (uint32_t)d * 16u;
        };
#if FW_RANGE_PF_ALL
        if constexpr (ALLNOSPIN) {
            // every round's loads requested up front (9 VGPRs per round for a type that cannot turn: the kernel's budget is set
            // by the OLD path, which holds a whole tile): twice the bytes in flight per streaming workgroup
            constexpr int PF = YR < 4 ? YR : 4;  // rounds in flight (a workgroup of more rounds refills the slot it has just used)
            float4 q0a[PF], q1a[PF];
            float lfa[PF];
#pragma unroll
            for (int r = 0; r < PF; r++) {
                const uint32_t ir = woff(r);
                q0a[r] = fw_ldb4<NT == 2>(r0, ir), lfa[r] = fw_ldb1<NT == 2>(rl, ir / 4u), q1a[r] = fw_ldb4<NT == 2>(r1, ir);
            }
            const FwType T = g.types[D.type_idx & ~FW_TYPE_IDX_NOSPIN];
            const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & ~FW_TYPE_IDX_NOSPIN);
            if (tid < keys_len) s_keys[tid] = key0;
            for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
            __syncthreads();
            const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes);
            bool bad = false;
#pragma unroll
            for (int r = 0; r < YR; r++) {
                const uint32_t s = sbase + r * BLK + tid;
                uint32_t yi = s - b;  // index within the young part
                if (s < b) yi += C;
                const bool mine = yi < y_exist;
                const float4 q0v = q0a[r % PF], q1v = q1a[r % PF];
                const float4 q3v = make_float4(0.0f, 0.0f, 0.0f, lfa[r % PF]);
                if (r + PF < YR) {
                    const uint32_t ir = woff(r + PF);
                    q0a[r % PF] = fw_ldb4<NT == 2>(r0, ir), lfa[r % PF] = fw_ldb1<NT == 2>(rl, ir / 4u), q1a[r % PF] = fw_ldb4<NT == 2>(r1, ir);
                }
                float age_new;
                const bool surv = fw_survives(q0v.w, a.dt, q3v.w, &age_new);
                bad |= mine && !surv;
                const unsigned long long mi = (INST && inst != nullptr) ? __ballot(mine) : 0ull;
                float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(mi) * 4u : nullptr;
                fw_v3 cpos, cvel;
                fw_coll_step<COLL>(g, CA, mine, a.dt, q0v, q1v, &cpos, &cvel);
                if (mine)
                    fw_integrate_store<true, -1, NT>(T, s_keys, a.dt, q0v, q1v, q3v, q3v, age_new, W, s, rec, (COLL && CA.on) ? &cpos : nullptr,
                                                     (COLL && CA.on) ? &cvel : nullptr, nullptr, false, yi >= y_full);
                if (INST) fw_range_inst_out<NT == 2>(inst, inst_cap, s_inst_wave, rec, lane, mi, rec0 + yi, false);
            }
            if (__any(bad) && lane == 0) fw_raise(g, 7u, seg, blockIdx.x);
            return;
        }
#endif
        float4 q0c, q1c, q2c, q3c, q0n, q1n, q2n, q3n;
        float lfc, lfn;
        // (a type that cannot turn reads neither rotation nor angular velocity: a zero-length window; one that can reads its
        // lifetime in Q3, not in the lifetime plane)
        const fw_rsrc r2 = fw_make_rsrc(p2 + (size_t)w_lo * 16u, m2 ? w_n * 16u : 0u), r3 = fw_make_rsrc(p3 + (size_t)w_lo * 16u, m2 ? w_n * 16u : 0u);
        const fw_rsrc rlf = fw_make_rsrc(pl + (size_t)w_lo * 4u, m2 ? 0u : w_n * 4u);
        const uint32_t i0 = woff(0), i1 = woff(min(1, YR - 1));
        q0c = fw_ldb4<NT == 2>(r0, i0), q3c = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r3, i0), lfc = fw_ldb1<NT == 2>(rlf, i0 / 4u);
        q1c = fw_ldb4<NT == 2>(r1, i0), q2c = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r2, i0);
        q0n = fw_ldb4<NT == 2>(r0, i1), q3n = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r3, i1), lfn = fw_ldb1<NT == 2>(rlf, i1 / 4u);
        q1n = fw_ldb4<NT == 2>(r1, i1), q2n = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r2, i1);
        const FwType T = g.types[D.type_idx & ~FW_TYPE_IDX_NOSPIN];
        const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & ~FW_TYPE_IDX_NOSPIN);
        if (tid < keys_len) s_keys[tid] = key0;
        for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
        __syncthreads();
        FW_STAMP(2, T.flags);  // type record + keys in LDS
        const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes);
        bool bad = false;
        FW_STAMP(6, __float_as_uint(q0c.w) | __float_as_uint(q1c.w));  // the first round's particles have arrived
#pragma unroll
        for (int r = 0; r < YR; r++) {
            const uint32_t s = sbase + r * BLK + tid;
            const uint32_t in_ = woff(min(r + 2, YR - 1));  // two rounds ahead (the last re-read)
            const float4 q0f = fw_ldb4<NT == 2>(r0, in_), q3f = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r3, in_);
            const float lff = fw_ldb1<NT == 2>(rlf, in_ / 4u);
            const float4 q1f = fw_ldb4<NT == 2>(r1, in_), q2f = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r2, in_);
            if (nospin) q3c = make_float4(0.0f, 0.0f, 0.0f, lfc);
            uint32_t yi = s - b;  // index within the young part
            if (s < b) yi += C;
            const bool mine = yi < y_exist;
            float age_new;
            const bool surv = fw_survives(q0c.w, a.dt, q3c.w, &age_new);
            bad |= mine && !surv;  // the host's cohort ages say nobody young can die
            const unsigned long long mi = (INST && inst != nullptr) ? __ballot(mine) : 0ull;
            float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(mi) * 4u : nullptr;
            fw_v3 cpos, cvel;
            fw_coll_step<COLL>(g, CA, mine, a.dt, q0c, q1c, &cpos, &cvel);
            if (mine)
                fw_integrate_store<true, -1, NT>(T, s_keys, a.dt, q0c, q1c, q2c, q3c, age_new, W, s, rec, (COLL && CA.on) ? &cpos : nullptr,
                                                 (COLL && CA.on) ? &cvel : nullptr, nullptr, false, yi >= y_full);
            if (INST) fw_range_inst_out<NT == 2>(inst, inst_cap, s_inst_wave, rec, lane, mi, rec0 + yi, false);
            q0c = q0n, q1c = q1n, q2c = q2n, q3c = q3n, lfc = lfn;
            q0n = q0f, q1n = q1f, q2n = q2f, q3n = q3f, lfn = lff;
        }
        if (__any(bad) && lane == 0) fw_raise(g, 7u, seg, blockIdx.x);
        return;
    }

    const uint32_t cnt_in = g.count[sidx];
    const uint32_t n_old_in = (rflags & (FW_RREC_MAT | FW_RREC_DEV)) ? n_old_dev : (cnt_in > y_exist ? cnt_in - y_exist : 0u);
    const uint32_t n_added = (rflags & (FW_RREC_MAT | FW_RREC_DEV)) ? y_exist - y_full : 0u;  // materialised this frame
    // (what does not fit is dropped and reported, as everywhere: the host grows a segment before its bound reaches the capacity)
    const uint32_t room = C - min(C, n_old_in + y_exist);
    const uint32_t n_spawn = min(n_spawn_h, room);

    if (role == FW_RANGE_NEW) {
        if (k * BLK >= n_spawn_h) return;
        const FwType T = g.types[D.type_idx & ~FW_TYPE_IDX_NOSPIN];
        const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & ~FW_TYPE_IDX_NOSPIN);
        if (tid < keys_len) s_keys[tid] = key0;
        for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
        __syncthreads();
        const uint32_t kk = k * BLK + tid;
        const bool is_new = kk < n_spawn;
        if (kk < n_spawn_h && kk == n_spawn) atomicOr(g.err, FW_ERR_CAPACITY);
        const unsigned long long mi = (INST && inst != nullptr) ? __ballot(is_new) : 0ull;
        float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(mi) * 4u : nullptr;
        if (!INST && !is_new) return;
        uint32_t s = b + y_exist;  // < 2 C
        if (s >= C) s -= C;
        s += kk;                   // < 2 C
        if (s >= C) s -= C;
        if (is_new) {
            uint32_t oi = Rc.op0;
            for (uint32_t x = Rc.op0; x < Rc.op0 + Rc.op_n; x++)
                if (kk >= a.ops[x].rel_base && kk - a.ops[x].rel_base < a.ops[x].n) oi = x;
            const FwOp &op = a.ops[oi];
            const FwSpawnOut so = fw_spawn_one(g.emits[op.emit], g.seed, op.serial_base + (kk - op.rel_base),
                                               fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                                               fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                                               fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
            float age_new;
            const bool surv = fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);
            if (!surv) fw_raise(g, 8u, seg, kk);
            // (the colour plane of a constant gradient holds that colour in every slot since the buffer was allocated)
            const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes);
            fw_v3 cpos, cvel;
            fw_coll_step<COLL>(g, CA, true, a.dt, so.q0, so.q1, &cpos, &cvel);
            fw_integrate_store<false, -1, NT>(T, s_keys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, W, s, rec, (COLL && CA.on) ? &cpos : nullptr,
                                              (COLL && CA.on) ? &cvel : nullptr);
            if (Sp->n_lplanes) fw_init_last_emitted(g, *Sp, buf, s, g.emits[op.emit].emission_index, so.q3.w);  // (other particles' entries emit from it)
        }
        if (INST) fw_range_inst_out<NT == 2>(inst, inst_cap, s_inst_wave, rec, lane, mi, n_old_in + y_exist + kk, false);
        return;
    }

    // ---- OLD: in-place compaction towards the young part.  Distance d from the young part: slot = b - 1 - d.
    // (ONE workgroup walking the few tiles of a small old part itself -- no status words, no waiting, no provisioned-but-idle
    // workgroups -- was built and measured in round 4: the tile loop costs the kernel 9-35 VGPRs, and even at equal occupancy
    // one GPU's share of configs[4] ran 86.4 us against 85.6 with the tiles in parallel: profiles/r04/range_seq_old_ab.txt)
    const uint32_t base = k * TILE;
    if (k == 0u && tid == 0u && n_old_in > D.n_old * TILE) {  // the host's bound of the old part was not one (internal error)
        atomicOr(g.err, FW_ERR_CAPACITY);
        g.err[1] = seg, g.err[2] = n_old_in, g.err[3] = D.n_old, g.err[4] = cnt_in;
    }
    // (the host's young count against the device's own record of the old part: what both derive the list's first slot from)
    if (k == 0u && tid == 0u && !(rflags & (FW_RREC_MAT | FW_RREC_DEV)) && g.rold[sidx] + Rc.grad != n_old_in) fw_raise(g, 10u, seg, n_old_in);
    const bool want_destroyed_any = Sp->destroyed != nullptr;
    if (base >= n_old_in) {
        if (k == 0u && tid == 0u) {  // nobody old: the segment's bookkeeping is still this workgroup's
            const uint32_t nc = y_exist + n_spawn;
            g.count[oidx] = nc, g.spawned[oidx] = 0, g.appended[oidx] = 0, g.ndestroyed[seg] = 0, g.rold[oidx] = 0;
            if (Rc.report) *Rc.report = ((unsigned long long)a.epoch << 32) | n_added;
            if (a.host_counts) a.host_counts[seg] = ((unsigned long long)a.epoch << 32) | nc;
            if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
            if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, (unsigned long long)nc);
        }
        return;
    }
    const uint32_t lim = min(base + TILE, n_old_in);
    const uint32_t bm1 = b + C - 1u;
    float4 q0[R], q1[R], q2[R], q3[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t d = min(base + r * BLK + tid, lim - 1u);
        uint32_t s = bm1 - d;  // in [0, 2 C)
        if (s >= C) s -= C;
        q0[r] = fw_ld4w<NT == 2>(p0, s * 16u);
        q3[r] = fw_ld4w_opt<ALLNOSPIN, NT == 2>(p3, (s * 16u) & m2);
        const float lf = fw_ld1w<NT == 2>(m2 ? p0 : pl, m2 ? 0u : s * 4u);
        q1[r] = fw_ld4w<NT == 2>(p1, s * 16u);
        q2[r] = fw_ld4w_opt<ALLNOSPIN, NT == 2>(p2, (s * 16u) & m2);
        if (nospin) q3[r] = make_float4(0.0f, 0.0f, 0.0f, lf);
    }
    // a type other particles' entries emit from (Nested, core.rs:471-546) carries last_emitted_age per entry: those planes
    // move with the survivors (at most FW_RANGE_LK of them: the host keeps types with more off the range path)
    constexpr int FW_RANGE_LK = 2;
    const uint32_t nlp = Sp->n_lplanes;
    float lkv[FW_RANGE_LK][R];
#pragma unroll
    for (int j = 0; j < FW_RANGE_LK; j++)
#pragma unroll
        for (int r = 0; r < R; r++) lkv[j][r] = 0.0f;
    if (nlp) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t d = min(base + r * BLK + tid, lim - 1u);
            uint32_t s = bm1 - d;
            if (s >= C) s -= C;
#pragma unroll
            for (int j = 0; j < FW_RANGE_LK; j++)
                if ((uint32_t)j < nlp) lkv[j][r] = fw_ld1w<NT == 2>(buf + FW_OFF_L(C, j), s * 4u);
        }
    }
    const FwType T = g.types[D.type_idx & ~FW_TYPE_IDX_NOSPIN];
    const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & ~FW_TYPE_IDX_NOSPIN);
    if (tid < keys_len) s_keys[tid] = key0;
    for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
    float age_new[R];
    unsigned long long m[R];
    // Everything this tile will ever read from its slots is in registers before it publishes: the tiles that wait for
    // its count go on to overwrite those slots.  So every wave's count passes, on its way to LDS (and from there, behind
    // the barrier, into the published word), through an opaque instruction that also consumes one component of every
    // loaded vector (a load returns whole): the compiler must have waited for all of them before the count exists.
#pragma unroll
    for (int r = 0; r < R; r++) {
        const bool valid = base + r * BLK + tid < lim;
        const bool alive = valid && fw_survives(q0[r].w, a.dt, q3[r].w, &age_new[r]);
        m[r] = __ballot(alive);
        uint32_t c = (uint32_t)__popcll(m[r]);
        asm volatile("; fw_k_update_range: input held before the count is published"
                     : "+v"(c) : "v"(q0[r].x), "v"(q1[r].x), "v"(q2[r].x), "v"(q3[r].w));
        if (nlp) asm volatile("; ... and the last_emitted_age planes" : "+v"(c) : "v"(lkv[0][r]), "v"(lkv[1][r]));
        if (lane == 0) s_cnt[r][wave] = c;
    }
    __syncthreads();
    uint32_t tile_surv = 0;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int w = 0; w < NW; w++) tile_surv += s_cnt[r][w];
    const uint32_t tile = D.old_first + k;  // the segment's look-back words: [old_first, old_first + its OLD workgroups)
    // (FW_DEBUG 256, `make ab` only -- fault injection: the segment's second OLD tile never publishes, whoever waits for it
    // times out: tests/test_gpu_range.py drives the error path with it)
    const bool withhold = FW_DBG(a.dbg, 256u) && k == 1u;
    uint32_t excl = 0;
    if (k != 0u) {
        if (tid == 0 && !withhold) __hip_atomic_store(&a.status[tile], fw_pack_status(a.epoch, FW_ST_AGG, tile_surv), RLX, AGENT);
        bool timed_out = false;
        excl = fw_lookback<BLK, NW, LBW>(a.status, D.old_first, tile, a.epoch, a.spin_limit * 64u + 1024u, s_lb, &timed_out);
        // (no recount is possible: a predecessor that has not published may not have read its slots yet.  Predecessors have
        // lower workgroup indices, so they are resident or done: the wait is bounded.)
        if (timed_out && tid == 0) fw_raise(g, 9u, seg, tile);
    }
    if (tid == 0 && !withhold) __hip_atomic_store(&a.status[tile], fw_pack_status(a.epoch, FW_ST_INCL, excl + tile_surv), RLX, AGENT);
    const bool want_destroyed = T.report_destroyed && want_destroyed_any;
    const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes);
    uint32_t run = excl;
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t wbase = run;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = s_cnt[r][w];
            if ((uint32_t)w < wave) wbase += c;
            run += c;
        }
        const uint32_t d = base + r * BLK + tid;
        const bool valid = d < lim;
        const bool alive = (m[r] >> lane) & 1ull;
        const uint32_t od = wbase + fw_lane_prefix(m[r]);  // survivors nearer to the young part = the new distance
        // (records: index n_old_in - 1 - od, descending with the lane -- staged in reverse so that LDS holds them ascending)
        float4 *rec = (INST && inst != nullptr) ? s_inst_wave + ((uint32_t)__popcll(m[r]) - 1u - fw_lane_prefix(m[r])) * 4u : nullptr;
        fw_v3 cpos, cvel;
        fw_coll_step<COLL>(g, CA, alive, a.dt, q0[r], q1[r], &cpos, &cvel);
        if (alive) {
            uint32_t s = bm1 - od;
            if (s >= C) s -= C;
            fw_integrate_store<false, -1, NT>(T, s_keys, a.dt, q0[r], q1[r], q2[r], q3[r], age_new[r], W, s, rec, (COLL && CA.on) ? &cpos : nullptr,
                                              (COLL && CA.on) ? &cvel : nullptr);
            if (nlp) {
#pragma unroll
                for (int j = 0; j < FW_RANGE_LK; j++)
                    if ((uint32_t)j < nlp) fw_st1w<NT == 2>(buf + FW_OFF_L(C, j), s * 4u, lkv[j][r]);
            }
        }
        if (INST) fw_range_inst_out<NT == 2>(inst, inst_cap, s_inst_wave, rec, lane, m[r], n_old_in - 1u - od, true);
        if (!alive && valid && want_destroyed) {
            // destroyed record (core.rs:596-599): the clone with the age advanced, pose, colours and scale of the previous
            // frame.  The colour and scale planes of the slot hold what the previous update computed from the age that was
            // just loaded: evaluated again here (same functions, same inputs) instead of being read -- the slot may
            // already belong to somebody else.  Records are filled from the END of the buffer (the youngest dead first):
            // fw_spawner_read_destroyed reads the last `ndestroyed` records, which are then in list order.
            const float ap = q0[r].w / q3[r].w;
            float bc[4], em[4];
            fw_gradient_sample(T.bc_kind, T.bc_n, s_keys + T.o_bc_t, s_keys + T.o_bc_v, ap, bc);
            fw_gradient_sample(T.em_kind, T.em_n, s_keys + T.o_em_t, s_keys + T.o_em_v, ap, em);
            const float sc = q1[r].w * fw_curve_sample(T.sc_kind, T.sc_n, s_keys, s_keys + T.o_sc_v, ap);
            const uint32_t dead_rank = d - od;  // the dead nearer to the young part
            fw_store_destroyed_vals(Sp->destroyed, (size_t)(C - 1u - dead_rank), T, q0[r], q1[r], q2[r], q3[r], q0[r].w + a.dt, bc, em, sc);
        }
    }
    if (lim == n_old_in && tid == 0) {  // the tile furthest from the young part knows the totals
        const uint32_t n_old_out = excl + tile_surv;
        const uint32_t nc = n_old_out + y_exist + n_spawn;
        g.count[oidx] = nc, g.spawned[oidx] = 0, g.appended[oidx] = 0, g.rold[oidx] = n_old_out;
        g.ndestroyed[seg] = n_old_in - n_old_out;
        if (Rc.report) *Rc.report = ((unsigned long long)a.epoch << 32) | n_added;
        if (a.host_counts) a.host_counts[seg] = ((unsigned long long)a.epoch << 32) | nc;
        if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, (unsigned long long)(n_old_in + y_exist + n_spawn));
    }
}

// split mode, pass 1: survivors per tile
__global__ __launch_bounds__(FW_BLOCK) void fw_k_count(FwGlobals g, FwUpdateArgs a) {
    __shared__ uint32_t s_c[4];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t seg = fw_upper_slot(a.seg_tile_first, a.n_seg, tile);
    const uint32_t tis = tile - a.seg_tile_first[seg];
    const uint32_t sidx = a.parity * g.max_seg + seg;
    const uint32_t n_tot = g.count[sidx] + g.spawned[sidx] + g.appended[sidx];
    const uint32_t base = tis * FW_TILE;
    uint32_t c = 0;
    if (base < n_tot) {
        const FwSeg &S = g.segs[seg];
        const char *ib = S.buf[a.parity];
        const FwTypeColl &T = g.type_coll[S.type_idx];
        const bool nospin = (g.types[S.type_idx].flags & FW_TYPE_NOSPIN) != 0u;
        const bool coll_kill = (T.coll_flags & (FW_COLL_ENABLED | FW_COLL_DESTROY)) == (FW_COLL_ENABLED | FW_COLL_DESTROY);
        for (int r = 0; r < FW_ROUNDS; r++) {
            const uint32_t idx = base + r * FW_BLOCK + tid;
            if (idx < n_tot) {
                float an;
                const float4 q0 = fw_ld4(ib + FW_OFF_Q0(S.capacity), idx);
                bool al = fw_survives(q0.w, a.dt, fw_load_q3(ib, S.capacity, S.n_lplanes, idx, nospin).w, &an);
                if (al && coll_kill) {  // destroy_on_collision removes particles too (core.rs:636-639)
                    const float4 q1 = fw_ld4(ib + FW_OFF_Q1(S.capacity), idx);
                    fw_v3 pos{q0.x, q0.y, q0.z}, vel{q1.x, q1.y, q1.z};
                    al = !fw_particle_collision(&pos, &vel, a.dt, T.coll_restitution, T.coll_friction, true, T.coll_mask,
                                                g.colliders, g.n_colliders);
                }
                c += al ? 1u : 0u;
            }
        }
    }
    c = fw_wave_sum(c);
    if ((tid & 63u) == 0) s_c[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) g.tile_cnt[tile] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// split mode, pass 3 for frames with colliding particle types (FW_MODE_SPLIT_COLL): update_particles with the
// physics_avian arm (core.rs:607-624, 633-639, 744-800).  Same tiling as fw_k_count (tiles of FW_TILE over
// [0, count + spawned + appended), everything materialised), output offset from fw_k_scan; per round: load, age test,
// particle_collision for types that have collision settings, rank, integrate, store.  This is the feature path, not
// the tuned one: no forecast, no fused spawn (the streaming kernels never run collisions and keep their registers).
__global__ __launch_bounds__(FW_BLOCK) void fw_k_update_coll(FwGlobals g, FwUpdateArgs a) {
    constexpr int NW = FW_BLOCK / 64;
    __shared__ __attribute__((aligned(16))) float s_keys_lds[FW_KEYS_MAX];
    __shared__ uint32_t s_c[2][NW];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t seg = fw_upper_slot(a.seg_tile_first, a.n_seg, tile);
    const uint32_t tis = tile - a.seg_tile_first[seg];
    const uint32_t p = a.parity;
    const uint32_t sidx = p * g.max_seg + seg, oidx = (p ^ 1u) * g.max_seg + seg;
    const uint32_t n_before = g.count[sidx];  // particles that existed before this frame's spawns
    const uint32_t n_tot = n_before + g.spawned[sidx] + g.appended[sidx];
    const uint32_t base = tis * FW_TILE;
    if (blockIdx.x == 0 && tid == 0 && a.live_next) *a.live_next = 0ull;
    if (blockIdx.x == 0 && tid == 0 && a.done_tag) *a.done_tag = a.done_value;
    if (base >= n_tot) {
        if (tid == 0 && n_tot == 0 && tis == 0) {  // empty segment: its first tile still owns the bookkeeping
            g.count[oidx] = 0, g.spawned[oidx] = 0, g.appended[oidx] = 0, g.ndestroyed[seg] = 0;
            if (a.host_counts) a.host_counts[seg] = (unsigned long long)a.epoch << 32;
        }
        return;
    }
    const FwSeg *Sp = &g.segs[seg];
    const uint32_t C = Sp->capacity, n_lplanes = Sp->n_lplanes;
    const char *ib = Sp->buf[p];
    char *ob = Sp->buf[p ^ 1u];
    char *destroyed = Sp->destroyed;
    char *inst = Sp->inst;
    const uint32_t inst_cap = Sp->inst_cap;
    const FwType T = g.types[Sp->type_idx];
    // curves of any length (curve.rs:40-75): what fits the staging area is sampled from LDS, longer key sets straight from
    // device memory (this is the feature path; the streaming kernels only ever see types whose keys fit)
    const bool bigkeys = T.keys_len > FW_KEYS_MAX;
    if (!bigkeys)
        for (uint32_t i = tid; i < T.keys_len; i += FW_BLOCK) s_keys_lds[i] = g.keys[T.keys_off + i];
    __syncthreads();
    const float *s_keys = bigkeys ? g.keys + T.keys_off : s_keys_lds;
    const FwTypeColl TC = g.type_coll[Sp->type_idx];
    const bool coll = (TC.coll_flags & FW_COLL_ENABLED) != 0u, coll_kill = (TC.coll_flags & FW_COLL_DESTROY) != 0u;
    const bool want_destroyed = T.report_destroyed && destroyed != nullptr;
    const uint32_t excl = g.tile_off[tile];
    const FwOutWin W = fw_out_window(ob, C, excl, T, a.force_colors, n_lplanes);
    uint32_t run = excl;
    const uint32_t lim = min(base + FW_TILE, n_tot);
    const int n_rounds = (int)((lim - base + FW_BLOCK - 1u) / FW_BLOCK);
#pragma unroll 1
    for (int r = 0; r < n_rounds; r++) {
        const uint32_t idx = base + r * FW_BLOCK + tid;
        const bool valid = idx < lim;
        const uint32_t li = min(idx, lim - 1u);
        const float4 q0 = fw_ld4(ib + FW_OFF_Q0(C), li), q1 = fw_ld4(ib + FW_OFF_Q1(C), li),
                     q2 = fw_ld4(ib + FW_OFF_Q2(C), li), q3 = fw_load_q3(ib, C, n_lplanes, li, (T.flags & FW_TYPE_NOSPIN) != 0u);
        float age_new;
        const bool young = valid && fw_survives(q0.w, a.dt, q3.w, &age_new);
        fw_v3 cpos{q0.x, q0.y, q0.z}, cvel{q1.x, q1.y, q1.z};
        bool killed = false;
        if (young && coll)
            killed = fw_particle_collision(&cpos, &cvel, a.dt, TC.coll_restitution, TC.coll_friction, coll_kill, TC.coll_mask,
                                           g.colliders, g.n_colliders);
        const bool alive = young && !killed;
        const unsigned long long m = __ballot(alive);
        if (lane == 0) s_c[r & 1][wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t wbase = run;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = s_c[r & 1][w];
            if ((uint32_t)w < wave) wbase += c;
            run += c;
        }
        const uint32_t o = wbase + fw_lane_prefix(m);
        if (alive) {
            float4 rec[4];
            fw_integrate_store(T, s_keys, a.dt, q0, q1, q2, q3, age_new, W, o, inst ? rec : nullptr, coll ? &cpos : nullptr,
                               coll ? &cvel : nullptr);
            if (inst != nullptr && o < inst_cap) {
                fw_st4(inst, o * 4u + 0u, rec[0]), fw_st4(inst, o * 4u + 1u, rec[1]);
                fw_st4(inst, o * 4u + 2u, rec[2]), fw_st4(inst, o * 4u + 3u, rec[3]);
            }
            for (uint32_t k = 0; k < n_lplanes; k++) fw_st1(ob + FW_OFF_L(C, k), o, fw_ld1(ib + FW_OFF_L(C, k), idx));
        } else if (valid && want_destroyed) {
            if (!killed) {  // died of age: the clone with the advanced age, pose of the previous frame (core.rs:596-599)
                // (idx >= count: materialised this frame, never updated -- spawn-time colours and scale, evaluated)
                fw_store_destroyed(destroyed, ib, C, idx, idx < n_before, T, s_keys, q0, q1, q2, q3, age_new, idx - o);
            } else {  // destroyed by a collision (core.rs:633-639): new position, velocity and scale; the rest as loaded
                const float sc = q1.w * fw_curve_sample(T.sc_kind, T.sc_n, s_keys, s_keys + T.o_sc_v, age_new / q3.w);
                float *rec = reinterpret_cast<float *>(destroyed) + (size_t)(idx - o) * 26;
                float4 bc, em;
                if ((T.flags & FW_TYPE_DERIVED) && idx < n_before) {
                    float unused;
                    fw_derived_values(T, s_keys, q0.w, q3.w, q1.w, &bc, &em, &unused);
                } else {  // (a particle materialised this frame: its slot was written in full when it was spawned)
                    bc = fw_ld4(ib + FW_OFF_Q5(C), idx), em = fw_ld4(ib + FW_OFF_Q6(C), idx);
                }
                // (a type that cannot turn keeps no rotation plane: the loaded q2 is whatever the slot last held)
                const float4 r2 = fw_record_rotation(T, q2);
                rec[0] = cpos.x, rec[1] = cpos.y, rec[2] = cpos.z, rec[3] = cvel.x, rec[4] = cvel.y, rec[5] = cvel.z;
                rec[6] = r2.x, rec[7] = r2.y, rec[8] = r2.z, rec[9] = r2.w, rec[10] = q3.x, rec[11] = q3.y, rec[12] = q3.z;
                rec[13] = q1.w, rec[14] = sc, rec[15] = age_new, rec[16] = q3.w;
                rec[17] = bc.x, rec[18] = bc.y, rec[19] = bc.z, rec[20] = bc.w;
                rec[21] = em.x, rec[22] = em.y, rec[23] = em.z, rec[24] = em.w;
                reinterpret_cast<int32_t *>(rec)[25] = T.pbr;
            }
        }
    }
    if (lim == n_tot && tid == 0) {  // the segment's last tile
        const uint32_t nc = run;
        g.count[oidx] = nc, g.spawned[oidx] = 0, g.appended[oidx] = 0;
        g.ndestroyed[seg] = n_tot - nc;
        if (a.host_counts) a.host_counts[seg] = ((unsigned long long)a.epoch << 32) | nc;
        if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, (unsigned long long)n_tot);  // (FW_DEBUG 128: profiling, no statistics)
    }
}

// split mode, pass 2: one workgroup per segment scans its tiles
__global__ __launch_bounds__(FW_BLOCK) void fw_k_scan(FwGlobals g, FwUpdateArgs a) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_run;
    const uint32_t seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t first = a.seg_tile_first[seg], n_tiles = a.seg_tile_first[seg + 1] - first;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += FW_BLOCK) {
        const uint32_t t = t0 + tid;
        const uint32_t v = t < n_tiles ? g.tile_cnt[first + t] : 0u;
        uint32_t inc = v;  // wave inclusive scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if (lane >= (uint32_t)o) inc += u;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += s_w[w];
        const uint32_t run = s_run;
        if (t < n_tiles) g.tile_off[first + t] = run + woff + inc - v;
        __syncthreads();
        if (tid == FW_BLOCK - 1) s_run = run + woff + inc;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// Nested emission (reference src/core.rs:471-546): parents -> children
// ---------------------------------------------------------------------------------

struct FwNestCtx {
    uint32_t op, tile_in_op, n_par;
};

__device__ __forceinline__ uint32_t fw_nest_children(const FwEmit &e, float age, float lea, float lifetime, float *next) {
    const uint64_t n = fw_emission_count(age, lea, lifetime, e.n_start, e.n_end, e.n_count, next);  // core.rs:490-498
    return n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;
}

__device__ __forceinline__ uint32_t fw_find_nest_op(const FwNestOp *ops, uint32_t n_ops, uint32_t tile) {
    uint32_t lo = 0, hi = n_ops;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (ops[mid].first_tile <= tile)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Nested emission in ONE launch per emission level (was: count, scan, spawn = three latency-bound launches).
// A workgroup owns a tile of FW_NEST_TILE parents of one op:
//   1. per parent: compute_emission_count from (age, last_emitted_age, lifetime) -- device fp32, IEEE divide, fmodf,
//      truncf, contraction off: bit-exact -- and store the advanced last_emitted_age (core.rs:490-500);
//   2. the tile's child total is published and the exclusive prefix over the earlier tiles of the op comes from a
//      decoupled look-back (status words tagged with the launch's sequence number: no memset between launches).  The
//      launch runs alone in the stream, a few hundred tiles at most, so the hop costs ~1 us here, not the ~3 us it
//      costs under a streaming update;
//   3. children are written WAVE-COOPERATIVELY: a wave's parents of one round have their counts prefix-summed; lane l
//      of the wave then takes child c = 64 m + l of the wave (binary search of c in the prefix finds its parent, whose
//      pose sits in LDS), so consecutive lanes write consecutive child slots -- seven coalesced plane stores per 64
//      children instead of one lane walking through up to `count` children with scattered stores.  Child order stays
//      parent-major (core.rs:488-544): slot = base + prefix(parent) + k.
//   4. the LAST workgroup to finish (a ticket per op) adds the op's total to the child segment's `appended` count and
//      to the entry's RNG serial: nobody can still be reading the counters that fix the parent bound (core.rs:488).
struct FwNestInline {
    FwNestOp ops[FW_INLINE_OPS];
};

__global__ __launch_bounds__(FW_BLOCK) void fw_k_nest(FwGlobals g, FwNestInline inl, const FwNestOp *ops, uint32_t n_ops,
                                                     uint32_t parity, uint32_t tag, uint32_t spin_limit, uint32_t dbg) {
    constexpr int NW = FW_BLOCK / 64;
    constexpr int LBW = 4;
    constexpr int NR = FW_NEST_TILE / FW_BLOCK;  // rounds per tile
    __shared__ uint32_t s_w[NR][NW];
    __shared__ uint32_t s_lb[2 * LBW * NW];
    __shared__ uint32_t s_inc[NW][64];
    __shared__ __attribute__((aligned(16))) float4 s_par[NW][3][64];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t oi = 0;
    FwNestOp op = inl.ops[0];
    if (ops) {
        oi = fw_find_nest_op(ops, n_ops, tile);
        op = ops[oi];
    } else {
#pragma unroll  // constant indices: the record stays in scalar registers (a dynamic index would go through scratch)
        for (uint32_t i = 1; i < FW_INLINE_OPS; i++)
            if (i < n_ops && inl.ops[i].first_tile <= tile) oi = i, op = inl.ops[i];
    }
    const uint32_t sidx = parity * g.max_seg + op.parent_seg, cidx = parity * g.max_seg + op.child_seg;
    // (a RANGE ring is addressed through the size of its old part, which only the device knows: one dependent hop more before
    // the parents can be requested)
    const uint32_t parent_head = op.parent_range ? fw_range_head(op.parent_head, g.rold[sidx], op.parent_cap) : op.parent_head;
    // the parents' counting inputs: requested now, at an index clamped into the buffer, together with the counters below
    const uint32_t pbase = (tile - op.first_tile) * FW_NEST_TILE;
    float p_age[FW_NEST_TILE / FW_BLOCK], p_life[FW_NEST_TILE / FW_BLOCK], p_lea[FW_NEST_TILE / FW_BLOCK];
#pragma unroll
    for (int r = 0; r < FW_NEST_TILE / FW_BLOCK; r++) {
        const uint32_t ci = fw_ring_slot(parent_head, min(pbase + r * FW_BLOCK + tid, op.parent_cap - 1u), op.parent_cap);
        p_age[r] = fw_ld4(op.parent_buf + FW_OFF_Q0(op.parent_cap), ci).w;
        p_life[r] = (op.parent_nospin != 0u && op.parent_life_plane == 0xFFFFFFFFu)
                        ? op.parent_life_const
                        : fw_load_q3(op.parent_buf, op.parent_cap, op.parent_life_plane, ci, op.parent_nospin != 0u).w;
        p_lea[r] = fw_ld1(op.parent_buf + FW_OFF_L(op.parent_cap, op.parent_lplane), ci);
    }
    const uint32_t n_par = g.count[sidx] + g.spawned[sidx] + g.appended[sidx];  // bound fixed once (core.rs:488)
    const uint32_t cbase = g.count[cidx] + g.spawned[cidx] + g.appended[cidx];  // first child slot of the op
    const unsigned long long serial0 = g.emit_serial[op.emit_slot];
    const uint32_t base = (tile - op.first_tile) * FW_NEST_TILE;
    const FwSeg &Cs = g.segs[op.child_seg];
    const uint32_t ccap = Cs.capacity;
    const uint32_t child_head = op.child_range ? fw_range_head(op.child_head, g.rold[cidx], ccap) : op.child_head;
    uint32_t op_total = 0;  // non-zero only in the op's last active tile
    if (base < n_par) {
        const FwEmit &e = g.emits[op.emit];
        char *pb = op.parent_buf;
        const uint32_t PC = op.parent_cap;
        uint32_t n[NR], inc[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint32_t idx = base + r * FW_BLOCK + tid;
            n[r] = 0;
            if (idx < n_par) {
                float next;
                const uint64_t cnt = fw_emission_count(p_age[r], p_lea[r], p_life[r], op.n_start, op.n_end, op.n_count, &next);
                n[r] = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt;  // core.rs:490-498
                fw_st1(pb + FW_OFF_L(PC, op.parent_lplane), fw_ring_slot(parent_head, idx, PC), next);  // other_particle.last_emitted_age[i] = next (core.rs:500)
            }
            uint32_t x = n[r];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(x, o, 64);
                if (lane >= (uint32_t)o) x = (x + u < x) ? 0xFFFFFFFFu : x + u;  // saturating
            }
            inc[r] = x;
            if (lane == 63) s_w[r][wave] = x;
        }
        __syncthreads();
        unsigned long long tot64 = 0;
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int w = 0; w < NW; w++) tot64 += s_w[r][w];
        uint32_t tile_total = tot64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot64;
        // The op's last active tile commits (child `appended`, RNG serial) as soon as its look-back has seen the status
        // word of every earlier tile -- so a tile must have READ those counters (n_par, cbase, serial0) before it
        // publishes.  Make that a data dependency, not an accident of instruction scheduling: the published value
        // passes through an opaque instruction that also consumes the three loaded values (they are in registers, i.e.
        // the loads have returned, when it executes; the status stores below consume its result).
        {
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(cbase), d1 = __builtin_amdgcn_readfirstlane((uint32_t)serial0),
                           d2 = __builtin_amdgcn_readfirstlane((uint32_t)(serial0 >> 32)), d3 = __builtin_amdgcn_readfirstlane(n_par);
            asm volatile("; fw_k_nest: counters read before the tile publishes" : "+v"(tile_total) : "s"(d0), "s"(d1), "s"(d2), "s"(d3));
        }
        // ---- exclusive prefix over the earlier parent tiles of this op
        const bool lb_needed = tile > op.first_tile;
        if (lb_needed && tid == 0)
            __hip_atomic_store(&g.nest_status[tile], fw_pack_status(tag, FW_ST_AGG, tile_total), RLX, AGENT);
        uint32_t excl = 0;
        if (lb_needed && !FW_DBG(dbg, 64u)) {  // (FW_DEBUG 64: profiling only, no look-back)
            bool timed_out = false;
            excl = fw_lookback<FW_BLOCK, NW, LBW>(g.nest_status, op.first_tile, tile, tag, spin_limit * 64u + 1024u, s_lb,
                                                   &timed_out);
            // (no recount is possible here: the earlier tiles have already advanced their parents' last_emitted_age.
            // Workgroups are dispatched in index order, so every predecessor is resident or done: the wait is bounded.)
            if (timed_out && tid == 0) fw_raise(g, 6u, 0xFFFFFFFFu, tile);
        }
        const unsigned long long incl64 = (unsigned long long)excl + tile_total;
        const uint32_t incl = incl64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)incl64;
        if (tid == 0) __hip_atomic_store(&g.nest_status[tile], fw_pack_status(tag, FW_ST_INCL, incl), RLX, AGENT);
        if (base + FW_NEST_TILE >= n_par) op_total = incl;  // the op's last active tile knows the total
        // ---- children, wave-cooperatively (parent-major order)
        uint32_t run = excl;  // children of the tile before (round r, wave 0)
#pragma unroll
        for (int r = 0; r < NR; r++) {
            uint32_t woff = run;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if ((uint32_t)w < wave) woff += s_w[r][w];
                run += s_w[r][w];
            }
            const uint32_t tw = FW_DBG(dbg, 32u) ? 0u : s_w[r][wave];  // wave-uniform (FW_DEBUG 32: profiling only, no children)
            if (tw == 0) continue;
            const uint32_t idx = base + r * FW_BLOCK + tid;
            s_inc[wave][lane] = inc[r];
            if (n[r] != 0) {
                const uint32_t ps = fw_ring_slot(parent_head, idx, PC);
                s_par[wave][0][lane] = fw_ld4(pb + FW_OFF_Q0(PC), ps);
                s_par[wave][1][lane] = fw_ld4(pb + FW_OFF_Q1(PC), ps);
                s_par[wave][2][lane] = op.parent_nospin ? make_float4(op.parent_rot[0], op.parent_rot[1], op.parent_rot[2], op.parent_rot[3])
                                                        : fw_ld4(pb + FW_OFF_Q2(PC), ps);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t c0 = 0; c0 < tw; c0 += 64u) {
                const uint32_t c = c0 + lane;
                if (c < tw) {
                    uint32_t lo = 0, hi = 63;  // first lane whose inclusive prefix exceeds c
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_inc[wave][mid] > c) hi = mid;
                        else lo = mid + 1;
                    }
                    const unsigned long long j = (unsigned long long)woff + c;  // child index within the op
                    const unsigned long long slot = (unsigned long long)cbase + j;
                    if (slot < ccap) {
                        const float4 pq0 = s_par[wave][0][lo], pq1 = s_par[wave][1][lo], pq2 = s_par[wave][2][lo];
                        FwSpawnOut o = fw_spawn_one(e, g.seed, serial0 + j, fw_v3{pq0.x, pq0.y, pq0.z},
                                                    fw_q4{pq2.x, pq2.y, pq2.z, pq2.w}, fw_v3{pq1.x, pq1.y, pq1.z}, op.speed,
                                                    op.scale);
                        fw_store_new(g, Cs, Cs.buf[parity], fw_ring_slot(child_head, (uint32_t)slot, ccap), o);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the wave's LDS rows are reused by the next round
        }
    }
    // ---- commit the op's totals (children appended to the child segment, RNG serial of the entry).
    // Normally by the op's last ACTIVE tile: once its look-back is done every earlier tile has started, i.e. has read
    // the counters that fix its parent bound and child base, and the later tiles hold no parents and read nothing that
    // changes.  Only when particles emit onto their OWN type (parent segment == child segment) would a late workgroup
    // see the new children as parents: such ops commit through a ticket instead -- the last workgroup to FINISH does it.
    // The ticket word carries the total ({finished workgroups : 32 | children : 32}, one relaxed 64-bit atomic each): no
    // fence (a device-scope release writes back the XCD's whole L2; 800 of them cost 10 us), nothing to order.  (All
    // workgroups hitting one word serialise at the memory side -- 10 ns each -- which is why this is not the normal path.)
    const bool self_nested = op.parent_seg == op.child_seg;
    auto commit = [&](unsigned long long total) {
        const unsigned long long room = cbase < ccap ? (unsigned long long)(ccap - cbase) : 0ull;
        unsigned long long take = total;
        if (take > room) {
            take = room;
            atomicOr(g.err, FW_ERR_CAPACITY);
        }
        g.emit_serial[op.emit_slot] = serial0 + total;
        g.appended[cidx] += (uint32_t)take;
    };
    if (!self_nested) {
        if (tid == 0 && base < n_par && base + FW_NEST_TILE >= n_par) commit(op_total);
    } else {
        __syncthreads();
        if (tid == 0) {
            const unsigned long long mine = (1ull << 32) | (unsigned long long)op_total;
            const unsigned long long seen = __hip_atomic_fetch_add(&g.nest_ticket[oi], mine, RLX, AGENT) + mine;
            if ((uint32_t)(seen >> 32) == op.n_tiles) {
                commit((uint32_t)seen);
                __hip_atomic_store(&g.nest_ticket[oi], 0ull, RLX, AGENT);  // ready for the next launch
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// readback / upload / render hand-off helpers
// ---------------------------------------------------------------------------------

// SoA planes -> fw_particle records (26 x 4 B)
// (rot: the rotation of every particle of a type that cannot turn -- FW_TYPE_NOSPIN, its plane is not maintained -- or null)
__global__ void fw_k_gather(const char *buf, uint32_t C, uint32_t head, uint32_t n, int32_t pbr, float *out, bool nospin, float4 rot,
                            uint32_t life_plane, float life_const, const FwType *derived, const float *keys) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const uint32_t i = fw_ring_slot(head, li, C);
    const float4 q0 = fw_ld4(buf + FW_OFF_Q0(C), i), q1 = fw_ld4(buf + FW_OFF_Q1(C), i),
                 q2 = nospin ? rot : fw_ld4(buf + FW_OFF_Q2(C), i),
                 // (cannot turn: angular velocity 0; the lifetime from its plane, or -- a ring -- the type's one value)
                 q3 = !nospin ? fw_ld4(buf + FW_OFF_Q3(C), i)
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
    fw_st4(buf + FW_OFF_Q1(C), i, make_float4(r[3], r[4], r[5], r[13]));
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
__global__ void fw_k_rederive(char *buf, uint32_t C, const FwType *T, const float *keys, bool nospin, uint32_t life_plane, float life_const) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const float4 q0 = fw_ld4(buf + FW_OFF_Q0(C), i);
    const float life = !nospin ? fw_ld4(buf + FW_OFF_Q3(C), i).w
                               : (life_plane != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(C, life_plane), i) : life_const);
    float4 bc, em;
    float sc;
    fw_derived_values(*T, keys + T->keys_off, q0.w, life, fw_ld4(buf + FW_OFF_Q1(C), i).w, &bc, &em, &sc);
    fw_st4(buf + FW_OFF_Q5(C), i, bc), fw_st4(buf + FW_OFF_Q6(C), i, em), fw_st1(buf + FW_OFF_S4(C), i, sc);
}
hipError_t fw_launch_rederive(hipStream_t s, char *buf, uint32_t capacity, const FwType *d_type, const float *d_keys, bool nospin,
                              uint32_t life_plane, float life_const) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_rederive, dim3((capacity + 255) / 256), dim3(256), 0, s, buf, capacity, d_type, d_keys, nospin, life_plane, life_const);
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
__global__ void fw_k_restore_q3(char *buf0, char *buf1, uint32_t C, uint32_t life_plane, float life_const) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const bool pl = life_plane != 0xFFFFFFFFu;
    fw_st4(buf0 + FW_OFF_Q3(C), i, make_float4(0.f, 0.f, 0.f, pl ? fw_ld1(buf0 + FW_OFF_L(C, life_plane), i) : life_const));
    if (buf1) fw_st4(buf1 + FW_OFF_Q3(C), i, make_float4(0.f, 0.f, 0.f, pl ? fw_ld1(buf1 + FW_OFF_L(C, life_plane), i) : life_const));
}

// ParticleInstance packing (reference src/render.rs:95-115): {pos, scale, rot, base, emissive}
// SoA planes -> ParticleInstance records (render.rs:95-115).  Loads are plane-wise coalesced; the 64-byte records are
// transposed through LDS so that every store instruction of a wave writes 1 KiB of consecutive bytes (a lane writing
// its own record with four float4 stores would touch 64 lines a quarter at a time).
__global__ __launch_bounds__(256) void fw_k_pack(const char *buf, uint32_t C, uint32_t head, const uint32_t *d_count,
                                                 uint32_t n_upper, float4 *out, bool nospin, float4 rot, const uint32_t *d_rold,
                                                 const FwType *derived, const float *keys, uint32_t life_plane, float life_const) {
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
            const float life = !nospin ? fw_ld4(buf + FW_OFF_Q3(C), i).w
                                       : (life_plane != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(C, life_plane), i) : life_const);
            fw_derived_values(*derived, keys + derived->keys_off, q0.w, life, fw_ld4(buf + FW_OFF_Q1(C), i).w, &q5, &q6, &sc);
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
                const float life = !(TT.flags & FW_TYPE_NOSPIN) ? fw_ld4(buf + FW_OFF_Q3(S.capacity), i).w
                                   : (L.life_plane[k] != 0xFFFFFFFFu ? fw_ld1(buf + FW_OFF_L(S.capacity, L.life_plane[k]), i) : L.life_const[k]);
                const float *keys = g.keys + TT.keys_off;
                sc = fw_ld4(buf + FW_OFF_Q1(S.capacity), i).w * fw_curve_sample(TT.sc_kind, TT.sc_n, keys, keys + TT.o_sc_v, q0.w / life);
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

// ---------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------

hipError_t fw_launch_spawn(hipStream_t s, const FwGlobals &g, const FwOp *d_ops, const FwOp *h_ops, uint32_t n_ops,
                           uint32_t total_blocks, uint32_t parity) {
    if (!n_ops || !total_blocks) return hipSuccess;
    static FwInlineOps io;  // (calls on a context are serialised by the caller; the launch copies its arguments)
    if (!d_ops)
        for (uint32_t i = 0; i < n_ops && i < FW_INLINE_OPS; i++) io.ops[i] = h_ops[i];
    hipLaunchKernelGGL(fw_k_spawn, dim3(total_blocks), dim3(FW_BLOCK), 0, s, g, io, d_ops, n_ops, parity);
    return hipGetLastError();
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

template <int R, bool INST, bool SUMS>
static void fw_launch_update_r(hipStream_t s, const FwGlobals &g, const FwUpdateArgs &a, const FwInlineOps &io,
                               int spawn_form, int mode, hipEvent_t e0, hipEvent_t e1) {
    const dim3 grid(a.total_tiles), block(FW_TILE / R);
    if (mode == FW_MODE_SPLIT_COLL) {  // frames with colliding particle types: count (with collisions), scan, update
        FW_LAUNCH_T(fw_k_count, grid, dim3(FW_BLOCK), s, e0, (hipEvent_t) nullptr, g, a);
        hipLaunchKernelGGL(fw_k_scan, dim3(a.n_seg), dim3(FW_BLOCK), 0, s, g, a);
        FW_LAUNCH_T(fw_k_update_coll, grid, dim3(FW_BLOCK), s, (hipEvent_t) nullptr, e1, g, a);
    } else if (mode == FW_MODE_SPLIT) {  // debugging / A-B mode: three launches, no inter-workgroup traffic
        FW_LAUNCH_T(fw_k_count, grid, dim3(FW_BLOCK), s, e0, (hipEvent_t) nullptr, g, a);
        hipLaunchKernelGGL(fw_k_scan, dim3(a.n_seg), dim3(FW_BLOCK), 0, s, g, a);
        FW_LAUNCH_T((fw_k_update<false, FW_SPAWN_NONE, R, INST, SUMS>), grid, block, s, (hipEvent_t) nullptr, e1, g, a, io);
    } else if (a.use_stream && a.fc_in && a.fc_out) {  // forecast frame: streaming schedule
        const bool lone = a.n_seg == 1u && a.seg0_ib != nullptr;  // a single segment: its record rides in the arguments
        if (spawn_form == FW_SPAWN_INLINE && lone)
            FW_LAUNCH_T((fw_k_update_stream<FW_SPAWN_INLINE, INST, SUMS, true>), grid, dim3(FW_BLOCK), s, e0, e1, g, a, io);
        else if (spawn_form == FW_SPAWN_INLINE)
            FW_LAUNCH_T((fw_k_update_stream<FW_SPAWN_INLINE, INST, SUMS, false>), grid, dim3(FW_BLOCK), s, e0, e1, g, a, io);
        else if (spawn_form == FW_SPAWN_TABLE)
            FW_LAUNCH_T((fw_k_update_stream<FW_SPAWN_TABLE, INST, SUMS, false>), grid, dim3(FW_BLOCK), s, e0, e1, g, a, io);
        else if (lone)
            FW_LAUNCH_T((fw_k_update_stream<FW_SPAWN_NONE, INST, SUMS, true>), grid, dim3(FW_BLOCK), s, e0, e1, g, a, io);
        else
            FW_LAUNCH_T((fw_k_update_stream<FW_SPAWN_NONE, INST, SUMS, false>), grid, dim3(FW_BLOCK), s, e0, e1, g, a, io);
    } else if (spawn_form == FW_SPAWN_INLINE) {
        FW_LAUNCH_T((fw_k_update<true, FW_SPAWN_INLINE, R, INST, SUMS>), grid, block, s, e0, e1, g, a, io);
    } else if (spawn_form == FW_SPAWN_TABLE) {
        FW_LAUNCH_T((fw_k_update<true, FW_SPAWN_TABLE, R, INST, SUMS>), grid, block, s, e0, e1, g, a, io);
    } else {
        FW_LAUNCH_T((fw_k_update<true, FW_SPAWN_NONE, R, INST, SUMS>), grid, block, s, e0, e1, g, a, io);
    }
}

hipError_t fw_launch_update(hipStream_t s, const FwGlobals &g, const FwUpdateArgs &a, const FwInlineOps *inl,
                            int spawn_form, int mode, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (!a.total_tiles) {
        if (ev_start) (void)hipEventRecord(ev_start, s);
        if (ev_stop) (void)hipEventRecord(ev_stop, s);
        return hipGetLastError();
    }
    static const FwInlineOps none{};
    const FwInlineOps &io = inl ? *inl : none;
    if (mode != FW_MODE_FUSED && spawn_form != FW_SPAWN_NONE) return hipErrorInvalidValue;
    // 256 threads x 4 rounds is the measured optimum; 512 x 2 and 1024 x 1 were 25-30 % slower
    // kernels that also write attached ParticleInstance buffers are separate instantiations: the plain ones keep
    // their register budget
    // likewise the two forecast formats: plain per-tile entries when every segment is small, atomic sums otherwise
    if (a.any_inst && a.fc_sums)
        fw_launch_update_r<FW_ROUNDS, true, true>(s, g, a, io, spawn_form, mode, ev_start, ev_stop);
    else if (a.any_inst)
        fw_launch_update_r<FW_ROUNDS, true, false>(s, g, a, io, spawn_form, mode, ev_start, ev_stop);
    else if (a.fc_sums)
        fw_launch_update_r<FW_ROUNDS, false, true>(s, g, a, io, spawn_form, mode, ev_start, ev_stop);
    else
        fw_launch_update_r<FW_ROUNDS, false, false>(s, g, a, io, spawn_form, mode, ev_start, ev_stop);
    return hipGetLastError();
}

hipError_t fw_launch_update_fifo(hipStream_t s, const FwGlobals &g, const FwFifoArgs &a, const FwInlineOps &inl,
                                 uint32_t total_tiles, int nt, hipEvent_t e0, hipEvent_t e1) {
    if (!total_tiles || !a.n_segs) return hipErrorInvalidValue;
    const dim3 grid(total_tiles), block(FW_BLOCK);
    if (a.any_coll) {  // some ring of the launch collides (FwCollArm): generic write mask, plain or fully non-temporal
        if (a.any_inst && nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 2, true>), grid, block, s, e0, e1, g, a, inl);
        else if (a.any_inst)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 0, true>), grid, block, s, e0, e1, g, a, inl);
        else if (nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 2, true>), grid, block, s, e0, e1, g, a, inl);
        else
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 0, true>), grid, block, s, e0, e1, g, a, inl);
        return hipGetLastError();
    }
    if (nt) {  // non-temporal forms (fw_ld4w): the generic write mask only -- beyond the Infinity Cache the compile-time one buys nothing
        if (a.any_inst && nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 2>), grid, block, s, e0, e1, g, a, inl);
        else if (a.any_inst)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 1>), grid, block, s, e0, e1, g, a, inl);
        else if (nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 2>), grid, block, s, e0, e1, g, a, inl);
        else
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 1>), grid, block, s, e0, e1, g, a, inl);
        return hipGetLastError();
    }
#define FW_FIFO_CASE(wm)                                                                   \
    case wm:                                                                               \
        if (a.any_inst)                                                                    \
            FW_LAUNCH_T((fw_k_update_fifo<true, wm>), grid, block, s, e0, e1, g, a, inl);  \
        else                                                                               \
            FW_LAUNCH_T((fw_k_update_fifo<false, wm>), grid, block, s, e0, e1, g, a, inl); \
        break;
    switch (a.write_mask) {
        FW_FIFO_CASE(0) FW_FIFO_CASE(1) FW_FIFO_CASE(2) FW_FIFO_CASE(3) FW_FIFO_CASE(4) FW_FIFO_CASE(5) FW_FIFO_CASE(6)
        FW_FIFO_CASE(7)
        default:
            if (a.any_inst)
                FW_LAUNCH_T((fw_k_update_fifo<true, -1>), grid, block, s, e0, e1, g, a, inl);
            else
                FW_LAUNCH_T((fw_k_update_fifo<false, -1>), grid, block, s, e0, e1, g, a, inl);
    }
#undef FW_FIFO_CASE
    return hipGetLastError();
}

template <int NT>
static void fw_launch_update_range_t(hipStream_t s, const FwGlobals &g, const FwRangeArgs &a, bool all_nospin, hipEvent_t e0,
                                     hipEvent_t e1) {
    const dim3 grid(a.total_tiles), block(FW_BLOCK);
    if constexpr (NT != 1) {
        if (a.any_coll) {  // some range ring of the launch collides (FwCollArm)
            if (a.any_inst) {
                if (all_nospin) FW_LAUNCH_T((fw_k_update_range<true, true, NT, true>), grid, block, s, e0, e1, g, a);
                else FW_LAUNCH_T((fw_k_update_range<false, true, NT, true>), grid, block, s, e0, e1, g, a);
            } else if (all_nospin) {
                FW_LAUNCH_T((fw_k_update_range<true, false, NT, true>), grid, block, s, e0, e1, g, a);
            } else {
                FW_LAUNCH_T((fw_k_update_range<false, false, NT, true>), grid, block, s, e0, e1, g, a);
            }
            return;
        }
    }
    if (a.any_inst) {
        if (all_nospin)
            FW_LAUNCH_T((fw_k_update_range<true, true, NT>), grid, block, s, e0, e1, g, a);
        else
            FW_LAUNCH_T((fw_k_update_range<false, true, NT>), grid, block, s, e0, e1, g, a);
    } else if (all_nospin) {
        FW_LAUNCH_T((fw_k_update_range<true, false, NT>), grid, block, s, e0, e1, g, a);
    } else {
        FW_LAUNCH_T((fw_k_update_range<false, false, NT>), grid, block, s, e0, e1, g, a);
    }
}

hipError_t fw_launch_update_range(hipStream_t s, const FwGlobals &g, const FwRangeArgs &a, bool all_nospin, int nt,
                                  hipEvent_t e0, hipEvent_t e1) {
    if (!a.total_tiles) return hipErrorInvalidValue;
    if (a.any_coll && nt == 1) nt = 0;  // (the collision instantiations exist plain and fully non-temporal)
    if (nt == 2) fw_launch_update_range_t<2>(s, g, a, all_nospin, e0, e1);
    else if (nt == 1) fw_launch_update_range_t<1>(s, g, a, all_nospin, e0, e1);
    else fw_launch_update_range_t<0>(s, g, a, all_nospin, e0, e1);
    return hipGetLastError();
}

hipError_t fw_launch_nested(hipStream_t s, const FwGlobals &g, const FwNestOp *d_ops, const FwNestOp *h_ops, uint32_t n_ops,
                            uint32_t total_tiles, uint32_t parity, uint32_t tag, uint32_t spin_limit, uint32_t dbg) {
    if (!n_ops || !total_tiles) return hipSuccess;
    static FwNestInline io;
    if (!d_ops)
        for (uint32_t i = 0; i < n_ops && i < FW_INLINE_OPS; i++) io.ops[i] = h_ops[i];
    hipLaunchKernelGGL(fw_k_nest, dim3(total_tiles), dim3(FW_BLOCK), 0, s, g, io, d_ops, n_ops, parity, tag, spin_limit, dbg);
    return hipGetLastError();
}

hipError_t fw_launch_gather(hipStream_t s, const char *buf, uint32_t capacity, uint32_t head, uint32_t n, int32_t pbr, void *d_out,
                            const float *const_rot, uint32_t life_plane, float life_const, const FwType *derived, const float *keys) {
    if (!n) return hipSuccess;
    const float4 rot = const_rot ? make_float4(const_rot[0], const_rot[1], const_rot[2], const_rot[3]) : make_float4(0.f, 0.f, 0.f, 1.f);
    hipLaunchKernelGGL(fw_k_gather, dim3((n + 255) / 256), dim3(256), 0, s, buf, capacity, head, n, pbr, (float *)d_out,
                       const_rot != nullptr, rot, life_plane, life_const, derived, keys);
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
hipError_t fw_launch_restore_q3(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, uint32_t life_plane, float life_const) {
    if (!capacity) return hipSuccess;
    hipLaunchKernelGGL(fw_k_restore_q3, dim3((capacity + 255) / 256), dim3(256), 0, s, buf0, buf1, capacity, life_plane, life_const);
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
                                    const float *keys, uint32_t life_plane, float life_const) {
    if (!n_upper) return hipSuccess;
    uint32_t blocks = (n_upper + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const float4 rot = const_rot ? make_float4(const_rot[0], const_rot[1], const_rot[2], const_rot[3]) : make_float4(0.f, 0.f, 0.f, 1.f);
    hipLaunchKernelGGL(fw_k_pack, dim3(blocks), dim3(256), 0, s, buf, capacity, head, d_count, n_upper, (float4 *)d_out,
                       const_rot != nullptr, rot, d_rold, derived, keys, life_plane, life_const);
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
