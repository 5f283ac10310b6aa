// fw_device.h -- device-resident data layout of the firework HIP backend (gfx950).
//
// Particle state lives in HBM as a structure-of-arrays mirror of the reference's
// ParticleData (reference src/core.rs:305-321), one SEGMENT per (spawner,
// particle type) pair -- the reference's `particles: Vec<Vec<ParticleData>>`
// (src/core.rs:274).  A segment owns two buffers (ping/pong); the update kernel
// reads buf[parity] and writes the stably compacted survivors to buf[parity^1].
//
// One buffer of capacity C particles holds these planes (C is a multiple of 256,
// every plane 16-byte aligned, one particle per lane => every load/store is a
// full-width dwordx4 and the compacted store position of a particle is always
// 16-byte aligned):
//
//   Q0  float4 {pos.x, pos.y, pos.z, age}            offset   0*C   read+write
//   Q1  float4 {vel.x, vel.y, vel.z, initial_scale}  offset  16*C   read+write
//   Q2  float4 {rot.x, rot.y, rot.z, rot.w}          offset  32*C   read+write
//   Q3  float4 {angvel.x, .y, .z, lifetime}          offset  48*C   read+write
//       (round 6, RING segments -- FwSeg::cpl: the Q1 and Q3 regions hold their four components as four planes of C floats each,
//       4*C bytes apart: initial_scale and lifetime never change, an in-place update moves three dwords per lane, not a dwordx4)
//   Q5  float4 base_color rgba                       offset  64*C   write only
//   Q6  float4 emissive_color rgba                   offset  80*C   write only
//   S4  float  scale                                 offset  96*C   write only
//   Lk  float  last_emitted_age[k]                   offset (100+4k)*C   only for
//       emission entries that are Nested on this type (src/core.rs:320,467,493-500)
//
// Reads 64 B + writes 100 B per particle per frame = 164 B (the "156 B + 8 B
// ping-pong" variant of SURVEY.md §8d); +8 B per Lk plane.  `pbr` is a per-type
// constant (src/core.rs:462) and is not stored per particle.
#pragma once
#include <stdint.h>

#define FW_BLOCK 256          // threads per workgroup = 4 wave64
#define FW_ROUNDS 4           // particles per thread per tile
#define FW_TILE (FW_BLOCK * FW_ROUNDS)
#define FW_VTILE FW_BLOCK     // tile size of the new-particle region (fw_k_update)
#define FW_NEST_TILE FW_BLOCK  // parents per workgroup of fw_k_nest: one per lane (spawning a child is ~2.5k instructions:
                              // small tiles = several waves per SIMD to hide them; 1024-parent tiles ran one wave per SIMD)
#define FW_VFRONT 256u        // at most this many new-particle tiles are dispatched first
#define FW_KEYS_MAX 400       // floats of curve keys staged in LDS per type
#define FW_DEV_MAX_EMISSIONS 8

#define FW_OFF_Q0(C) ((size_t)0)
#define FW_OFF_Q1(C) ((size_t)16 * (C))
#define FW_OFF_Q2(C) ((size_t)32 * (C))
#define FW_OFF_Q3(C) ((size_t)48 * (C))
#define FW_OFF_Q5(C) ((size_t)64 * (C))
#define FW_OFF_Q6(C) ((size_t)80 * (C))
#define FW_OFF_S4(C) ((size_t)96 * (C))
#define FW_OFF_L(C, k) ((size_t)(100 + 4 * (k)) * (C))
#define FW_BUF_BYTES(C, nl) ((size_t)(100 + 4 * (nl)) * (C))

// one (spawner, particle type) segment
struct alignas(16) FwSeg {
    char *buf[2];
    char *destroyed;      // fw_particle AoS records of the last step, or null
    uint32_t capacity;    // particles per buffer, multiple of FW_BLOCK
    uint32_t type_idx;    // -> FwType
    uint32_t n_lplanes;   // number of Lk planes
    uint32_t inst_cap;    // records that fit in `inst`
    char *inst;           // attached ParticleInstance output (render hand-off fused into the update), or null
    // A type other particles' entries emit from, spawned INSIDE its ring's update kernel even in frames with a Nested pass
    // (fw_engine.h: SegHost::virt_parent): the FwEmit of the Nested entry that owns last_emitted_age plane k (k < 2), or
    // 0xFFFFFFFF.  The pass of the frame would have visited the new particle -- age 0, last_emitted_age f32::MIN -- emitted
    // nothing (offsets >= 0) and left `next` in the plane (core.rs:488-500): the spawning lane computes that value itself
    // (fw_init_last_emitted) and fw_k_spawn has nothing to materialise.
    uint32_t lplane_emit[2];
    // 1: a RING segment (FIFO / range ring) -- its Q1 and Q3 regions are component planes (fw_dev.h: FW_CP); 0: float4 planes
    uint32_t cpl;
    uint32_t pad_[3];
};

// per particle type constants (ParticleSettings, reference src/core.rs:99-142)
struct alignas(16) FwType {
    float acc[3];
    float lin_drag;
    float angacc[3];
    float ang_drag;
    int32_t sc_kind, sc_n;   // scale_curve
    int32_t bc_kind, bc_n;   // base_color
    int32_t em_kind, em_n;   // emissive_color
    int32_t pbr, report_destroyed;
    // key pool layout (floats, each sub-array padded to a multiple of 4):
    //  [sc_times | sc_vals | bc_times | bc_rgba | em_times | em_rgba]
    uint32_t keys_off, keys_len;
    uint32_t o_sc_v, o_bc_t, o_bc_v, o_em_t, o_em_v;
    // FW_TYPE_NOSPIN: no particle of the type can ever turn -- every emission entry that feeds it has an angular-velocity
    // magnitude range of exactly {0, 0} and the same initial_rotation, and the type's angular_acceleration is zero.  Then
    // angular_velocity stays (+-0, +-0, +-0) (core.rs:648-650) and rotation = from_scaled_axis(0) * rotation stays the
    // entries' initial_rotation (core.rs:645-647; numerically: signs of zero components aside), for every particle, for
    // ever: the update neither reads nor writes the rotation plane (32 of the 164 bytes a compacting update moves per
    // particle), readers get const_rot instead.  Cleared for good when the caller rewrites the type's particles or a
    // non-finite dt is stepped (the plane is filled with const_rot first).
    uint32_t flags;
    float const_rot[4];
};
#define FW_TYPE_NOSPIN 1u
// FW_TYPE_DERIVED: the type has an attached ParticleInstance buffer (fw_spawner_attach_instances), which receives scale,
// base colour and emissive colour of every survivor in its 64-byte record -- the planes S4 / Q5 / Q6 would only hold a
// second copy (36 of the 228 bytes an update + records moved per particle).  They are pure functions of (age, lifetime,
// initial_scale): the update does not store them, and whoever reads them (fw_k_gather, fw_k_pack, the AABB query, the
// destroyed records) evaluates them again from the stored age -- the same functions on the same inputs the update used,
// bit for bit (render.rs:95-115, 403).  Cleared, after fw_k_rederive has filled the planes, when the buffer is detached.
#define FW_TYPE_DERIVED 2u
// (the update kernels of the general path learn the flag before the type record arrives -- their loads depend on it --
// from bit 31 of the type index in the tile descriptor / FwUpdateArgs::seg0_type / FwFifoSeg::type_idx)
#define FW_TYPE_IDX_NOSPIN 0x80000000u
// (range descriptors, round 6: bit 30 = FW_TYPE_DERIVED and no attached instance buffer -- nobody in a YOUNG tile needs the particle's
// lifetime except the consistency check, which the tile at the boundary to the old part runs for everybody: ages never increase along
// the list, so if the oldest young particles cannot die nobody behind them can.  The other young tiles do not load it: 60 -> 56 B)
#define FW_TYPE_IDX_NOLIFE 0x40000000u
#define FW_TYPE_IDX_MASK 0x3FFFFFFFu
// collision_settings of a particle type (core.rs:137-138, 240-248), in a table of its own next to FwType: only the
// collision kernels read it, the streaming kernels' per-type record (and their scalar-register budget) stays as it was
struct alignas(16) FwTypeColl {
    uint32_t coll_flags, coll_mask;  // bit 0 = Some(..), bit 1 = destroy_on_collision
    float coll_restitution, coll_friction;
};
#define FW_COLL_ENABLED 1u
#define FW_COLL_DESTROY 2u

// static settings of one emission entry (EmissionSettings, src/core.rs:144-162)
// plus the spawn-relevant ranges of its particle type
struct alignas(16) FwEmit {
    int32_t shape_kind;
    float shape_radius;
    uint32_t uid;             // spawner uid (RNG key word 1)
    uint32_t emission_index;  // RNG counter word 2
    float shape_arc[4];       // Quat::from_rotation_arc(Y, normal)
    float v_mag_min, v_mag_max, v_spread;
    int32_t inherit;
    float v_dir[4];
    float v_arc[4];           // from_rotation_arc(Y, direction)
    float w_mag_min, w_mag_max, w_spread;
    uint32_t type_idx;
    float w_dir[4];
    float w_arc[4];
    float init_rot[4];
    float radial_min, radial_max, iscale_min, iscale_max;
    float life_min, life_max;
    // Nested pacing (src/core.rs:474-500)
    float n_count, n_start, n_end;
    uint32_t n_lplane;        // which Lk plane of the PARENT type belongs to this entry
    uint32_t pad0[2];
};

// one Global spawn operation of the current frame (src/core.rs:395-470)
struct alignas(16) FwOp {
    uint32_t seg;         // destination segment
    uint32_t emit;        // -> FwEmit
    uint32_t n;           // particles_to_spawn
    uint32_t rel_base;    // slots already taken this frame by earlier Global ops on `seg`
    uint64_t serial_base; // RNG serial of the first particle
    uint32_t first_block; // first workgroup of this op in the spawn launch
    uint32_t head;        // ring head of `seg` when it is a FIFO ring (slot of its particle 0), else 0
    float origin_pos[4];
    float origin_rot[4];
    float parent_vel[4];
    float speed, scale;   // EffectModifier (src/core.rs:323-327)
    uint32_t range_ring;  // 1: `seg` is a RANGE ring -- `head` is the slot of its first young particle, particle 0 sits
                          // FwGlobals::rold slots before it (fw_ring_head)
    uint32_t pad1;
};

// one Nested emission operation of the current frame (src/core.rs:471-546)
struct alignas(16) FwNestOp {
    uint32_t parent_seg, child_seg;
    uint32_t emit;
    uint32_t first_tile;  // first parent tile of this op in the nested launches
    uint32_t n_tiles;     // parent tiles launched (upper bound)
    uint32_t emit_slot;   // index into the device serial counters
    float speed, scale;
    // what a tile needs to ADDRESS its parents, so that their loads depend on the op record only (kernel arguments
    // for short op lists) and go out together with the counter loads: the launch is latency-bound (a few hundred
    // small workgroups), every dependent memory level costs ~1 us
    char *parent_buf;        // parent segment buffer of this frame's parity
    uint32_t parent_cap;     // its capacity (plane stride)
    uint32_t parent_lplane;  // which last_emitted_age plane of the parent type belongs to this entry
    float n_count, n_start, n_end;  // CountOverDuration of the entry (core.rs:474-481)
    uint32_t parent_head;    // ring heads of the two segments (FIFO rings; 0 otherwise): particle i sits in slot
    uint32_t child_head;     // (head + i) mod capacity
    uint32_t parent_nospin;  // bit 0: the parent type cannot turn (FW_TYPE_NOSPIN): its rotation is parent_rot, not in the plane,
                             // (bit 1: the parent segment is a ring -- its Q1 / Q3 are component planes, FwSeg::cpl)
    uint32_t parent_life_plane;  // ... and its lifetimes sit in this 4-byte plane (FW_OFF_L index), not in Q3;
    float parent_life_const;     // 0xFFFFFFFF: the parent is a ring, all its particles have this lifetime
    float parent_rot[4];
    uint32_t parent_range, child_range;  // 1: the segment is a RANGE ring -- its `head` above is the slot of its first young
                                         // particle, particle 0 sits FwGlobals::rold slots before it (fw_ring_head)
    uint32_t ticket_base;    // value of FwGlobals::nest_start[emit_slot] when the launch starts (fw_kernels.h: START tickets)
    uint32_t pad2;
};

// decoupled look-back status word: {epoch:30 | state:2 | value:32}
#define FW_ST_AGG 1u
#define FW_ST_INCL 2u

// device-side error flags (sticky until read)
#define FW_ERR_CAPACITY 1u
#define FW_ERR_LOOKBACK_TIMEOUT 2u
#define FW_ERR_FORECAST 4u          // a forecast entry carried the wrong frame tag (internal error)

// survivor forecast sums (fw_dev.h), in 64-bit words: stride between consecutive group counters P2 (one 64-byte
// line each: they are hot) and the segment size up to which tiles sum P directly instead of using P2
#define FW_FC_S2_STRIDE 8u
#define FW_FC_DIRECT 2048u

