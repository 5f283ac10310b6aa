// fw_kernels.h -- launch interface between the host engine (fw_engine.h, fw_engine_*.cpp) and the
// gfx950 kernels (fw_k_general.hip, fw_k_rings.hip, fw_k_nested.hip, fw_k_aux.hip; shared device helpers: fw_dev.h).
#pragma once
#include <hip/hip_runtime.h>

#include "fw_device.h"
#include "fw_collide.h"

// The product build (the default `make`) carries no experiment surface: the FW_DEBUG profiling modes (kernel ablations, in-kernel
// timestamps) and the A/B switches of measured-and-rejected variants exist only in the `make ab` build (-DFW_AB,
// libfirework_hip_ab.so, loaded by the tools through FW_LIB_PATH).  FW_DBG(x, bit) is a compile-time 0 in the product: the
// branches fold away (an UNUSED timestamp branch was measured at 6 % of one GPU's share of configs[4], profiles/r03).
#ifdef FW_AB
#define FW_DBG(x, bit) (((x) & (bit)) != 0u)
#else
#define FW_DBG(x, bit) (false)
#endif

// pointers to the context's persistent device state (passed by value as a kernel argument)
#define FW_STAT_SLOTS 32
struct FwGlobals {
    FwSeg *segs;
    FwType *types;
    FwTypeColl *type_coll;  // [like types] collision settings per particle type
    float *keys;
    FwEmit *emits;
    uint32_t *count;     // [2][max_seg]  live particles per segment, by buffer parity
    uint32_t *spawned;   // [2][max_seg]  Global spawns appended this frame
    uint32_t *appended;  // [2][max_seg]  Nested children appended this frame
    uint32_t *ndestroyed;  // [max_seg]   particles destroyed by the last update
    // [2][max_seg], by buffer parity like `count`: survivors in the OLD part of a range ring (FwRangeRec) after the last update.
    // Particle 0 of such a ring sits `rold` slots before its first young particle -- the one fact about a range ring only the
    // device knows; everything that addresses its particles by list index (fw_k_spawn, fw_k_nest, fw_k_pack, the AABB query,
    // the host's readers) derives the slot of particle 0 from it
    uint32_t *rold;
    uint32_t max_seg;
    uint32_t seed;
    uint32_t *tile_cnt;              // split mode: survivors per tile
    uint32_t *tile_off;              // split mode: exclusive prefix per tile
    unsigned long long *tile_status; // fused mode: decoupled look-back words
    float *tile_box;                 // [tiles][8] {min.xyz, epoch, max.xyz, -} of the survivors each tile stored (FwUpdateArgs::boxes)
    uint32_t *err;                   // sticky FW_ERR_* flags
    // pinned host word: the first internal error of a kernel also lands here ({1 : 1 | check : 31 | segment : 32}, segment
    // 0xFFFFFFFF = not tied to one), so that the NEXT fw_step sees it without a synchronisation and stops stepping the
    // spawner it belongs to (fw_engine_mem.cpp: poll_device_error) -- an in-place ring update that went wrong cannot be redone
    unsigned long long *err_host;
    unsigned long long *stats;       // [FW_STAT_SLOTS] particles that entered update (running total = their sum; most kernels add to [0])
    unsigned long long *dbg_ts;      // FW_DEBUG & 8: 4 timestamps per tile of the last update (profiling)
    unsigned long long *emit_serial; // RNG serials of Nested emission entries
    unsigned long long *nest_status; // [nested tiles] look-back words of fw_k_nest, tagged with the launch's sequence number
    unsigned long long *nest_ticket; // [n_ops] {workgroups of the op that have finished, children of the op}: the last one commits, then zeroes
    // START tickets (round 5): a workgroup that will wait for workgroups of lower rank -- an OLD tile of a range ring, a tile of
    // parents of a Nested entry -- takes its rank from an atomic counter FIRST THING instead of from its workgroup index, so that
    // "whoever has a lower rank has started" (hence publishes its status word without waiting for anybody of higher rank) is a
    // fact of the program, not a habit of the dispatcher.  Counters only ever grow; the host, which knows how many workgroups of
    // each launch take a ticket, sends the value each counter has when the launch starts (FwRangeRec / FwNestOp / FwFifoNest::
    // ticket_base): no reset, no second pass.  range_ticket: one per segment; nest_start: one per Nested entry (emit slot).
    uint32_t *range_ticket;
    uint32_t *nest_start;
    // MEASURED AND NOT SHIPPED (profiles/r05/tickets_ab.txt, same box, interleaved): the ticket is one more dependent hop in front
    // of everything a latency-bound workgroup does -- configs[2] +1.0 %, one GPU's share of configs[4] +4.5 % (88.4 against 84.6 us),
    // configs[3] +8.6 % with the Nested entry inside the FIFO launch (62.9 against 57.9 us) and +14 % with the separate pass (76.7
    // against 67.0) -- far past the 2 % the guarantee was judged worth.  The product build takes ranks from workgroup indices
    // (FW_TICKETS = 0); the wait stays bounded by the dispatcher's habit of starting lower indices first, and a wait that does run
    // out raises the sticky per-spawner error of DESIGN.md section 11.  `tools/build_variant.sh tickets -DFW_TICKETS=1` builds the
    // guaranteed form (the whole GPU suite passes on it).
#ifndef FW_TICKETS
#define FW_TICKETS 0
#endif
    const FwCollider *colliders;         // the world particle_collision casts its rays into (fw_ctx_set_colliders)
    uint32_t n_colliders;
};

struct FwUpdateArgs {
    const uint32_t *seg_tile_first;  // [n_seg + 1] first tile of each segment (device)
    const uint4 *tile_desc;          // [total_tiles] {segment, its first tile, its tile count, type index} (device)
    uint32_t n_seg;
    uint32_t total_tiles;
    uint32_t parity;       // read buf[parity], write buf[parity ^ 1]
    uint32_t epoch;        // frame number (look-back tag), never 0
    float dt;
    uint32_t spin_limit;   // look-back polls before the self-computed fallback
    uint32_t any_inst;    // some segment has an attached instance buffer: run the kernels that also write the records
    // pinned host word: workgroup 0 stores done_value at its START.  Launches of a stream run in order, so the host
    // reads "every launch before frame done_value has finished" without an event (recycling of per-frame host rings)
    unsigned long long *done_tag;
    unsigned long long done_value;
    uint32_t new_static;  // 1: every particle spawned this frame survives the step (host-proved), offsets are static
    unsigned long long *host_counts;  // pinned host snapshot row for this frame (or null): epoch << 32 | count
    // Global spawn ops fused into the update (virtual particles appended after the live ones);
    // sorted by destination segment, emission order inside a segment
    const FwOp *ops;               // table form (device memory), or null
    const uint4 *seg_op_first;     // [n_seg] per segment {first op, one past its last op, particles they spawn in all, 0} (table
                                   // form; pinned host memory: ONE bus round trip tells a tile its role -- summing the ops' counts
                                   // over the bus was a second, dependent one)
    uint32_t n_ops;                // ops this frame (inline form: entries of FwInlineOps used)
    uint32_t dbg;                  // FW_DEBUG (profiling only, results wrong): 1 = no look-back, 2 = no integrate
    // survivor forecast (fw_k_update header): table written last frame / table to write this frame
    uint32_t vt_rounds;            // new-particle tiles are vt_rounds * (threads per workgroup) particles
    uint32_t resident_slots;       // fw_k_update workgroups resident at once (256 CUs x 4)
    uint32_t seg0_type;            // type index of segment 0 (used when n_seg == 1)
    uint32_t use_stream;           // forecast frames: run fw_k_update_stream 
    uint32_t seg0_keys_off, seg0_keys_len;  // key pool window of segment 0's type (n_seg == 1)
    // n_seg == 1: the lone segment's record (FwSeg) for THIS frame's parity, so that a tile can address its input from
    // the kernel arguments alone (fw_k_update_stream<.., LONE>); seg0_ib == null: not provided
    const char *seg0_ib;
    char *seg0_ob, *seg0_destroyed, *seg0_inst;
    uint32_t seg0_capacity, seg0_n_lplanes, seg0_inst_cap;
    const uint2 *tile_keys;        // [n_seg] {keys_off, keys_len} of each segment's type (device)
    // optional per-frame total of live particles (feed of the RCCL all-reduce): every segment's finalizer adds its
    // new count to *live_out; workgroup 0 zeroes *live_next (the slot the next frame will use)
    unsigned long long *live_out, *live_next;
    // Survivor forecast, as per-tile sums: P[t] = {lo: particles that survive one more step of this dt and will sit in
    // input tile t of the next frame, counted by the tiles whose output starts in t; hi: those that spill into t + 1}
    // (global tile index); P2[g] = sum of P over tiles [64 g, 64 g + 64), one per 64-byte line from offset fc_s2; the
    // word at fc_tag holds the epoch of the frame that produced the buffer.  Three buffers rotate: read the previous
    // frame's (fc_in; null = not applicable this frame -> decoupled look-back), accumulate into fc_out with atomics,
    // clear fc_zero for the frame after.
    const unsigned long long *fc_in;
    unsigned long long *fc_out;
    unsigned long long *fc_zero;
    // Segments of up to FW_FC_DIRECT tiles keep the forecast as one plain entry per tile instead: {survivors landing in
    // output tile A, in A + 1, A, epoch} written with an ordinary store (no atomics: at 1M particles the thousand
    // atomics cost ~0.7 us of a 25 us kernel), summed by every tile of the segment (at most 4 entries per lane).
    const uint4 *fce_in;
    uint4 *fce_out;
    uint32_t force_colors;  // 1: write base / emissive colour even for constant gradients (the caller rewrote particles)
    uint32_t boxes;    // 1: every tile also leaves the box of position -/+ scale of its survivors in FwGlobals::tile_box
    uint32_t fc_sums;  // 1: some segment exceeds FW_FC_DIRECT tiles -> this launch uses the sums (all segments)
    uint32_t fc_s2, fc_tag;
    // Threshold forecast (round 6, per-tile entries only): the forecast above counts the survivors of ONE more step of the SAME dt.
    // With fc_theta > 0 every tile also leaves what the next frame needs to turn its entry into the forecast for ANY dt' < fc_theta:
    // fct_out[tile] = {survivors stored into output tile A, into A + 1, nA | nB << 16 (0xFFFFFFFF: more than FW_TF_K), first
    // output slot} and the (age, lifetime) pairs of the "risky" survivors -- those a step of fc_theta would destroy -- in
    // fcl_out[tile * FW_TF_K ..]: the ones stored into A from the front, those into A + 1 from the back.  fw_k_fc_resolve
    // (a wave per tile, launched in front of the next update when its dt differs) evaluates fw_survives(age, dt', lifetime) on
    // the pairs -- the update's own expression -- and rewrites the entry's two counts: the streaming schedule then runs as if
    // dt had repeated.  Everybody else survives any dt' < fc_theta (fp32 addition is monotone).
    float fc_theta;
    uint4 *fct_out;
    float2 *fcl_out;
};
#define FW_TF_K 256u  // (2 KB per tile and buffer; a tile of lifetimes in [0.8, 1.2] s lists ~100 at 60 Hz: 64 overflowed on every OLD tile)
// fw_k_fc_resolve: the previous frame's entries / headers / lists, this frame's dt and parity
struct FwResolveArgs {
    uint4 *fce;            // entries of the previous frame (this frame's fce_in): .x / .y are rewritten
    const uint4 *fct;
    const float2 *fcl;
    const uint4 *tile_desc;  // tile -> {segment, ...} (null: one segment)
    uint32_t total_tiles, parity, epoch;
    float dt;
};

// small frames carry their spawn ops in the kernel arguments: no H2D copy, no extra dependency
#define FW_INLINE_OPS 8
struct FwInlineOps {
    FwOp ops[FW_INLINE_OPS];
};

// ---- FIFO (ring) segments ----------------------------------------------------------------------------------------
// A particle type whose lifetime range is a single value destroys its particles in the order it received them: ages
// grow by the same dt for everybody (fp32 addition is monotone), new particles start at 0, so at every update the
// destroyed particles are a PREFIX of the list (reference order, core.rs:590-600 keeps survivors in order).  Nothing
// ever has to move: such a segment lives in ONE buffer used as a ring -- logical particle i sits in slot
// (head + i) mod capacity -- the update works in place (a particle is read and written by the same lane, planes that
// did not change are not written), and the host, which knows every spawn count and replays the fp32 age of each spawn
// cohort, knows head / live count / destroyed count of every frame exactly: no counting pass, no look-back, no
// forecast, whatever dt does.  One record per segment per frame, in the kernel arguments:
struct FwFifoSeg {
    char *buf;               // the ring (FwSeg::buf[0] == buf[1])
    char *destroyed, *inst;  // FwSeg::destroyed / inst (or null)
    uint32_t inst_cap, capacity, seg, type_idx;  // type_idx: | FW_TYPE_IDX_NOSPIN for a type that cannot turn
    uint32_t keys_off, keys_len;
    float life;  // the type's one lifetime value (a no-spin type's Q3 plane -- angular velocity 0, this lifetime -- is not read)
    uint32_t head;     // slot of logical particle 0 BEFORE this update
    uint32_t n_in;     // live particles before this frame's spawns
    uint32_t n_spawn;  // Global particles spawned this frame (logical indices [n_in, n_in + n_spawn))
    uint32_t dead;     // logical indices [0, dead) are destroyed by this update (may reach into the new ones)
    uint32_t op0, op1; // this segment's spawn ops in FwInlineOps
    // workgroups [tile_first, tile_first + n_tiles) of the launch: first n_vt_a + n_vt_b of FW_BLOCK new particles each
    // (new particles [0, spawn_a) occupy the slots up to the end of the buffer, [spawn_a, n_spawn) those from slot 0),
    // then one per ring tile (FW_TILE slots) from tile0 on, covering the particles that were there before
    uint32_t tile0;
    uint32_t tile_first, n_tiles;
    uint32_t spawn_a, n_vt_a, n_vt_b;
    // mat = 1: this frame's new particles of the segment were MATERIALISED in the ring before the update (frames with
    // Nested entries: Global ops by fw_k_spawn, children by fw_k_nest) -- n_spawn is 0, the live count is read from the
    // device counters (count + spawned + appended; n_in is the host's value, or 0xFFFFFFFF when only the device knows:
    // a type that receives Nested children) and the particles from index `count` on get their first update (every
    // plane written).  report (or null): pinned host word that receives {epoch << 32 | particles added this frame} --
    // how the host learns the size of each cohort of children, long before it needs it (when the cohort dies)
    uint32_t mat;
    uint32_t n_lplanes;  // FwSeg::n_lplanes (new particles spawned here initialise those planes: fw_init_last_emitted)
    // a Nested entry run INSIDE this launch (FwFifoNest, round 5): 0 = none; otherwise 1 + its index in FwFifoArgs::nest, with
    // FW_FIFO_NEST_CHILD set in the record of the ring that RECEIVES the children (the other one is the parents' ring)
    uint32_t nest;
    unsigned long long *report;
};
#define FW_FIFO_PER_LAUNCH 8
// Nested emission (core.rs:471-546) inside the FIFO launch (round 5; the frame of configs[3] used to be fw_k_nest -- 13.6 us of
// one dependent chain on 192 workgroups moving 4 % of the frame's bytes -- a launch gap, then the update).  When both rings of a
// Nested entry are FIFO rings of the same launch -- parents fed by Global entries only (the host knows their count), spawned
// inside the update kernel (SegHost::virt_parent); children received by a ring nothing else feeds -- the parents' ring tiles do
// the entry's per-parent pass for their own slots BEFORE they update them (no workgroup reads what another one overwrites):
// compute_emission_count per parent, the advanced last_emitted_age stored, the tile's child total published and its exclusive
// prefix taken from a decoupled look-back over the tiles of lower rank (rank = distance from the ring's head tile = list order,
// core.rs:488-544 keeps children parent-major), children spawned wave-cooperatively and given their FIRST UPDATE right there,
// stored once, in the slot they will live in (behind the child ring's live particles).  The child ring's own tiles never see
// them; its bookkeeping workgroup takes the entry's total from the same status words (one more look-back, over all the parent
// tiles -- they have lower workgroup indices) and books count, RNG serial and the cohort report.
struct FwFifoNest {
    uint32_t parent, child;      // indices into FwFifoArgs::s (parent < child: the parents' workgroups come first)
    uint32_t emit, emit_slot;    // -> FwEmit, -> FwGlobals::emit_serial
    uint32_t status_first;       // first look-back word of the entry in FwGlobals::nest_status (tile of rank r: status_first + r)
    uint32_t n_ptiles;           // ring tiles of the parents' ring that take part (ranks 0 .. n_ptiles - 1)
    uint32_t parent_lplane;      // which last_emitted_age plane of the parent type belongs to the entry
    uint32_t tag;                // tag of the look-back words of this launch (fw_ctx::nest_seq)
    uint32_t ticket_base;        // value of FwGlobals::nest_start[emit_slot] when this launch starts (the tile of rank r holds ticket_base + r)
    float n_count, n_start, n_end;  // CountOverDuration of the entry (core.rs:474-481)
    float speed, scale;          // EffectModifier
    uint32_t spin_limit, pad;
};
#define FW_FIFO_NEST_MAX 4
#define FW_FIFO_NEST_CHILD 0x80000000u
#define FW_LDS_OPS 4u  // ops of one segment fw_k_update_stream parks in LDS (table form: pinned host memory otherwise)
#define FW_FIFO_COLL_TILE FW_BLOCK  // ring tile of a FIFO launch with colliding types: one round per workgroup (fw_k_update_fifo: TR)
struct FwFifoArgs {
    FwFifoSeg s[FW_FIFO_PER_LAUNCH];
    uint32_t n_segs, parity, epoch, dbg;
    float dt;
    uint32_t any_inst;
    uint32_t any_coll;  // some segment's particle type has collision settings: the launch runs the COLL instantiation (fw_k_rings.hip: FwCollArm)
    uint32_t small_tiles;  // 1: the host laid the launch out on ring tiles of FW_FIFO_COLL_TILE slots (one round per workgroup):
                           // colliding launches, and launches too small to fill the chip with four-round workgroups
    // which optional planes the particle types of this launch write: bit 0 base colour (gradient not constant), bit 1
    // emissive colour, bit 2 scale (curve not constant) when all its segments agree -- the kernel is then compiled for
    // exactly that set of stores; -1: they differ, read the flags from each type
    int32_t write_mask;
    unsigned long long *done_tag;      // as in FwUpdateArgs
    unsigned long long done_value;
    unsigned long long *host_counts;
    unsigned long long *live_out, *live_next;
    uint32_t n_nest, pad_nest;         // Nested entries run inside this launch (the NEST instantiations of fw_k_update_fifo)
    FwFifoNest nest[FW_FIFO_NEST_MAX];
};

// ---- range rings: particle types whose lifetime is a RANGE, updated in place ---------------------------------------
// Ages never increase along the particle list (everybody is born with age 0 behind all older particles and every update
// adds the same dt, core.rs:523, 594) and nobody dies before `age + dt >= lifetime.min`: only a PREFIX of the list -- the
// OLD part -- can lose particles in an update; everybody younger is in the situation of a FIFO ring particle.  Such a type
// lives in ONE buffer used as a ring:   [ ... free ... | old survivors | young | free ... ]
//   * the YOUNG part is updated in place (a lane owns its slot from load to store; no counting, no look-back, any dt);
//     the host knows it exactly -- it made every spawn count and replays the fp32 age of every spawn cohort -- as
//     {b = slot of the first young particle, y = their number};
//   * the OLD part [b - n_old, b) is compacted IN PLACE towards the young part (order kept): a tile of 1024 holds its
//     whole input in registers before it publishes its survivor count, so the tiles further from b -- which need that
//     count for their offset, and whose survivors land in slots of the tiles nearer to b -- cannot overwrite anything
//     that is still to be read.  Tiles nearer to b have lower workgroup indices: whoever is waited for is resident or done;
//   * cohorts whose age reaches lifetime.min simply join the old part: b moves, nothing is copied;
//   * new particles are spawned at the tail, in the slot they will live in.
// The first particle of the list sits in slot b - n_old: `n_old` = count - y is known to the device only.
struct alignas(16) FwRangeRec {  // per segment, per frame: pinned host memory, read by the tiles in place
    uint32_t b;         // slot of the first young particle, after this frame's cohorts have joined the old part
    uint32_t y_exist;   // young particles before this frame's spawns (host-known; FW_RREC_DEV: unknown, the device derives it)
    uint32_t n_spawn;   // particles spawned this frame by the NEW workgroups (all of them outlive the step: dt < lifetime.min)
    uint32_t op0, op_n; // their spawn ops in FwRangeArgs::ops
    uint32_t grad;      // particles that joined the old part this frame (b moved by so many slots): old part = rold + grad
    uint32_t flags;     // FW_RREC_*
    uint32_t ticket_base;  // value of FwGlobals::range_ticket[segment] when this launch starts (its OLD workgroup of rank k holds ticket_base + k)
    unsigned long long *report;  // FW_RREC_DEV: pinned host word that receives {epoch << 32 | particles added this frame}
    unsigned long long pad2;
};
// Range rings of spawners WITH Nested entries (core.rs:471-546):
// FW_RREC_MAT  this frame's new particles of the segment were MATERIALISED behind the young part before the update (frames
//              with a Nested pass: Global ops by fw_k_spawn -- the per-parent pass must find them in memory, core.rs:488 --
//              children by fw_k_nest): n_spawn is 0, the device counters say how many (spawned + appended), and those
//              particles get their first update with every plane written;
// FW_RREC_DEV  the type receives Nested children: how many particles it holds is known to the device only -- the young
//              count is  count + spawned + appended - (rold + grad)  -- and the size of each frame's cohort reaches the host
//              through `report`, long before the host needs it (when the cohort joins the old part).
#define FW_RREC_MAT 1u
#define FW_RREC_DEV 2u
#define FW_RANGE_OLD 0u
#define FW_RANGE_NEW 1u
#define FW_RANGE_YOUNG 2u
struct alignas(16) FwRangeDesc {  // per workgroup (device table, re-sent only when a bound leaves its band)
    uint32_t seg;
    uint32_t role_k;     // role << 30 | index of the workgroup within its role and segment
    uint32_t old_first;  // index of the segment's first look-back word in FwRangeArgs::status (its OLD workgroup k uses word old_first + k)
    uint32_t type_idx;   // | FW_TYPE_IDX_NOSPIN
    uint32_t keys_off, keys_len;
    uint32_t n_old;      // OLD workgroups the table provides for the segment (the kernel checks the old part against it)
    uint32_t pad;
};
struct FwRangeArgs {
    const FwRangeDesc *desc;
    const FwRangeRec *recs;         // [max_seg] indexed by segment (pinned host)
    const FwOp *ops;                // this frame's spawn ops of range segments (pinned host)
    unsigned long long *status;     // look-back words of the OLD workgroups (FwRangeDesc::old_first + k)
    uint32_t total_tiles, parity, epoch, spin_limit, dbg;
    float dt;
    uint32_t any_inst;              // some segment has a windowed instance buffer attached: the kernels that also write records
    uint32_t any_coll;              // some segment's particle type has collision settings: the COLL instantiation (FwCollArm)
    uint32_t small_tiles;           // 1: OLD and YOUNG workgroups cover ONE round (256 slots) each: the host laid the launch out so
                                    // (colliding launches, and launches too small to fill the chip with four-round workgroups)
    uint32_t young_rounds;          // rounds of a YOUNG workgroup of a four-round launch: 4, or 2 (launches of large segments, no records)
    unsigned long long *done_tag;   // as in FwUpdateArgs
    unsigned long long done_value;
    unsigned long long *host_counts;
    unsigned long long *live_out, *live_next;
    unsigned long long *ts;         // FW_DEBUG & 8: 8 words per workgroup {start, 0, 0, end of wave 0, role_k, seg, 0, 0} (profiling)
};
#define FW_RANGE_MAX_CAPACITY 0x10000000u  // slots are addressed as 32-bit byte offsets into a float4 plane

// ---- small particle types: one WAVE per (spawner, particle type) (fw_k_small.hip, round 5) ---------------------------------
// A type of a few hundred particles needs no workgroup, no tile table, no look-back and no forecast: a wave walks its list in
// rounds of 64, a ballot + a running count give the stable compaction (core.rs:589-659), its uniform values live in the wave's own
// scalar registers -- four types per workgroup, thousands of emitters resident at once.  Same ping-pong layout as the compacting
// path: a type enters and leaves the mode by a host flag (SegHost::small).
struct FwSmallArgs {
    const uint32_t *list;          // [n] segments of the launch (device): n_narrow types a wave walks, then n - n_narrow WIDE ones
                                   // (up to a few thousand particles: a workgroup each)
    uint32_t n, n_narrow, parity, epoch;
    uint32_t any_inst;             // some type of the launch has an instance buffer attached (FwSeg::inst): the INST instantiation
    uint32_t any_coll;             // ... collision settings (FwTypeColl): the COLL instantiation
    float dt;
    const uint4 *seg_op_first;     // as FwUpdateArgs (table form: pinned host memory), or null: no virtual spawns this frame
    const FwOp *ops;
    uint32_t force_colors, dbg;
    unsigned long long *done_tag;  // as in FwUpdateArgs
    unsigned long long done_value;
    unsigned long long *host_counts;
    unsigned long long *live_out, *live_next;
};
hipError_t fw_launch_update_small(hipStream_t s, const FwGlobals &g, const FwSmallArgs &a, hipEvent_t ev_start = nullptr,
                                  hipEvent_t ev_stop = nullptr);

enum { FW_SPAWN_NONE = 0, FW_SPAWN_INLINE = 1, FW_SPAWN_TABLE = 2 };

enum { FW_MODE_FUSED = 0, FW_MODE_SPLIT = 1, FW_MODE_SPLIT_COLL = 2 };  // SPLIT_COLL: frames with colliding particle types

// d_ops: device table, or null -> the (at most FW_INLINE_OPS) ops at h_ops travel in the kernel arguments
hipError_t fw_launch_spawn(hipStream_t s, const FwGlobals &g, const FwOp *d_ops, const FwOp *h_ops, uint32_t n_ops,
                           uint32_t total_blocks, uint32_t parity);
hipError_t fw_launch_fc_resolve(hipStream_t s, const FwGlobals &g, const FwResolveArgs &a);
hipError_t fw_launch_update(hipStream_t s, const FwGlobals &g, const FwUpdateArgs &a, const FwInlineOps *inl,
                            int spawn_form, int mode, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// in-place update of up to FW_FIFO_PER_LAUNCH FIFO segments (their spawn ops in `inl`)
hipError_t fw_launch_update_fifo(hipStream_t s, const FwGlobals &g, const FwFifoArgs &a, const FwInlineOps &inl,
                                 uint32_t total_tiles, int nt, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
uint32_t fw_range_young_tile(void);  // ring slots a YOUNG workgroup of fw_k_update_range covers (a build-time choice)
// in-place update of every range ring of the context (all_nospin: no segment of the launch keeps a rotation plane)
// nt: which form of the kernel (fw_dev.h: fw_ld4w): 0 plain, 1 the write-only planes non-temporal, 2 every plane access
hipError_t fw_launch_update_range(hipStream_t s, const FwGlobals &g, const FwRangeArgs &a, bool all_nospin, int nt,
                                  hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t fw_launch_nested(hipStream_t s, const FwGlobals &g, const FwNestOp *d_ops, const FwNestOp *h_ops, uint32_t n_ops,
                            uint32_t total_tiles, uint32_t parity, uint32_t tag, uint32_t spin_limit, uint32_t dbg = 0);
// SoA -> AoS gather of `n` particles of one segment buffer into fw_particle records (device)
// (head: slot of particle 0 -- 0 for every segment but a FIFO ring)
// (const_rot: the rotation of a type that cannot turn -- FW_TYPE_NOSPIN, its plane is not maintained -- or null)
// (... and then its lifetimes sit in plane `life_plane` behind the last_emitted_age planes, or -- a ring: 0xFFFFFFFF -- all
// equal life_const)
// (derived: the type's device record when it is FW_TYPE_DERIVED -- scale and colours are then evaluated, not read -- + the key pool)
hipError_t fw_launch_gather(hipStream_t s, const char *buf, uint32_t capacity, uint32_t head, uint32_t n, int32_t pbr, void *d_out,
                            const float *const_rot = nullptr, uint32_t life_plane = 0xFFFFFFFFu, float life_const = 0.0f,
                            const FwType *derived = nullptr, const float *keys = nullptr, bool cpl = false);
// (cpl: the segment is a ring -- its Q1 / Q3 regions are component planes, fw_dev.h)
// fills the scale / colour planes of one buffer from age, lifetime and initial_scale (a type leaves FW_TYPE_DERIVED)
hipError_t fw_launch_rederive(hipStream_t s, char *buf, uint32_t capacity, const FwType *d_type, const float *d_keys, bool nospin,
                              uint32_t life_plane, float life_const, bool cpl);
hipError_t fw_launch_fill_plane1(hipStream_t s, char *buf0, char *buf1, size_t plane_off, uint32_t capacity, float v);
hipError_t fw_launch_restore_q3(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, uint32_t life_plane, float life_const, bool cpl);
hipError_t fw_launch_scatter(hipStream_t s, char *buf, uint32_t capacity, uint32_t n, uint32_t n_lplanes,
                             const void *d_in);
// fills the base / emissive colour planes of one (buf1 == nullptr) or both buffers of a segment (capacity slots each)
hipError_t fw_launch_fill_colors(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, const float bc[4], const float em[4]);
// (d_rold != null: a range ring -- `head` is the slot of its first YOUNG particle and particle 0 sits *d_rold slots before it:
// the segment's word of FwGlobals::rold)
hipError_t fw_launch_pack_instances(hipStream_t s, const char *buf, uint32_t capacity, uint32_t head, const uint32_t *d_count,
                                    uint32_t n_upper, void *d_out, const float *const_rot = nullptr,
                                    const uint32_t *d_rold = nullptr, const FwType *derived = nullptr, const float *keys = nullptr,
                                    uint32_t life_plane = 0xFFFFFFFFu, float life_const = 0.0f, bool cpl = false);
// fills the rotation plane of both buffers of a segment (a type leaves FW_TYPE_NOSPIN)
hipError_t fw_launch_fill_rotation(hipStream_t s, char *buf0, char *buf1, uint32_t capacity, const float rot[4]);
// seg_ids: host array; d_part: device scratch of 256 * 8 floats; h_out8: PINNED host {min.xyz, any, max.xyz, -}
// seg_heads: ring heads of the segments (host array, or null = all 0); seg_range_y (or null): per segment 0xFFFFFFFF, or --
// a range ring -- anything else: seg_heads[i] is then the slot of its first young particle (see fw_launch_pack_instances)
hipError_t fw_launch_aabb(hipStream_t s, const FwGlobals &g, const uint32_t *seg_ids, const uint32_t *seg_heads, uint32_t n_segs,
                          uint32_t parity, float *d_part, float *h_out8, const uint32_t *seg_range_y = nullptr,
                          const uint32_t *seg_life_plane = nullptr, const float *seg_life_const = nullptr);
// the same query answered from the per-tile boxes of the last update (epoch = that update's)
hipError_t fw_launch_aabb_from_tiles(hipStream_t s, const FwGlobals &g, const uint32_t *seg_ids, uint32_t n_segs,
                                     uint32_t parity, uint32_t epoch, const uint32_t *d_seg_tile_first, float *h_out8);
hipError_t fw_launch_total(hipStream_t s, const uint32_t *counts, uint32_t n_seg, unsigned long long *d_out);
hipError_t fw_launch_copy_probe(hipStream_t s, const void *src, void *dst, size_t bytes);
