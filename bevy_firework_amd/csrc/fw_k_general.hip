// fw_k_general.hip -- update_particles with stable compaction: fw_k_update (any frame), fw_k_update_stream (forecast frames), and the three-launch feature path fw_k_count / fw_k_scan / fw_k_update_coll
// (gfx950 only; device helpers in fw_dev.h, launch interface in fw_kernels.h)
#include "fw_dev.h"

// The gfx950 (MI355X, CDNA4) kernels of the firework particle backend -- this file and its siblings fw_k_rings.hip,
// fw_k_nested.hip, fw_k_aux.hip.
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
    __shared__ uint32_t s_tf[2];        // threshold forecast: risky survivors noted so far (into A / into A + 1)

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_tf[0] = 0u, s_tf[1] = 0u;  // (the count barrier lies between this and the first round)
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
        const uint4 oh = a.seg_op_first[seg];
        o0 = oh.x, o1 = oh.y, n_spawn += oh.z;
    }
    // virtual spawns beyond the segment's capacity are dropped (and reported): nothing may be read or written past it
    const uint32_t seg_cap = g.segs[seg].capacity;
    if (SPAWN != FW_SPAWN_NONE) {
        const uint32_t spawn_room = seg_cap - min(n_in, seg_cap);
        if (n_spawn > spawn_room) {  // (reported here: nothing about it has to stay live through the kernel)
            n_spawn = spawn_room;
            if (blockIdx.x == first && threadIdx.x == 0) fw_flag(g, FW_ERR_CAPACITY);
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
            if (fc_out && fc_small && a.fc_theta > 0.0f) a.fct_out[tile] = make_uint4(0u, 0u, 0u, 0u);
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
        fw_flag(g, FW_ERR_CAPACITY);
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
                if (tid == 0) fw_flag(g, FW_ERR_LOOKBACK_TIMEOUT);
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
    const float tf_theta = (fc_out && fc_small) ? a.fc_theta : 0.0f;  // threshold forecast (fw_kernels.h): per-tile entries only
    float2 *tf_list = tf_theta > 0.0f ? a.fcl_out + (size_t)tile * FW_TF_K : nullptr;
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
            if (tf_theta > 0.0f) fw_tf_note(tf_theta, tf_list, s_tf, alive, age_new, q3.w, o, fc_bnd);
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
            if (tf_theta > 0.0f) fw_tf_note(tf_theta, tf_list, s_tf, alive, age_new, so.q3.w, o, fc_bnd);
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
            if (tf_theta > 0.0f) fw_tf_header(a.fct_out, tile, s_tf, excl, excl + cnt, fc_bnd);  // (cnt: every survivor of the tile)
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
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats + (seg % FW_STAT_SLOTS), (unsigned long long)n_tot);  // (FW_DEBUG 128: profiling, no statistics)
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
                                                bool fresh = false, float tf_theta = 0.0f, float2 *tf_list = nullptr,
                                                uint32_t *s_tf = nullptr) {  // fresh: materialised this frame, never updated
    if (forecast) {  // will it survive one more step of the same dt?  (same expression as fw_survives)
        float an2;
        const bool nx = alive && fw_survives(age_new, dt, q3.w, &an2);
        acc.fa += (uint32_t)__popcll(__ballot(nx && o < fc_bnd));
        acc.fb += (uint32_t)__popcll(__ballot(nx && o >= fc_bnd));
        if (tf_theta > 0.0f) fw_tf_note(tf_theta, tf_list, s_tf, alive, age_new, q3.w, o, fc_bnd);  // (workgroup-uniform)
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
    __shared__ uint32_t s_tf[2];        // threshold forecast: risky survivors noted so far (into A / into A + 1)

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_tf[0] = 0u, s_tf[1] = 0u;  // (a barrier lies between this and the first round)
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
        const uint4 oh = a.seg_op_first[seg];
        o0 = oh.x, o1 = oh.y, n_spawn += oh.z;
    }
    // Table form: the ops live in pinned HOST memory -- every field a lane reads there is a trip over the bus (~2 us).  A
    // segment's few ops are requested once, one word per lane, as soon as the header has named them, and parked in LDS right
    // before the spawn round needs them (the request is long back by then); segments with more ops read them in place.
    constexpr uint32_t OPW = sizeof(FwOp) / 4u;
    __shared__ __attribute__((aligned(16))) uint32_t s_ops[SPAWN == FW_SPAWN_TABLE ? FW_LDS_OPS * OPW : 1];
    const uint32_t ops_words = (SPAWN == FW_SPAWN_TABLE && o1 - o0 <= FW_LDS_OPS) ? (o1 - o0) * OPW : 0u;  // workgroup-uniform
    uint32_t opw = 0u;
    if (SPAWN == FW_SPAWN_TABLE)  // (unconditional, at a clamped index: see the note on predicated loads below)
        opw = reinterpret_cast<const uint32_t *>(a.ops + (ops_words ? o0 : 0u))[ops_words ? min(tid, ops_words - 1u) : 0u];
    if (SPAWN != FW_SPAWN_NONE) {  // virtual spawns beyond the capacity are dropped (and reported)
        const uint32_t spawn_room = seg_cap - min(n_in, seg_cap);
        if (n_spawn > spawn_room) {
            n_spawn = spawn_room;
            if (blockIdx.x == first && threadIdx.x == 0) fw_flag(g, FW_ERR_CAPACITY);
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
            if (fc_small && a.fc_theta > 0.0f) a.fct_out[tile] = make_uint4(0u, 0u, 0u, 0u);
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
        fw_flag(g, FW_ERR_CAPACITY);
        g.err[1] = seg, g.err[2] = n_tot, g.err[3] = seg_tiles, g.err[4] = n_in;  // diagnostics
    }
    if (blockIdx.x == 0 && tid == 0 && a.live_next) *a.live_next = 0ull;
    if (blockIdx.x == 0 && tid == 0 && a.done_tag) *a.done_tag = a.done_value;
    const float tf_theta = fc_small ? a.fc_theta : 0.0f;  // (threshold forecast: per-tile entries only)
    float2 *tf_list = tf_theta > 0.0f ? a.fcl_out + (size_t)tile * FW_TF_K : nullptr;

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
                if (tid == 0) fw_flag(g, FW_ERR_LOOKBACK_TIMEOUT);
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
                            destroyed, want_destroyed, C, n_lplanes, true, fc_bnd, acc, rec, box, box_on, idx >= n_in, tf_theta, tf_list, s_tf);
            if (INST && inst != nullptr && !FW_DBG(a.dbg, 2u)) fw_inst_flush(inst, inst_cap, s_inst + wave * 256u, lane, m, wbase);
            q0c = q0n, q1c = q1n, q2c = q2n, q3c = q3n, lfc = lfn;
            if (FW_DBG(a.dbg, 8u) && r == 0) tsR1 = __builtin_amdgcn_s_memrealtime() + (o & 0u);
        }
    }
    if (SPAWN != FW_SPAWN_NONE && (!loaded_tile || tail_new)) {
        // ---- new-particle tile (or the one extra round of a live tile that carries its segment's few new particles):
        // spawn_particles (src/core.rs:437-469) right before update_particles, per slot
        const FwOp *opsp = a.ops + o0;  // the segment's first op (a generic pointer: LDS or pinned host memory)
        if (SPAWN == FW_SPAWN_TABLE && ops_words != 0u) {
            if (tid < ops_words) s_ops[tid] = opw;
            __syncthreads();
            opsp = reinterpret_cast<const FwOp *>(s_ops);
        }
#define FW_OPS(i) (SPAWN == FW_SPAWN_INLINE ? inl.ops[i] : opsp[(i) - o0])
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
                    if (k >= FW_OPS(i).rel_base && k - FW_OPS(i).rel_base < FW_OPS(i).n) oi = i;
                const FwOp &op = FW_OPS(oi);
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
                            ob, W, destroyed, want_destroyed, C, n_lplanes, true, fc_bnd, acc, rec, box, box_on, false, tf_theta, tf_list, s_tf);
            if (INST && inst != nullptr && !FW_DBG(a.dbg, 2u)) fw_inst_flush(inst, inst_cap, s_inst + wave * 256u, lane, m, wbase);
        }
#undef FW_OPS
    }
    if (lane == 0) s_part[2][wave] = acc.fa, s_part[3][wave] = acc.fb;
    __syncthreads();
    if (tid == 0) {
        uint32_t sa = 0, sb = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) sa += s_part[2][w], sb += s_part[3][w];
        if (fc_small) a.fce_out[tile] = make_uint4(sa, sb, fcA, a.epoch);
        else fw_fc_add(fc_out, a.fc_s2, first + fcA, sa, sb, seg_tiles);
        if (tf_theta > 0.0f) fw_tf_header(a.fct_out, tile, s_tf, excl, run, fc_bnd);
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
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats + (seg % FW_STAT_SLOTS), (unsigned long long)n_tot);  // (FW_DEBUG 128: profiling, no statistics)
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
            // (the record is always built and the collision values always passed by address, with run-time flags next to them: a
            // pointer selected at run time between a local's address and null forces the local into scratch memory)
            fw_integrate_store(T, s_keys, a.dt, q0, q1, q2, q3, age_new, W, o, rec, &cpos, &cvel, nullptr, false, false, coll);
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
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats + (seg % FW_STAT_SLOTS), (unsigned long long)n_tot);  // (FW_DEBUG 128: profiling, no statistics)
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

// ---- launch wrappers

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

// ---------------------------------------------------------------------------------
// fw_k_fc_resolve (round 6): the previous frame's forecast entries -- "survivors of one more step of the SAME dt" -- turned into the
// forecast for THIS frame's dt (threshold forecast, fw_kernels.h).  One wave per tile of the previous update: its stored
// survivors minus those of its risky ones that a step of `dt` destroys -- fw_survives on the listed (age, lifetime) pairs, the
// update's own expression on the update's own operands, so the counts are what a counting pass over the particles would find.
// A tile that listed more than FW_TF_K risky survivors (lifetimes of a few frames) is recounted from the particles it stored:
// slots [first, first + n) of this frame's input buffer.  Everybody not listed survives any dt below the previous frame's theta
// (fp32 addition is monotone in dt); the host launches this only for such a dt.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(FW_BLOCK) void fw_k_fc_resolve(FwGlobals g, FwResolveArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * (FW_BLOCK / 64u) + (threadIdx.x >> 6);
    if (tile >= a.total_tiles) return;
    const uint4 e = a.fce[tile];
    const uint4 h = a.fct[tile];  // (requested together with the entry)
    if (e.w != a.epoch - 1u) return;  // (not written by the previous update: the consumer's own tag check reports it)
    const uint32_t ta = h.x, tb = h.y;
    uint32_t da = 0, db = 0;  // risky survivors this dt destroys: stored into A / into A + 1
    // (every load of a trip is requested before the first is looked at: a wave that waited for them one by one -- sixteen dependent
    // round trips in the recount -- set the duration of the whole launch: 18.4 us at configs[2] for 8 MB of traffic)
    if (h.z != 0xFFFFFFFFu) {
        const uint32_t na = h.z & 0xFFFFu, nb = h.z >> 16, n = na + nb;
        if (n != 0u) {  // (wave-uniform; most tiles are young: nobody is listed)
            const float2 *L = a.fcl + (size_t)tile * FW_TF_K;
            constexpr int U = FW_TF_K / 64u;
            float2 v[U];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t k = min(lane + (uint32_t)j * 64u, n - 1u);
                v[j] = L[k >= na ? FW_TF_K - 1u - (k - na) : k];
            }
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t k = lane + (uint32_t)j * 64u;
                float an;
                const bool dies = k < n && !fw_survives(v[j].x, a.dt, v[j].y, &an);
                da += (dies && k < na) ? 1u : 0u, db += (dies && !(k < na)) ? 1u : 0u;
            }
        }
    } else {
        const uint32_t seg = a.tile_desc ? a.tile_desc[tile].x : 0u;
        const FwSeg &S = g.segs[seg];
        const char *ib = S.buf[a.parity];
        const uint32_t C = S.capacity, fc_bnd = (e.z + 1u) * FW_TILE, end = h.w + ta + tb;
        const bool nospin = (g.types[S.type_idx].flags & FW_TYPE_NOSPIN) != 0u;
        const uint32_t ls = nospin ? 4u : 16u;
        const char *pa = ib + FW_OFF_Q0(C) + (size_t)h.w * 16u + 12u;  // ages: .w of Q0, from the tile's first slot on
        const char *pl = (nospin ? ib + FW_OFF_L(C, S.n_lplanes) : ib + FW_OFF_Q3(C) + 12u) + (size_t)h.w * ls;  // lifetimes: their own plane, or .w of Q3
        constexpr int U = FW_TILE / 64;
        for (uint32_t i0 = h.w; i0 < end; i0 += FW_TILE) {  // (one trip: a tile stores at most FW_TILE survivors)
            float ag[U], lf[U];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t i = min(i0 + (uint32_t)j * 64u + lane, end - 1u) - h.w;
                ag[j] = fw_ld1w(pa, i * 16u), lf[j] = fw_ld1w(pl, i * ls);
            }
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t i = i0 + (uint32_t)j * 64u + lane;
                float an;
                const bool dies = i < end && !fw_survives(ag[j], a.dt, lf[j], &an);
                da += (dies && i < fc_bnd) ? 1u : 0u, db += (dies && !(i < fc_bnd)) ? 1u : 0u;
            }
        }
    }
    da = fw_wave_sum(da), db = fw_wave_sum(db);
    if (lane == 0) a.fce[tile] = make_uint4(ta - da, tb - db, e.z, e.w);
}
hipError_t fw_launch_fc_resolve(hipStream_t s, const FwGlobals &g, const FwResolveArgs &a) {
    if (!a.total_tiles) return hipSuccess;
    const uint32_t per = FW_BLOCK / 64u;
    hipLaunchKernelGGL(fw_k_fc_resolve, dim3((a.total_tiles + per - 1u) / per), dim3(FW_BLOCK), 0, s, g, a);
    return hipGetLastError();
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

