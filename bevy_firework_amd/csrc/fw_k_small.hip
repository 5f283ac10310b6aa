// fw_k_small.hip -- update_particles for SMALL particle types: ONE WAVE per (spawner, particle type) (round 5)
// (gfx950 only; device helpers in fw_dev.h, launch interface in fw_kernels.h)
//
// The reference advertises "tens of thousands of particles" spread over many spawner entities (README.md:11-16): thousands of
// emitters of a few hundred particles each.  On the compacting kernels such a type is one workgroup: ~18 us of dependent
// latencies (descriptor -> segment record / counters / op header -> particles -> its three or four new particles) for 200
// particles, 2048 of them two rounds of residency -- 44 us per frame, whatever the host does (profiles/r04/small_emitters.txt).
// Nothing in that workgroup needs more than a wave: 200 particles are four rounds of 64 lanes, their stable compaction
// (core.rs:589-659) a ballot + a running count, no counting pass, no look-back, no forecast, no tile table -- and a wave's
// uniform values (segment record, type constants, spawn ops) live in scalar registers of ITS OWN, so the four waves of a
// workgroup serve four different particle types: 2048 emitters are 512 workgroups, resident at once.
//
// A wave walks its type's particle list in rounds of 64 (next round's loads in flight), integrates the survivors in the
// reference's operation order (fw_integrate_store) and stores them at the running survivor count in the other buffer of the
// ping-pong pair -- the layout of the compacting path, so a type enters and leaves this mode by a flag on the host, nothing is
// copied --, then spawns the frame's new particles behind them (virtual particles, as in the compacting kernels) or, in frames
// that materialised them (Nested passes, collisions elsewhere in the context), finds them behind the live ones.
//
// WIDE types (round 5, later): a type of up to a few thousand particles -- hundreds of emitters of a thousand particles each: 1024 x
// 1000 takes the compacting kernels 55 us a frame, two workgroups per type, each a chain of tile table -> forecast -> count ->
// look-back -> update -- is walked the same way by a WORKGROUP: rounds of 256 lanes, the waves' survivor counts exchanged through LDS
// (one barrier per round, double-buffered), everything else as above.  One launch serves both kinds: its first workgroups take four
// narrow types each, the rest one wide type each (a workgroup-uniform branch).
//
// Instantiations <INST, COLL>: INST writes the 64-byte render record of every survivor into the type's attached instance buffer
// (render.rs:95-115; the scale / colour planes are then not stored: FW_TYPE_DERIVED); COLL runs particle_collision (core.rs:607-643) on
// every particle that passed the age test, destroy_on_collision included.  The host picks the instantiation per launch from what the
// context's small types need (FwSmallArgs::any_inst / any_coll): the plain one keeps its 120 registers.
#include "fw_dev.h"
#include "fw_collide.h"
#ifndef FW_SMALL_EXP  // (profiling experiments, results wrong: 1 no spawn phase -- the populations die out --, 2 no integration, 4 no statistics atomics,
                      // 8 no op / header reads, 16 spawn without random numbers and trigonometry)
#define FW_SMALL_EXP 0
#endif

// one particle type: by one wave (WIDE = false; `lanes` = 64) or by the four waves of a workgroup (WIDE = true; `lanes` = FW_BLOCK)
// (INST: some type of the launch has an instance buffer attached -- fw_spawner_attach_instances: the update writes the 64-byte render
// record of every survivor itself, at its list index, and the type's scale / colour planes are not stored: FW_TYPE_DERIVED)
// (COLL: some type of the launch has collision settings -- core.rs:607-643: a particle that survives the age test is moved by
// particle_collision instead of the plain Euler step, and destroyed by it when the type says destroy_on_collision; the stable
// compaction takes a destroyed particle out like one that died of age, so ANY colliding type can live here)
template <bool WIDE, bool INST, bool COLL>
__device__ __forceinline__ void fw_small_type(const FwGlobals &g, const FwSmallArgs &a, uint32_t seg, float *s_keys, uint32_t (*s_cnt)[FW_BLOCK / 64],
                                              unsigned long long &entered, unsigned long long &live) {
    constexpr int NW = FW_BLOCK / 64;
    constexpr uint32_t LANES = WIDE ? (uint32_t)FW_BLOCK : 64u;
    const uint32_t wlane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = WIDE ? wave * 64u + wlane : wlane;  // index within the round
    const FwSeg *Sp = &g.segs[seg];
    const uint32_t C = Sp->capacity, n_lplanes = Sp->n_lplanes;
    const char *ib = Sp->buf[a.parity];
    char *ob = Sp->buf[a.parity ^ 1u];
    char *destroyed = Sp->destroyed;
    char *inst = INST ? Sp->inst : nullptr;
    const uint32_t inst_cap = INST ? Sp->inst_cap : 0u;
    const uint32_t sidx = a.parity * g.max_seg + seg, oidx = (a.parity ^ 1u) * g.max_seg + seg;
    const uint32_t n_cnt = g.count[sidx];
    const uint32_t n_in = min(n_cnt + g.spawned[sidx] + g.appended[sidx], C);  // loaded: the live ones + what a pass materialised
    FwTypeColl TC{};
    if (COLL) TC = g.type_coll[Sp->type_idx];
    const bool coll = COLL && (TC.coll_flags & FW_COLL_ENABLED) != 0u, coll_kill = COLL && (TC.coll_flags & FW_COLL_DESTROY) != 0u;
    // a survivor's update: integrate, store its planes at slot o -- and its render record, when the type has a buffer for them
    auto update_one = [&](float4 q0, float4 q1, float4 q2, float4 q3, float age_new, const FwType &T_, const float *keys, const FwOutWin &W_, uint32_t o,
                          const fw_v3 &cpos, const fw_v3 &cvel) {
        float4 rec[4];
        fw_integrate_store(T_, keys, a.dt, q0, q1, q2, q3, age_new, W_, o, INST ? rec : nullptr, COLL ? &cpos : nullptr, COLL ? &cvel : nullptr, nullptr,
                           false, false, coll);
        if (INST && inst != nullptr && o < inst_cap) {
            fw_st4(inst, o * 4u + 0u, rec[0]), fw_st4(inst, o * 4u + 1u, rec[1]);
            fw_st4(inst, o * 4u + 2u, rec[2]), fw_st4(inst, o * 4u + 3u, rec[3]);
        }
    };
    // this frame's spawn ops of the segment (table form: one header per segment)
    uint32_t o0 = 0u, o1 = 0u, n_spawn = 0u;
    if (a.seg_op_first && !(FW_SMALL_EXP & 8)) {
        const uint4 oh = a.seg_op_first[seg];
        o0 = oh.x, o1 = oh.y, n_spawn = oh.z;
    }
    const uint32_t room = C - n_in;
    if (n_spawn > room) {  // virtual spawns beyond the capacity are dropped (and reported), as everywhere
        n_spawn = room;
        if (lane == 0) fw_flag(g, FW_ERR_CAPACITY);
    }
    const FwType T = g.types[Sp->type_idx];
    for (uint32_t i = lane; i < T.keys_len; i += LANES) s_keys[i] = g.keys[T.keys_off + i];
    const bool nospin = (T.flags & FW_TYPE_NOSPIN) != 0u;
    const bool want_destroyed = T.report_destroyed && destroyed != nullptr;
    const FwOutWin W = fw_out_window(ob, C, 0u, T, a.force_colors, n_lplanes);
    const char *p0 = ib + FW_OFF_Q0(C), *p1 = ib + FW_OFF_Q1(C), *p2 = ib + FW_OFF_Q2(C), *p3 = ib + FW_OFF_Q3(C);
    const char *pl = ib + FW_OFF_L(C, n_lplanes);  // lifetimes of a type that cannot turn (FwOutWin::lf)
    if (WIDE) {
        __syncthreads();
    } else {  // (the wave's LDS row was written by its own lanes: a wave-scope fence orders it against the reads below)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the record of a particle a collision destroyed (core.rs:633-639): new position, velocity and scale; the rest as it was
    // (existed: it was there before this frame's spawns; in_memory: its planes can be read -- a virtual new particle has none)
    auto store_killed = [&](uint32_t i, bool existed, bool in_memory, float4 q0, float4 q1, float4 q2, float4 q3, float age_new, const fw_v3 &cpos,
                            const fw_v3 &cvel, uint32_t d) {
        const float sc = q1.w * fw_curve_sample(T.sc_kind, T.sc_n, s_keys, s_keys + T.o_sc_v, age_new / q3.w);
        float *rec = reinterpret_cast<float *>(destroyed) + (size_t)d * 26;
        float4 bc, em;
        if ((T.flags & FW_TYPE_DERIVED) && existed) {
            float unused;
            fw_derived_values(T, s_keys, q0.w, q3.w, q1.w, &bc, &em, &unused);
        } else if (in_memory) {  // (materialised this frame: its slot was written in full when it was spawned)
            bc = fw_ld4(ib + FW_OFF_Q5(C), i), em = fw_ld4(ib + FW_OFF_Q6(C), i);
        } else {  // spawn-time colours (core.rs:457-461)
            float b4[4], e4[4];
            fw_gradient_sample(T.bc_kind, T.bc_n, s_keys + T.o_bc_t, s_keys + T.o_bc_v, 0.0f, b4);
            fw_gradient_sample(T.em_kind, T.em_n, s_keys + T.o_em_t, s_keys + T.o_em_v, 0.0f, e4);
            bc = make_float4(b4[0], b4[1], b4[2], b4[3]), em = make_float4(e4[0], e4[1], e4[2], e4[3]);
        }
        const float4 r2 = fw_record_rotation(T, q2);
        rec[0] = cpos.x, rec[1] = cpos.y, rec[2] = cpos.z, rec[3] = cvel.x, rec[4] = cvel.y, rec[5] = cvel.z;
        rec[6] = r2.x, rec[7] = r2.y, rec[8] = r2.z, rec[9] = r2.w, rec[10] = q3.x, rec[11] = q3.y, rec[12] = q3.z;
        rec[13] = q1.w, rec[14] = sc, rec[15] = age_new, rec[16] = q3.w;
        rec[17] = bc.x, rec[18] = bc.y, rec[19] = bc.z, rec[20] = bc.w;
        rec[21] = em.x, rec[22] = em.y, rec[23] = em.z, rec[24] = em.w;
        reinterpret_cast<int32_t *>(rec)[25] = T.pbr;
    };
    uint32_t run = 0u;  // survivors stored so far = the next output slot
    uint32_t xr = 0u;   // WIDE: rounds so far (which half of the count exchange a round uses)
    // where a lane's survivor goes: behind the survivors so far, behind those of the lower waves of this round (WIDE), behind those of
    // the lower lanes of its wave; every lane leaves with `run` advanced by the round's total
    auto place = [&](bool alive, uint32_t *o) {
        const unsigned long long m = __ballot(alive);
        uint32_t base = run, tot = (uint32_t)__popcll(m);
        if (WIDE) {
            if (wlane == 0u) s_cnt[xr & 1u][wave] = tot;
            __syncthreads();  // (the other half is not written before every wave has passed the NEXT round's barrier: no second one)
            tot = 0u;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t c = s_cnt[xr & 1u][w];
                base += (uint32_t)w < wave ? c : 0u;
                tot += c;
            }
            xr++;
        }
        *o = base + fw_lane_prefix(m);
        run += tot;
    };
    // ---- the particles that are in memory, in list order
    const uint32_t rounds = (n_in + LANES - 1u) / LANES;
    auto load = [&](uint32_t r, float4 &q0, float4 &q1, float4 &q2, float4 &q3) {
        const uint32_t i = min(r * LANES + lane, n_in ? n_in - 1u : 0u);  // (unconditional loads at a clamped index)
        q0 = fw_ld4w(p0, i * 16u), q1 = fw_ld4w(p1, i * 16u);
        if (nospin) {  // (uniform branch)
            q2 = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            q3 = make_float4(0.0f, 0.0f, 0.0f, fw_ld1w(pl, i * 4u));
        } else {
            q2 = fw_ld4w(p2, i * 16u), q3 = fw_ld4w(p3, i * 16u);
        }
    };
    float4 q0n, q1n, q2n, q3n;
    if (rounds) load(0u, q0n, q1n, q2n, q3n);
    for (uint32_t r = 0; r < rounds; r++) {
        const float4 q0 = q0n, q1 = q1n, q2 = q2n, q3 = q3n;
        if (r + 1u < rounds) load(r + 1u, q0n, q1n, q2n, q3n);
        const uint32_t i = r * LANES + lane;
        const bool valid = i < n_in;
        float age_new;
        const bool young = valid && fw_survives(q0.w, a.dt, q3.w, &age_new);
        fw_v3 cpos{q0.x, q0.y, q0.z}, cvel{q1.x, q1.y, q1.z};
        bool killed = false;
        if (COLL && young && coll)
            killed = fw_particle_collision(&cpos, &cvel, a.dt, TC.coll_restitution, TC.coll_friction, coll_kill, TC.coll_mask, g.colliders, g.n_colliders);
        const bool alive = young && !killed;
        uint32_t o;
        place(alive, &o);
        if (alive && (FW_SMALL_EXP & 2)) {
            fw_st4(W.q0, o, q0), fw_st4(W.q1, o, q1);
        } else if (alive) {
            update_one(q0, q1, q2, q3, age_new, T, s_keys, W, o, cpos, cvel);
        } else if (valid && want_destroyed) {
            // (a particle a pass materialised this frame carries its spawn-time scale and colours: evaluated, not read)
            if (!COLL || !killed) fw_store_destroyed(destroyed, ib, C, i, i < n_cnt, T, s_keys, q0, q1, q2, q3, age_new, i - o);
            else store_killed(i, i < n_cnt, true, q0, q1, q2, q3, age_new, cpos, cvel, i - o);
        }
    }
    // ---- this frame's new particles: spawn_particles (core.rs:437-469) right before update_particles, in op order
    for (uint32_t x = o0; x < ((FW_SMALL_EXP & 1) ? o0 : o1); x++) {
        const FwOp &op = a.ops[x];  // (uniform: scalar loads)
        const uint32_t rel = op.rel_base, cnt = rel < n_spawn ? min(op.n, n_spawn - rel) : 0u;
        const FwEmit &e = g.emits[op.emit];
        for (uint32_t c0 = 0; c0 < cnt; c0 += LANES) {
            const uint32_t k = c0 + lane;
            const bool valid = k < cnt;
            FwSpawnOut so;
            so.q0 = so.q1 = so.q2 = so.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && (FW_SMALL_EXP & 16)) {  // (no RNG, no trigonometry: what the memory chain of the spawn phase costs by itself)
                so.q0 = make_float4(op.origin_pos[0], op.origin_pos[1], op.origin_pos[2], 0.0f);
                so.q1 = make_float4(0.f, op.speed, 0.f, op.scale), so.q2 = make_float4(0.f, 0.f, 0.f, 1.f), so.q3 = make_float4(0.f, 0.f, 0.f, e.life_min);
            } else if (valid)
                so = fw_spawn_one(e, g.seed, op.serial_base + k, fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                                  fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                                  fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
            float age_new;
            const bool young = valid && fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);
            fw_v3 cpos{so.q0.x, so.q0.y, so.q0.z}, cvel{so.q1.x, so.q1.y, so.q1.z};
            bool killed = false;
            if (COLL && young && coll)
                killed = fw_particle_collision(&cpos, &cvel, a.dt, TC.coll_restitution, TC.coll_friction, coll_kill, TC.coll_mask, g.colliders, g.n_colliders);
            const bool alive = young && !killed;
            uint32_t o;
            place(alive, &o);
            const uint32_t i = n_in + rel + k;  // its list index before the update
            if (alive) {
                update_one(so.q0, so.q1, so.q2, so.q3, age_new, T, s_keys, W, o, cpos, cvel);
            } else if (valid && want_destroyed) {  // born and destroyed in the same frame (dt >= its lifetime, or a collision)
                if (!COLL || !killed) fw_store_destroyed(destroyed, ib, C, i, false, T, s_keys, so.q0, so.q1, so.q2, so.q3, age_new, i - o);
                else store_killed(i, false, false, so.q0, so.q1, so.q2, so.q3, age_new, cpos, cvel, i - o);
            }
        }
    }
    const uint32_t n_tot = n_in + n_spawn;
    if (lane == 0) {
        g.count[oidx] = run, g.spawned[oidx] = 0, g.appended[oidx] = 0;
        g.ndestroyed[seg] = n_tot - run;
        if (a.host_counts) a.host_counts[seg] = ((unsigned long long)a.epoch << 32) | run;
    }
    entered = n_tot, live = run;
}

template <bool INST, bool COLL>
__global__ __launch_bounds__(FW_BLOCK) void fw_k_update_small(FwGlobals g, FwSmallArgs a) {
    constexpr int NW = FW_BLOCK / 64;
    __shared__ __attribute__((aligned(16))) float s_keys_all[NW][FW_KEYS_MAX];
    __shared__ unsigned long long s_entered[NW], s_live[NW];
    __shared__ uint32_t s_cnt[2][NW];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (a scalar: everything indexed by it is wave-uniform)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.live_next) *a.live_next = 0ull;
        if (a.done_tag) *a.done_tag = a.done_value;
    }
    unsigned long long entered = 0ull, live = 0ull;
    const uint32_t narrow_wg = (a.n_narrow + NW - 1u) / NW;  // the first workgroups: a narrow type per wave
    if (blockIdx.x < narrow_wg) {
        const uint32_t idx = blockIdx.x * NW + wave;
        if (idx < a.n_narrow) fw_small_type<false, INST, COLL>(g, a, a.list[idx], s_keys_all[wave], s_cnt, entered, live);
    } else {  // ... then a wide type per workgroup
        unsigned long long e = 0ull, l = 0ull;
        fw_small_type<true, INST, COLL>(g, a, a.list[a.n_narrow + (blockIdx.x - narrow_wg)], s_keys_all[0], s_cnt, e, l);
        if (wave == 0u) entered = e, live = l;  // (every wave leaves with the type's totals: counted once)
    }
    // statistics and the frame's live total: one atomic each per WORKGROUP (thousands on one word serialise at the memory side)
    if (lane == 0) s_entered[wave] = entered, s_live[wave] = live;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long e = 0ull, l = 0ull;
#pragma unroll
        for (int w = 0; w < NW; w++) e += s_entered[w], l += s_live[w];
        // (thousands of workgroups: the running total is kept in FW_STAT_SLOTS words -- readers add them up -- or the adds queue up
        // behind each other on one address: 3-5 us of a 1024-workgroup launch)
        if (e && !FW_DBG(a.dbg, 128u) && !(FW_SMALL_EXP & 4)) atomicAdd(g.stats + (blockIdx.x % FW_STAT_SLOTS), e);
        if (a.live_out && l && !(FW_SMALL_EXP & 4)) atomicAdd(a.live_out, l);
    }
}

hipError_t fw_launch_update_small(hipStream_t s, const FwGlobals &g, const FwSmallArgs &a, hipEvent_t e0, hipEvent_t e1) {
    if (!a.n) return hipSuccess;
    const uint32_t nw = FW_BLOCK / 64;
    const dim3 grid((a.n_narrow + nw - 1) / nw + (a.n - a.n_narrow)), block(FW_BLOCK);
    if (a.any_coll && a.any_inst) FW_LAUNCH_T((fw_k_update_small<true, true>), grid, block, s, e0, e1, g, a);
    else if (a.any_coll) FW_LAUNCH_T((fw_k_update_small<false, true>), grid, block, s, e0, e1, g, a);
    else if (a.any_inst) FW_LAUNCH_T((fw_k_update_small<true, false>), grid, block, s, e0, e1, g, a);
    else FW_LAUNCH_T((fw_k_update_small<false, false>), grid, block, s, e0, e1, g, a);
    return hipGetLastError();
}
