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
#include "fw_dev.h"

__global__ __launch_bounds__(FW_BLOCK) void fw_k_update_small(FwGlobals g, FwSmallArgs a) {
    constexpr int NW = FW_BLOCK / 64;
    __shared__ __attribute__((aligned(16))) float s_keys_all[NW][FW_KEYS_MAX];
    __shared__ unsigned long long s_entered[NW], s_live[NW];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (a scalar: everything indexed by it is wave-uniform)
    const uint32_t idx = blockIdx.x * NW + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.live_next) *a.live_next = 0ull;
        if (a.done_tag) *a.done_tag = a.done_value;
    }
    unsigned long long entered = 0ull, live = 0ull;
    if (idx < a.n) {
        const uint32_t seg = a.list[idx];
        const FwSeg *Sp = &g.segs[seg];
        const uint32_t C = Sp->capacity, n_lplanes = Sp->n_lplanes;
        const char *ib = Sp->buf[a.parity];
        char *ob = Sp->buf[a.parity ^ 1u];
        char *destroyed = Sp->destroyed;
        const uint32_t sidx = a.parity * g.max_seg + seg, oidx = (a.parity ^ 1u) * g.max_seg + seg;
        const uint32_t n_cnt = g.count[sidx];
        const uint32_t n_in = min(n_cnt + g.spawned[sidx] + g.appended[sidx], C);  // loaded: the live ones + what a pass materialised
        // this frame's spawn ops of the segment (table form, pinned host memory: one header per segment)
        uint32_t o0 = 0u, o1 = 0u, n_spawn = 0u;
        if (a.seg_op_first) {
            const uint4 oh = a.seg_op_first[seg];
            o0 = oh.x, o1 = oh.y, n_spawn = oh.z;
        }
        const uint32_t room = C - n_in;
        if (n_spawn > room) {  // virtual spawns beyond the capacity are dropped (and reported), as everywhere
            n_spawn = room;
            if (lane == 0) fw_flag(g, FW_ERR_CAPACITY);
        }
        const FwType T = g.types[Sp->type_idx];
        float *s_keys = s_keys_all[wave];
        for (uint32_t i = lane; i < T.keys_len; i += 64u) s_keys[i] = g.keys[T.keys_off + i];
        const bool nospin = (T.flags & FW_TYPE_NOSPIN) != 0u;
        const bool want_destroyed = T.report_destroyed && destroyed != nullptr;
        const FwOutWin W = fw_out_window(ob, C, 0u, T, a.force_colors, n_lplanes);
        const char *p0 = ib + FW_OFF_Q0(C), *p1 = ib + FW_OFF_Q1(C), *p2 = ib + FW_OFF_Q2(C), *p3 = ib + FW_OFF_Q3(C);
        const char *pl = ib + FW_OFF_L(C, n_lplanes);  // lifetimes of a type that cannot turn (FwOutWin::lf)
        // (the wave's LDS row was written by its own lanes: a wave-scope fence orders it against the reads below)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t run = 0u;  // survivors stored so far = the next output slot
        // ---- the particles that are in memory, in list order
        const uint32_t rounds = (n_in + 63u) / 64u;
        auto load = [&](uint32_t r, float4 &q0, float4 &q1, float4 &q2, float4 &q3) {
            const uint32_t i = min(r * 64u + lane, n_in ? n_in - 1u : 0u);  // (unconditional loads at a clamped index)
            q0 = fw_ld4w(p0, i * 16u), q1 = fw_ld4w(p1, i * 16u);
            if (nospin) {  // (wave-uniform branch)
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
            const uint32_t i = r * 64u + lane;
            const bool valid = i < n_in;
            float age_new;
            const bool alive = valid && fw_survives(q0.w, a.dt, q3.w, &age_new);
            const unsigned long long m = __ballot(alive);
            const uint32_t o = run + fw_lane_prefix(m);
            if (alive) {
                fw_integrate_store(T, s_keys, a.dt, q0, q1, q2, q3, age_new, W, o);
            } else if (valid && want_destroyed) {
                // (a particle a pass materialised this frame carries its spawn-time scale and colours: evaluated, not read)
                fw_store_destroyed(destroyed, ib, C, i, i < n_cnt, T, s_keys, q0, q1, q2, q3, age_new, i - o);
            }
            run += (uint32_t)__popcll(m);
        }
        // ---- this frame's new particles: spawn_particles (core.rs:437-469) right before update_particles, in op order
        for (uint32_t x = o0; x < o1; x++) {
            const FwOp &op = a.ops[x];  // (wave-uniform: scalar loads, over the bus)
            const uint32_t rel = op.rel_base, cnt = rel < n_spawn ? min(op.n, n_spawn - rel) : 0u;
            const FwEmit &e = g.emits[op.emit];
            for (uint32_t c0 = 0; c0 < cnt; c0 += 64u) {
                const uint32_t k = c0 + lane;
                const bool valid = k < cnt;
                FwSpawnOut so;
                so.q0 = so.q1 = so.q2 = so.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid)
                    so = fw_spawn_one(e, g.seed, op.serial_base + k, fw_v3{op.origin_pos[0], op.origin_pos[1], op.origin_pos[2]},
                                      fw_q4{op.origin_rot[0], op.origin_rot[1], op.origin_rot[2], op.origin_rot[3]},
                                      fw_v3{op.parent_vel[0], op.parent_vel[1], op.parent_vel[2]}, op.speed, op.scale);
                float age_new;
                const bool alive = valid && fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);
                const unsigned long long m = __ballot(alive);
                const uint32_t o = run + fw_lane_prefix(m);
                const uint32_t i = n_in + rel + k;  // its list index before the update
                if (alive) {
                    fw_integrate_store(T, s_keys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, W, o);
                } else if (valid && want_destroyed) {  // born and destroyed in the same frame (dt >= its lifetime)
                    fw_store_destroyed(destroyed, ib, C, i, false, T, s_keys, so.q0, so.q1, so.q2, so.q3, age_new, i - o);
                }
                run += (uint32_t)__popcll(m);
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
    // statistics and the frame's live total: one atomic each per WORKGROUP (thousands on one word serialise at the memory side)
    if (lane == 0) s_entered[wave] = entered, s_live[wave] = live;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long e = 0ull, l = 0ull;
#pragma unroll
        for (int w = 0; w < NW; w++) e += s_entered[w], l += s_live[w];
        if (e && !FW_DBG(a.dbg, 128u)) atomicAdd(g.stats, e);
        if (a.live_out && l) atomicAdd(a.live_out, l);
    }
}

hipError_t fw_launch_update_small(hipStream_t s, const FwGlobals &g, const FwSmallArgs &a, hipEvent_t e0, hipEvent_t e1) {
    if (!a.n) return hipSuccess;
    const dim3 grid((a.n + FW_BLOCK / 64 - 1) / (FW_BLOCK / 64)), block(FW_BLOCK);
    FW_LAUNCH_T(fw_k_update_small, grid, block, s, e0, e1, g, a);
    return hipGetLastError();
}
