// fw_k_rings.hip -- update_particles IN PLACE: fw_k_update_fifo (one lifetime value: the dead are a prefix) and fw_k_update_range (lifetime ranges: young part in place, old part compacted in place)
// (gfx950 only; device helpers in fw_dev.h, launch interface in fw_kernels.h)
#include "fw_dev.h"

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

// ---- Nested emission inside the FIFO launch (fw_kernels.h: FwFifoNest; core.rs:471-546) -------------------------------------
// Called by a ring tile of the PARENTS' ring (rank = its distance from the ring's head tile = list order) before it updates its
// slots: what fw_k_nest does for a tile of parents -- per-parent compute_emission_count (device fp32, bit-exact), the advanced
// last_emitted_age stored (core.rs:490-500), the tile's child total published, its exclusive prefix from a decoupled look-back
// over the tiles of lower rank (lower workgroup indices), children written wave-cooperatively in parent-major order
// (core.rs:488-544) -- except that a child is not parked in memory for a later update: it gets its first update here (the frame's
// update_particles, core.rs:577-659, would visit it next) and is stored once, in the slot it will live in.  Parents the update
// is about to destroy still emit (spawn_particles runs first, plugin.rs:46-60).  Nothing is committed here: the child ring's
// bookkeeping workgroup reads the entry's total from the status words (fw_k_update_fifo, end).
template <int R, int NT>
__device__ __forceinline__ void fw_fifo_nest_parents(const FwGlobals &g, const FwFifoArgs &a, const FwFifoNest &N, const FwFifoSeg &F,
                                                     const FwFifoSeg &Fc, uint32_t rank, uint32_t sbase, uint32_t n_in) {
    constexpr int BLK = FW_BLOCK, NW = BLK / 64, LBW = 4;
    __shared__ uint32_t s_w[R][NW];
    __shared__ uint32_t s_lb[2 * LBW * NW];
    // (per-parent counts and their inclusive prefix within the wave, for every round: in LDS rather than in 2 R registers that would
    // be live across the whole spawn + first-update of the children -- the four-round form sat at 133 VGPRs with them)
    __shared__ uint32_t s_inc[R][NW][64];
    __shared__ uint32_t s_n[R][BLK];
    __shared__ __attribute__((aligned(16))) float4 s_par[NW][3][64];
    __shared__ __attribute__((aligned(16))) float s_ckeys[FW_KEYS_MAX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t C = F.capacity, head = F.head, Cc = Fc.capacity;
    char *pb = F.buf;
    // (window addressing as everywhere in this kernel: the tile's first slot on the scalar unit, 32-bit offsets per lane)
    const char *w0 = pb + FW_OFF_Q0(C) + (size_t)sbase * 16u, *w1 = pb + FW_OFF_Q1(C) + (size_t)sbase * 4u;  // (w1: component planes)
    const char *w2 = pb + FW_OFF_Q2(C) + (size_t)sbase * 16u;
    char *wl = pb + FW_OFF_L(C, N.parent_lplane) + (size_t)sbase * 4u;
    const bool pnospin = (F.type_idx & FW_TYPE_IDX_NOSPIN) != 0u;
    float p_age[R], p_lea[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t o = (uint32_t)(r * BLK) + tid;
        p_age[r] = fw_ld1w(w0, o * 16u + 12u);
        p_lea[r] = fw_ld1w(wl, o * 4u);
    }
    const uint32_t cidx = a.parity * g.max_seg + Fc.seg;
    const uint32_t cbase = g.count[cidx] + g.spawned[cidx] + g.appended[cidx];  // first child slot of the entry (list index)
    const unsigned long long serial0 = g.emit_serial[N.emit_slot];
    const FwEmit &e = g.emits[N.emit];
    const FwType Tc = g.types[Fc.type_idx & ~FW_TYPE_IDX_NOSPIN];
    for (uint32_t i = tid; i < Fc.keys_len; i += BLK) s_ckeys[i] = g.keys[Fc.keys_off + i];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t o = (uint32_t)(r * BLK) + tid, s = sbase + o;
        uint32_t i = s - head;  // list index of the slot
        if (s < head) i += C;
        uint32_t nr = 0;
        if (i < n_in) {
            float next;
            const uint64_t cnt = fw_emission_count(p_age[r], p_lea[r], F.life, N.n_start, N.n_end, N.n_count, &next);
            nr = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt;  // core.rs:490-498
            fw_st1w(wl, o * 4u, next);                               // other_particle.last_emitted_age[i] = next (core.rs:500)
        }
        uint32_t x = nr;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(x, d, 64);
            if (lane >= (uint32_t)d) x = (x + u < x) ? 0xFFFFFFFFu : x + u;  // saturating
        }
        s_n[r][tid] = nr, s_inc[r][wave][lane] = x;
        if (lane == 63) s_w[r][wave] = x;
    }
    __syncthreads();
    unsigned long long tot64 = 0;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int w = 0; w < NW; w++) tot64 += s_w[r][w];
    uint32_t tile_total = tot64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot64;
    // The child ring's bookkeeping workgroup books the entry's RNG serial once its look-back has seen the status word of every
    // parent tile: a tile must have READ the serial (and the child base) before it publishes.  A data dependency, as in fw_k_nest.
    {
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(cbase), d1 = __builtin_amdgcn_readfirstlane((uint32_t)serial0),
                       d2 = __builtin_amdgcn_readfirstlane((uint32_t)(serial0 >> 32));
        asm volatile("; fw_fifo_nest_parents: counters read before the tile publishes" : "+v"(tile_total) : "s"(d0), "s"(d1), "s"(d2));
    }
    const uint32_t tile = N.status_first + rank;
    const bool lb_needed = rank != 0u;
    if (lb_needed && tid == 0) __hip_atomic_store(&g.nest_status[tile], fw_pack_status(N.tag, FW_ST_AGG, tile_total), RLX, AGENT);
    uint32_t excl = 0;
    if (lb_needed) {
        bool timed_out = false;
        excl = fw_lookback<BLK, NW, LBW>(g.nest_status, N.status_first, tile, N.tag, N.spin_limit * 64u + 1024u, s_lb, &timed_out);
        // (no recount is possible: the tiles of lower rank have already advanced their parents' last_emitted_age.  They have lower
        // workgroup indices, so they are resident or done: the wait is bounded.)
        if (timed_out && tid == 0) fw_raise(g, 6u, F.seg, tile);
    }
    const unsigned long long incl64 = (unsigned long long)excl + tile_total;
    if (tid == 0)
        __hip_atomic_store(&g.nest_status[tile], fw_pack_status(N.tag, FW_ST_INCL, incl64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)incl64), RLX, AGENT);
    if (tile_total == 0u) return;  // (workgroup-uniform: most tiles of a ring hold parents past their emission window)
    // ---- children, wave-cooperatively (parent-major order), spawned and given their first update
    const FwOutWin Wc = fw_out_window(Fc.buf, Cc, 0u, Tc, 0u, 0u, true);  // (child capacity <= FW_RANGE_MAX_CAPACITY: 32-bit byte offsets, the host checks)
    float4 prot = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (pnospin) {
        const FwType &Tp = g.types[F.type_idx & ~FW_TYPE_IDX_NOSPIN];
        prot = make_float4(Tp.const_rot[0], Tp.const_rot[1], Tp.const_rot[2], Tp.const_rot[3]);
    }
    uint32_t run = excl;  // children of the entry before (round r, wave 0)
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        uint32_t woff = run;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if ((uint32_t)w < wave) woff += s_w[r][w];
            run += s_w[r][w];
        }
        const uint32_t tw = s_w[r][wave];  // wave-uniform
        if (tw == 0) continue;
        if (s_n[r][tid] != 0) {
            const uint32_t o16 = ((uint32_t)(r * BLK) + tid) * 16u;
            s_par[wave][0][lane] = fw_ld4w(w0, o16);
            s_par[wave][1][lane] = fw_ldc3w(w1, FW_CP(C), o16 / 4u, 0.0f);  // (the parent's velocity: core.rs:706-736)
            s_par[wave][2][lane] = pnospin ? prot : fw_ld4w(w2, o16);
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
                    if (s_inc[r][wave][mid] > c) hi = mid;
                    else lo = mid + 1;
                }
                const unsigned long long j = (unsigned long long)woff + c;  // child index within the entry
                const unsigned long long slotl = (unsigned long long)cbase + j;
                if (slotl < Cc) {  // (what does not fit is dropped; the child ring's bookkeeping reports FW_ERR_CAPACITY)
                    const float4 pq0 = s_par[wave][0][lo], pq1 = s_par[wave][1][lo], pq2 = s_par[wave][2][lo];
                    const FwSpawnOut so = fw_spawn_one(e, g.seed, serial0 + j, fw_v3{pq0.x, pq0.y, pq0.z}, fw_q4{pq2.x, pq2.y, pq2.z, pq2.w},
                                                       fw_v3{pq1.x, pq1.y, pq1.z}, N.speed, N.scale);
                    float age_new;
                    fw_survives(so.q0.w, a.dt, so.q3.w, &age_new);  // (the host keeps a step as long as the child lifetime off this path)
                    fw_integrate_store<true, -1, NT, true>(Tc, s_ckeys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, Wc,
                                                     fw_ring_slot(Fc.head, (uint32_t)slotl, Cc), nullptr, nullptr, nullptr, nullptr, false, true, false);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the wave's LDS rows are reused by the next round
    }
}

// WM: the optional planes this launch's particle types write (fw_integrate_store), or -1 = read from the type
#ifndef FW_FIFO_UNROLL
#define FW_FIFO_UNROLL 4
#endif
// TR: rounds per workgroup = ring tile / 256.  Four for the streaming instantiations; ONE for the colliding ones (FwFifoArgs::
// tile): a colliding particle is a few hundred dependent instructions of ray casts per sub-step, the launch is bound by how
// many waves work at once, not by memory -- the reference's own stress_test_collision (157k particles) is 154 workgroups of
// four rounds, fewer than the chip has CUs, or 615 of one.
// NEST: some Nested entry runs inside this launch (FwFifoArgs::nest): the parents' ring tiles run fw_fifo_nest_parents first
template <bool INST, int WM, int NT, bool COLL, int TR, bool NEST>
__device__ __forceinline__ void fw_update_fifo_body(const FwGlobals &g, const FwFifoArgs &a, const FwInlineOps &inl) {
    constexpr int BLK = FW_BLOCK;
    constexpr int NW = BLK / 64;
    constexpr int R = TR;  // (smaller ring tiles were measured for the streaming case: 2 rounds 26.7 us, 1 round 26.2 us, 4 rounds 24.6 us)
    constexpr uint32_t TILE = (uint32_t)(BLK * TR);
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
    if constexpr (NEST) {
        // a tile of a ring whose particles a Nested entry emits from: the entry's pass over the tile's own slots, before anything
        // of them is loaded for the update (workgroup-uniform branch)
        if (!spawner && F.nest != 0u && !(F.nest & FW_FIFO_NEST_CHILD)) {
            const FwFifoNest &N = a.nest[F.nest - 1u];
            // the tile's rank among the parents' tiles -- which ring tile it works on -- is a START ticket (fw_kernels.h)
            __shared__ uint32_t s_prank;
            if (FW_TICKETS) {
                if (tid == 0u) s_prank = atomicAdd(&g.nest_start[N.emit_slot], 1u) - N.ticket_base;
                __syncthreads();
            }
            const uint32_t prank = FW_TICKETS ? s_prank : tis - n_vt;
            pt = F.tile0 + prank;
            if (pt >= ring_tiles) pt -= ring_tiles;
            sbase = pt * TILE;
            fw_fifo_nest_parents<R, NT>(g, a, N, F, a.s[N.child], prank, sbase, n_in);
        }
    }
    const size_t sfirst = (size_t)sbase * 16u, cp = FW_CP(C);
    // (Q1 / Q3: component planes, fw_dev.h -- windows of the x plane, offsets of 4 bytes per slot.  A ring type has ONE lifetime value
    // (F.life) and its initial_scale matters to instance records and destroyed records only: neither `.w` is loaded by the streaming loop)
    const char *iw0 = buf + FW_OFF_Q0(C) + sfirst, *iw1 = buf + FW_OFF_Q1(C) + sfirst / 4u;
    const char *iw2 = buf + FW_OFF_Q2(C) + sfirst, *iw3 = buf + FW_OFF_Q3(C) + sfirst / 4u;
    auto ld1 = [&](uint32_t o16) -> float4 {  // velocity (+ initial_scale when this launch writes instance records)
        if constexpr (INST) return fw_ldc4w<NT == 2>(iw1, cp, o16 / 4u);
        else return fw_ldc3w<NT == 2>(iw1, cp, o16 / 4u, 0.0f);
    };
    // (rotation and angular velocity of a type that cannot turn: not even dummy loads -- a workgroup-uniform branch)
    auto ld3 = [&](uint32_t o16) -> float4 {
        float4 v = q3s;
        if (!nospin) v = fw_ldc3w<NT == 2>(iw3, cp, o16 / 4u, F.life);
        return v;
    };
    auto ld2 = [&](uint32_t o16) -> float4 {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        if (!nospin) v = fw_ld4w<NT == 2>(iw2, o16);
        return v;
    };
    if (!spawner && !defer) {
        q0c = fw_ld4w<NT == 2>(iw0, tid * 16u), q3c = ld3(tid * 16u);
        q1c = ld1(tid * 16u), q2c = ld2(tid * 16u);
        if constexpr (R > 1) {
            q0n = fw_ld4w<NT == 2>(iw0, i1), q3n = ld3(i1);
            q1n = ld1(i1), q2n = ld2(i1);
        }
    }
    if (defer) {
        uint32_t i0 = sbase - head;
        if (sbase < head) i0 += C;
        if (tis != 0u && !(i0 < n_tot || (i0 + TILE > C && n_tot != 0u))) return;
        q0c = fw_ld4w<NT == 2>(iw0, tid * 16u), q3c = ld3(tid * 16u);
        q1c = ld1(tid * 16u), q2c = ld2(tid * 16u);
        if constexpr (R > 1) {
            q0n = fw_ld4w<NT == 2>(iw0, i1), q3n = ld3(i1);
            q1n = ld1(i1), q2n = ld2(i1);
        }
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
    const FwOutWin W = fw_out_window(buf, C, sbase, T, 0u, 0u, true);
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
            fw_integrate_store<true, -1, NT, true>(T, s_keys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, W, s, rec, COLL ? &cpos : nullptr,
                                         COLL ? &cvel : nullptr, nullptr, false, true, CA.on);
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
                    const float4 q0 = fw_ld4w<NT == 2>(iw0, b16), q1 = fw_ldc4w<NT == 2>(iw1, cp, b16 / 4u), q2 = fw_ld4w<NT == 2>(iw2, b16 & m2);
                    const float4 q3 = nospin ? q3s : fw_ldc4w<NT == 2>(iw3, cp, b16 / 4u);
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
            float4 q0f = q0c, q3f = q3c, q1f = q1c, q2f = q2c;
            if constexpr (R > 1) {  // (a one-round workgroup has nothing to prefetch)
                q0f = fw_ld4w<NT == 2>(iw0, in_), q3f = ld3(in_);
                q1f = ld1(in_), q2f = ld2(in_);
            }
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
                    fw_st4w<NT == 2>(W.q0, b16, make_float4(q0c.x, q0c.y, q0c.z, age_new)), fw_stc3w<NT == 2>(W.q1, W.cp, b16 / 4u, q1c.x, q1c.y, q1c.z);
                    if (WM >= 0 ? (WM & 1) != 0 : W.wr5) fw_st4w<NT != 0>(W.q5, b16, q0c);
                    if (WM >= 0 ? (WM & 2) != 0 : W.wr6) fw_st4w<NT != 0>(W.q6, b16, q1c);
                    if (WM >= 0 ? (WM & 4) != 0 : T.sc_kind != 0) fw_st1w<NT != 0>(W.s4, (s - W.first) * 4u, q1c.w);
                } else {
                    fw_integrate_store<true, WM, NT, true>(T, s_keys, a.dt, q0c, q1c, q2c, q3c, age_new, W, s, rec, COLL ? &cpos : nullptr,
                                                 COLL ? &cvel : nullptr, nullptr, false, i >= full_from, CA.on, INST ? FW_W_MEM : FW_W_MEM_LAZY);
                }
            }
            fw_fifo_inst_out<INST, NT == 2>(F, inst, s_inst_wave, rec, lane, m, alive, i - n_dead);
            q0c = q0n, q1c = q1n, q2c = q2n, q3c = q3n;
            q0n = q0f, q1n = q1f, q2n = q2f, q3n = q3f;
        }
    }
    if (__any(bad) && lane == 0) fw_raise(g, 4u, F.seg, blockIdx.x);
    // a ring that RECEIVES the children of a Nested entry run inside this launch: the entry's total = the inclusive prefix of the
    // parents' last tile, read from the same status words (every parent tile has a lower workgroup index: resident or done)
    uint32_t nest_total = 0u;
    if constexpr (NEST) {
        if (tis == 0u && (F.nest & FW_FIFO_NEST_CHILD) != 0u) {  // (workgroup-uniform)
            __shared__ uint32_t s_lbc[2 * 4 * NW];
            const FwFifoNest &N = a.nest[(F.nest & ~FW_FIFO_NEST_CHILD) - 1u];
            bool timed_out = false;
            if (N.n_ptiles)
                nest_total = fw_lookback<BLK, NW, 4>(g.nest_status, N.status_first, N.status_first + N.n_ptiles, N.tag,
                                                     N.spin_limit * 64u + 1024u, s_lbc, &timed_out);
            if (timed_out && tid == 0) fw_raise(g, 6u, F.seg, N.status_first + N.n_ptiles);
        }
    }
    if (tis == 0 && tid == 0) {
        const uint32_t oidx = (a.parity ^ 1u) * g.max_seg + F.seg;
        if (F.mat ? (F.n_in != 0xFFFFFFFFu && F.n_in != n_in) : g.count[sidx] != n_in)
            fw_raise(g, 5u, F.seg, n_in);
        uint32_t n_all = n_tot, added = c_new;
        if constexpr (NEST) {
            if ((F.nest & FW_FIFO_NEST_CHILD) != 0u) {
                const FwFifoNest &N = a.nest[(F.nest & ~FW_FIFO_NEST_CHILD) - 1u];
                const uint32_t room = C - min(C, n_tot);  // (the parents' tiles dropped the children beyond it)
                if (nest_total > room) fw_flag(g, FW_ERR_CAPACITY);
                const uint32_t take = min(nest_total, room);
                g.emit_serial[N.emit_slot] += (unsigned long long)nest_total;  // (every parent tile read it before it published)
                n_all += take, added += take;
            }
        }
        if (F.report) *F.report = ((unsigned long long)a.epoch << 32) | added;
        const uint32_t nc = n_all - min(n_dead, n_all);
        g.count[oidx] = nc;
        g.spawned[oidx] = 0;
        g.appended[oidx] = 0;
        g.ndestroyed[F.seg] = n_all - nc;
        if (a.host_counts) a.host_counts[F.seg] = ((unsigned long long)a.epoch << 32) | nc;
        if (a.live_out) atomicAdd(a.live_out, (unsigned long long)nc);
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats + (F.seg % FW_STAT_SLOTS), (unsigned long long)n_all);
    }
}

// (FW_FIFO_WAVES: compile-time A/B -- a minimum of waves per SIMD for the plain streaming instantiations: configs[1]'s grid is 1042
// workgroups, 18 more than the 1024 slots that 4 workgroups per CU give)
#ifndef FW_FIFO_WAVES
#define FW_FIFO_WAVES 1
#endif
template <bool INST, int WM, int NT = 0, bool COLL = false, int TR = FW_ROUNDS>
__global__ __launch_bounds__(FW_BLOCK) __attribute__((amdgpu_waves_per_eu((!INST && !COLL && TR == FW_ROUNDS) ? FW_FIFO_WAVES : 1)))
void fw_k_update_fifo(FwGlobals g, FwFifoArgs a, FwInlineOps inl) {
    fw_update_fifo_body<INST, WM, NT, COLL, TR, false>(g, a, inl);
}
// ... with Nested entries inside the launch (FwFifoNest).  A kernel of its own so that the plain instantiations keep their code
// and their register budget; pinned at 4 waves per SIMD (the nest phase took the four-round form to 133 VGPRs: the bulk of such
// a launch is the child ring's streaming tiles, which want the fourth workgroup per CU)
// (FW_NEST_WAVES: compile-time A/B -- `tools/build_variant.sh nest3 -DFW_NEST_WAVES=3` is the form without scratch,
// profiles/r06/nest_waves_ab.txt)
#ifndef FW_NEST_WAVES
#define FW_NEST_WAVES 4
#endif
template <int NT, int TR>
__global__ __launch_bounds__(FW_BLOCK) __attribute__((amdgpu_waves_per_eu(FW_NEST_WAVES))) void fw_k_update_fifo_nest(FwGlobals g, FwFifoArgs a, FwInlineOps inl) {
    fw_update_fifo_body<false, -1, NT, false, TR, true>(g, a, inl);
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
// TR: rounds per workgroup of the OLD and YOUNG roles (their tile = TR * 256 slots).  Four for launches that stream; ONE for
// launches too small to fill the chip with four-round workgroups and for colliding launches (FwRangeArgs::small_tiles; as for
// FIFO rings: a single range ring of 156k particles is ~190 four-round workgroups, its old part a chain of 31 tiles of 1024
// particles each of which waits for the counts of the ones before -- 17 us per frame where the FIFO ring of the same size
// takes 8).
// YRP: rounds of a YOUNG workgroup of a four-round launch (FwRangeArgs::young_rounds): FW_RANGE_YR, or 2 -- young tiles of 512
// slots -- for launches of large segments (round 3 measured the fixed choices: +4 % at 16M particles in 64 Ki-particle segments,
// -18 % on 8192-particle segments; the host now chooses per launch from the mean segment size)
template <bool ALLNOSPIN, bool INST, int NT, bool COLL = false, int TR = FW_ROUNDS, int YRP = FW_RANGE_YR>
// (launches with a type that can turn, four-round tiles: pinned at 4 waves per SIMD -- with the OLD tiles' rotation planes parked in
// LDS the kernel is 6 registers past the step and fits when asked to; not the forms that also write instance records: their 50 KB of
// LDS allow three workgroups per CU anyway)
#ifndef FW_RANGE_SPIN_WAVES
#define FW_RANGE_SPIN_WAVES 4  // (1: no pin -- 134 VGPRs, 3 waves per SIMD; the A/B of profiles/r05/range_ab.txt)
#endif
__global__ __launch_bounds__(FW_BLOCK) __attribute__((amdgpu_waves_per_eu((!ALLNOSPIN && !INST && !COLL && TR == FW_ROUNDS) ? FW_RANGE_SPIN_WAVES : 1)))
void fw_k_update_range(FwGlobals g, FwRangeArgs a) {
    constexpr int BLK = FW_BLOCK;
    constexpr int NW = BLK / 64;
    constexpr int R = TR;
    constexpr int LBW = 4;
    constexpr uint32_t TILE = (uint32_t)(BLK * TR);
    __shared__ __attribute__((aligned(16))) float s_keys[FW_KEYS_MAX];
    __shared__ __attribute__((aligned(16))) float4 s_inst[INST ? NW * 256 : 1];  // per wave: 64 records of 4 float4
    __shared__ uint32_t s_cnt[R][NW];
    __shared__ uint32_t s_lb[2 * LBW * NW];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const FwRangeDesc &D = a.desc[blockIdx.x];  // block-uniform: scalar loads
    const uint32_t seg = D.seg, role = D.role_k >> 30;
    uint32_t k = D.role_k & 0x3FFFFFFFu;
    // an OLD workgroup waits for the OLD workgroups of lower rank: its rank is a START ticket (fw_kernels.h), requested before
    // anything else so that it travels together with the record and the counters
    uint32_t ticket = 0u;
    if (FW_TICKETS && role == FW_RANGE_OLD && threadIdx.x == 0u) ticket = atomicAdd(&g.range_ticket[seg], 1u);
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
    const size_t cp = FW_CP(C);  // (Q1 / Q3: component planes, fw_dev.h)
    // lifetimes: a plane of their own for a type that cannot turn (FwOutWin::lf), the w plane of Q3 otherwise
    const char *pl = nospin ? buf + FW_OFF_L(C, Sp->n_lplanes) : p3 + 3 * cp;
    char *inst = INST ? Sp->inst : nullptr;
    const uint32_t inst_cap = INST ? Sp->inst_cap : 0u;
    float4 *s_inst_wave = s_inst + (INST ? wave * 256u : 0u);

    if (role == FW_RANGE_YOUNG) {
        // ---- in place: a lane owns its slot from load to store
        constexpr int YR = TR == FW_ROUNDS ? YRP : TR;
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
        const fw_rsrc r0 = fw_make_rsrc(p0 + (size_t)w_lo * 16u, w_n * 16u);
        // (velocity: one descriptor per component plane; initial_scale is loaded only when the launch writes instance records -- a
        // zero-length window otherwise -- whoever else needs it reads it in fw_integrate_store: FW_W_MEM_LAZY)
        const fw_rsrc r1x = fw_make_rsrc(p1 + (size_t)w_lo * 4u, w_n * 4u), r1y = fw_make_rsrc(p1 + cp + (size_t)w_lo * 4u, w_n * 4u);
        const fw_rsrc r1z = fw_make_rsrc(p1 + 2 * cp + (size_t)w_lo * 4u, w_n * 4u);
        const fw_rsrc r1w = fw_make_rsrc(p1 + 3 * cp + (size_t)w_lo * 4u, INST ? w_n * 4u : 0u);
        // (the lifetime: every tile of a type somebody evaluates the scale of; otherwise the boundary tile alone -- FW_TYPE_IDX_NOLIFE)
        const bool need_lf = INST || k == 0u || !(D.type_idx & FW_TYPE_IDX_NOLIFE);
        const fw_rsrc rl = fw_make_rsrc(pl + (size_t)w_lo * 4u, need_lf ? w_n * 4u : 0u);
        auto ldq1 = [&](uint32_t ir) -> float4 {
            const uint32_t i4 = ir / 4u;
            float w = 0.0f;
            if constexpr (INST) w = fw_ldb1<NT == 2>(r1w, i4);
            return make_float4(fw_ldb1<NT == 2>(r1x, i4), fw_ldb1<NT == 2>(r1y, i4), fw_ldb1<NT == 2>(r1z, i4), w);
        };
        constexpr int WMODE = INST ? FW_W_MEM : FW_W_MEM_LAZY;
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
                q0a[r] = fw_ldb4<NT == 2>(r0, ir), lfa[r] = fw_ldb1<NT == 2>(rl, ir / 4u), q1a[r] = ldq1(ir);
            }
            const FwType T = g.types[D.type_idx & FW_TYPE_IDX_MASK];
            const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & FW_TYPE_IDX_MASK);
            if (tid < keys_len) s_keys[tid] = key0;
            for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
            __syncthreads();
            const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes, true);
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
                    q0a[r % PF] = fw_ldb4<NT == 2>(r0, ir), lfa[r % PF] = fw_ldb1<NT == 2>(rl, ir / 4u), q1a[r % PF] = ldq1(ir);
                }
                float age_new;
                const bool surv = fw_survives(q0v.w, a.dt, q3v.w, &age_new);
                bad |= need_lf && mine && !surv;
                const unsigned long long mi = (INST && inst != nullptr) ? __ballot(mine) : 0ull;
                float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(mi) * 4u : nullptr;
                fw_v3 cpos, cvel;
                fw_coll_step<COLL>(g, CA, mine, a.dt, q0v, q1v, &cpos, &cvel);
                if (mine)
                    fw_integrate_store<true, -1, NT, true>(T, s_keys, a.dt, q0v, q1v, q3v, q3v, age_new, W, s, rec, COLL ? &cpos : nullptr,
                                                     COLL ? &cvel : nullptr, nullptr, false, yi >= y_full, CA.on, WMODE);
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
        const fw_rsrc r2 = fw_make_rsrc(p2 + (size_t)w_lo * 16u, m2 ? w_n * 16u : 0u);
        const fw_rsrc r3x = fw_make_rsrc(p3 + (size_t)w_lo * 4u, m2 ? w_n * 4u : 0u), r3y = fw_make_rsrc(p3 + cp + (size_t)w_lo * 4u, m2 ? w_n * 4u : 0u);
        const fw_rsrc r3z = fw_make_rsrc(p3 + 2 * cp + (size_t)w_lo * 4u, m2 ? w_n * 4u : 0u);
        const fw_rsrc rlf = rl;  // (the lifetime: `pl` is the w plane of Q3 for a type that can turn)
        auto ldq3 = [&](uint32_t ir) -> float4 {
            if constexpr (ALLNOSPIN) return make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            else return make_float4(fw_ldb1<NT == 2>(r3x, ir / 4u), fw_ldb1<NT == 2>(r3y, ir / 4u), fw_ldb1<NT == 2>(r3z, ir / 4u), 0.0f);
        };
        const uint32_t i0 = woff(0), i1 = woff(min(1, YR - 1));
        q0c = fw_ldb4<NT == 2>(r0, i0), q3c = ldq3(i0), lfc = fw_ldb1<NT == 2>(rlf, i0 / 4u);
        q1c = ldq1(i0), q2c = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r2, i0);
        q0n = fw_ldb4<NT == 2>(r0, i1), q3n = ldq3(i1), lfn = fw_ldb1<NT == 2>(rlf, i1 / 4u);
        q1n = ldq1(i1), q2n = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r2, i1);
        const FwType T = g.types[D.type_idx & FW_TYPE_IDX_MASK];
        const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & FW_TYPE_IDX_MASK);
        if (tid < keys_len) s_keys[tid] = key0;
        for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
        __syncthreads();
        FW_STAMP(2, T.flags);  // type record + keys in LDS
        const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes, true);
        bool bad = false;
        FW_STAMP(6, __float_as_uint(q0c.w) | __float_as_uint(q1c.w));  // the first round's particles have arrived
#pragma unroll
        for (int r = 0; r < YR; r++) {
            const uint32_t s = sbase + r * BLK + tid;
            const uint32_t in_ = woff(min(r + 2, YR - 1));  // two rounds ahead (the last re-read)
            const float4 q0f = fw_ldb4<NT == 2>(r0, in_), q3f = ldq3(in_);
            const float lff = fw_ldb1<NT == 2>(rlf, in_ / 4u);
            const float4 q1f = ldq1(in_), q2f = fw_ldb4_opt<ALLNOSPIN, NT == 2>(r2, in_);
            q3c.w = lfc;  // (a type that cannot turn: x, y, z came back as zeros -- a zero-length window)
            uint32_t yi = s - b;  // index within the young part
            if (s < b) yi += C;
            const bool mine = yi < y_exist;
            float age_new;
            const bool surv = fw_survives(q0c.w, a.dt, q3c.w, &age_new);
            bad |= need_lf && mine && !surv;  // the host's cohort ages say nobody young can die
            const unsigned long long mi = (INST && inst != nullptr) ? __ballot(mine) : 0ull;
            float4 *rec = (INST && inst != nullptr) ? s_inst_wave + fw_lane_prefix(mi) * 4u : nullptr;
            fw_v3 cpos, cvel;
            fw_coll_step<COLL>(g, CA, mine, a.dt, q0c, q1c, &cpos, &cvel);
            if (mine)
                fw_integrate_store<true, -1, NT, true>(T, s_keys, a.dt, q0c, q1c, q2c, q3c, age_new, W, s, rec, COLL ? &cpos : nullptr,
                                                 COLL ? &cvel : nullptr, nullptr, false, yi >= y_full, CA.on, WMODE);
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
        const FwType T = g.types[D.type_idx & FW_TYPE_IDX_MASK];
        const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & FW_TYPE_IDX_MASK);
        if (tid < keys_len) s_keys[tid] = key0;
        for (uint32_t i = tid + BLK; i < keys_len; i += BLK) s_keys[i] = g.keys[keys_off + i];
        __syncthreads();
        const uint32_t kk = k * BLK + tid;
        const bool is_new = kk < n_spawn;
        if (kk < n_spawn_h && kk == n_spawn) fw_flag(g, FW_ERR_CAPACITY);
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
            const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes, true);
            fw_v3 cpos, cvel;
            fw_coll_step<COLL>(g, CA, true, a.dt, so.q0, so.q1, &cpos, &cvel);
            fw_integrate_store<false, -1, NT, true>(T, s_keys, a.dt, so.q0, so.q1, so.q2, so.q3, age_new, W, s, rec, COLL ? &cpos : nullptr,
                                              COLL ? &cvel : nullptr, nullptr, false, false, CA.on);
            if (Sp->n_lplanes) fw_init_last_emitted(g, *Sp, buf, s, g.emits[op.emit].emission_index, so.q3.w);  // (other particles' entries emit from it)
        }
        if (INST) fw_range_inst_out<NT == 2>(inst, inst_cap, s_inst_wave, rec, lane, mi, n_old_in + y_exist + kk, false);
        return;
    }

    // ---- OLD: in-place compaction towards the young part.  Distance d from the young part: slot = b - 1 - d.
    if (FW_TICKETS) {
        __shared__ uint32_t s_rank;
        if (tid == 0u) s_rank = ticket - Rc.ticket_base;  // (every provisioned OLD workgroup of the segment takes exactly one per launch)
        __syncthreads();
        k = s_rank;
    }
    // (ONE workgroup walking the few tiles of a small old part itself -- no status words, no waiting, no provisioned-but-idle
    // workgroups -- was built and measured in round 4: the tile loop costs the kernel 9-35 VGPRs, and even at equal occupancy
    // one GPU's share of configs[4] ran 86.4 us against 85.6 with the tiles in parallel: profiles/r04/range_seq_old_ab.txt)
    const uint32_t base = k * TILE;
    if (k == 0u && tid == 0u && n_old_in > D.n_old * TILE) {  // the host's bound of the old part was not one (internal error)
        fw_flag(g, FW_ERR_CAPACITY);
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
            if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats + (seg % FW_STAT_SLOTS), (unsigned long long)nc);
        }
        return;
    }
    const uint32_t lim = min(base + TILE, n_old_in);
    const uint32_t bm1 = b + C - 1u;
    // The tile holds position + age and velocity + initial_scale of its R rounds in registers (and each particle's lifetime);
    // rotation and angular velocity of a type that can turn are PARKED IN LDS until the store loop (round 5): with them in
    // registers as well the kernels of launches with such a type took 150-157 VGPRs -- 3 waves per SIMD for every workgroup of
    // the launch, the streaming YOUNG ones included -- against 116 for the launches in which nothing turns.  A lane writes and
    // reads its own entries only; the writes sit before the barrier in front of the published count, so the loads they consume
    // have returned by then.
    __shared__ __attribute__((aligned(16))) float4 s_q2[ALLNOSPIN ? 1 : TILE], s_q3[ALLNOSPIN ? 1 : TILE];
    float4 q0[R], q1[R];
    float lifev[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t d = min(base + r * BLK + tid, lim - 1u);
        uint32_t s = bm1 - d;  // in [0, 2 C)
        if (s >= C) s -= C;
        q0[r] = fw_ld4w<NT == 2>(p0, s * 16u);
        const float lf = fw_ld1w<NT == 2>(pl, s * 4u);  // (the lifetime plane, or the w plane of Q3)
        float4 q3v = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        if constexpr (!ALLNOSPIN) q3v = fw_ldc3w<NT == 2>(p3, cp, (s * 4u) & m2, lf);
        q1[r] = fw_ldc4w<NT == 2>(p1, cp, s * 4u);  // (compacted: initial_scale moves with the particle)
        const float4 q2v = fw_ld4w_opt<ALLNOSPIN, NT == 2>(p2, (s * 16u) & m2);
        lifev[r] = lf;
        if constexpr (!ALLNOSPIN) {
            if (!nospin) s_q2[r * BLK + tid] = q2v, s_q3[r * BLK + tid] = q3v;  // (workgroup-uniform branch)
        }
    }
    // (rotation / angular velocity + lifetime of round r: from LDS, or -- a type that cannot turn -- nothing and the lifetime)
    auto old_q2 = [&](int r) -> float4 {
        if constexpr (!ALLNOSPIN) {
            if (!nospin) return s_q2[r * BLK + tid];
        }
        return make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    };
    auto old_q3 = [&](int r) -> float4 {
        if constexpr (!ALLNOSPIN) {
            if (!nospin) return s_q3[r * BLK + tid];
        }
        return make_float4(0.0f, 0.0f, 0.0f, lifev[r]);
    };
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
    const FwType T = g.types[D.type_idx & FW_TYPE_IDX_MASK];
    const FwCollArm CA = fw_coll_arm<COLL>(g, D.type_idx & FW_TYPE_IDX_MASK);
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
        const bool alive = valid && fw_survives(q0[r].w, a.dt, lifev[r], &age_new[r]);
        m[r] = __ballot(alive);
        uint32_t c = (uint32_t)__popcll(m[r]);
        asm volatile("; fw_k_update_range: input held before the count is published"
                     : "+v"(c) : "v"(q0[r].x), "v"(q1[r].x), "v"(q1[r].y), "v"(q1[r].z), "v"(q1[r].w), "v"(lifev[r]));
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
    const FwOutWin W = fw_out_window(buf, C, 0u, T, 0u, Sp->n_lplanes, true);
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
            fw_integrate_store<false, -1, NT, true>(T, s_keys, a.dt, q0[r], q1[r], old_q2(r), old_q3(r), age_new[r], W, s, rec, COLL ? &cpos : nullptr,
                                              COLL ? &cvel : nullptr, nullptr, false, false, CA.on);
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
            const float ap = q0[r].w / lifev[r];
            float bc[4], em[4];
            fw_gradient_sample(T.bc_kind, T.bc_n, s_keys + T.o_bc_t, s_keys + T.o_bc_v, ap, bc);
            fw_gradient_sample(T.em_kind, T.em_n, s_keys + T.o_em_t, s_keys + T.o_em_v, ap, em);
            const float sc = q1[r].w * fw_curve_sample(T.sc_kind, T.sc_n, s_keys, s_keys + T.o_sc_v, ap);
            const uint32_t dead_rank = d - od;  // the dead nearer to the young part
            fw_store_destroyed_vals(Sp->destroyed, (size_t)(C - 1u - dead_rank), T, q0[r], q1[r], old_q2(r), old_q3(r), q0[r].w + a.dt, bc, em, sc);
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
        if (!FW_DBG(a.dbg, 128u)) atomicAdd(g.stats + (seg % FW_STAT_SLOTS), (unsigned long long)(n_old_in + y_exist + n_spawn));
    }
}

// ---- launch wrappers

hipError_t fw_launch_update_fifo(hipStream_t s, const FwGlobals &g, const FwFifoArgs &a, const FwInlineOps &inl,
                                 uint32_t total_tiles, int nt, hipEvent_t e0, hipEvent_t e1) {
    if (!total_tiles || !a.n_segs) return hipErrorInvalidValue;
    const dim3 grid(total_tiles), block(FW_BLOCK);
    if (a.n_nest) {  // Nested entries inside the launch (FwFifoNest): generic write mask; the host keeps instance buffers and colliders out
        // (... and such a launch holds a ring whose count only the device knows: never laid out on one-round tiles)
        if (a.any_inst || a.any_coll || a.small_tiles) return hipErrorInvalidValue;
        if (nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo_nest<2, FW_ROUNDS>), grid, block, s, e0, e1, g, a, inl);
        else if (nt == 1)
            FW_LAUNCH_T((fw_k_update_fifo_nest<1, FW_ROUNDS>), grid, block, s, e0, e1, g, a, inl);
        else
            FW_LAUNCH_T((fw_k_update_fifo_nest<0, FW_ROUNDS>), grid, block, s, e0, e1, g, a, inl);
        return hipGetLastError();
    }
    if (a.any_coll) {  // some ring of the launch collides (FwCollArm): generic write mask, plain or fully non-temporal
        // (one round per workgroup -- ring tiles of FW_FIFO_COLL_TILE slots: the host laid the launch out on that grid)
        if (a.any_inst && nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 2, true, 1>), grid, block, s, e0, e1, g, a, inl);
        else if (a.any_inst)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 0, true, 1>), grid, block, s, e0, e1, g, a, inl);
        else if (nt == 2)
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 2, true, 1>), grid, block, s, e0, e1, g, a, inl);
        else
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 0, true, 1>), grid, block, s, e0, e1, g, a, inl);
        return hipGetLastError();
    }
    if (a.small_tiles) {  // a launch of a few hundred four-round workgroups at most: one round each instead (generic write mask)
        if (a.any_inst)
            FW_LAUNCH_T((fw_k_update_fifo<true, -1, 0, false, 1>), grid, block, s, e0, e1, g, a, inl);
        else
            FW_LAUNCH_T((fw_k_update_fifo<false, -1, 0, false, 1>), grid, block, s, e0, e1, g, a, inl);
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
    // (round 6: every type leaves scale and colours to its readers -- write mask 0 is what the product launches; the other seven
    // compile-time masks of rounds 3-5 served types that stored those planes: FW_DERIVED=0 / 1 and colliding types now take the
    // generic form -- 14 instantiations and a third of the library's build time less)
    switch (a.write_mask) {
        FW_FIFO_CASE(0)
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
        if (a.any_coll) {  // some range ring of the launch collides (FwCollArm): one-round tiles, the host laid the launch out on them
            if (a.any_inst) {
                if (all_nospin) FW_LAUNCH_T((fw_k_update_range<true, true, NT, true, 1>), grid, block, s, e0, e1, g, a);
                else FW_LAUNCH_T((fw_k_update_range<false, true, NT, true, 1>), grid, block, s, e0, e1, g, a);
            } else if (all_nospin) {
                FW_LAUNCH_T((fw_k_update_range<true, false, NT, true, 1>), grid, block, s, e0, e1, g, a);
            } else {
                FW_LAUNCH_T((fw_k_update_range<false, false, NT, true, 1>), grid, block, s, e0, e1, g, a);
            }
            return;
        }
    }
    if constexpr (NT == 0) {
        if (a.small_tiles) {  // a small launch: one round per workgroup
            if (a.any_inst) {
                if (all_nospin) FW_LAUNCH_T((fw_k_update_range<true, true, 0, false, 1>), grid, block, s, e0, e1, g, a);
                else FW_LAUNCH_T((fw_k_update_range<false, true, 0, false, 1>), grid, block, s, e0, e1, g, a);
            } else if (all_nospin) {
                FW_LAUNCH_T((fw_k_update_range<true, false, 0, false, 1>), grid, block, s, e0, e1, g, a);
            } else {
                FW_LAUNCH_T((fw_k_update_range<false, false, 0, false, 1>), grid, block, s, e0, e1, g, a);
            }
            return;
        }
    }
    if (a.young_rounds == 2u && !a.any_inst) {  // young tiles of 512 slots (the host laid the launch out on them; never with records)
        if (all_nospin) FW_LAUNCH_T((fw_k_update_range<true, false, NT, false, FW_ROUNDS, 2>), grid, block, s, e0, e1, g, a);
        else FW_LAUNCH_T((fw_k_update_range<false, false, NT, false, FW_ROUNDS, 2>), grid, block, s, e0, e1, g, a);
        return;
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
    if (a.small_tiles && !a.any_coll) nt = 0;  // (a small launch fits the cache many times over)
    if (nt == 2) fw_launch_update_range_t<2>(s, g, a, all_nospin, e0, e1);
    else if (nt == 1) fw_launch_update_range_t<1>(s, g, a, all_nospin, e0, e1);
    else fw_launch_update_range_t<0>(s, g, a, all_nospin, e0, e1);
    return hipGetLastError();
}

