// fw_k_nested.hip -- spawn_particles on the device: fw_k_spawn (materialised Global emission) and fw_k_nest (Nested emission, one launch per level)
// (gfx950 only; device helpers in fw_dev.h, launch interface in fw_kernels.h)
#include "fw_dev.h"

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
        if (op.n > room) fw_flag(g, FW_ERR_CAPACITY);
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
    const uint32_t wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t oi = 0;
    FwNestOp op = inl.ops[0];
    if (ops) {
        oi = fw_find_nest_op(ops, n_ops, wg);
        op = ops[oi];
    } else {
#pragma unroll  // constant indices: the record stays in scalar registers (a dynamic index would go through scratch)
        for (uint32_t i = 1; i < FW_INLINE_OPS; i++)
            if (i < n_ops && inl.ops[i].first_tile <= wg) oi = i, op = inl.ops[i];
    }
    const uint32_t sidx = parity * g.max_seg + op.parent_seg, cidx = parity * g.max_seg + op.child_seg;
    // which tile of the op's parents this workgroup takes -- its rank in the op's look-back -- is a START ticket (fw_kernels.h):
    // every workgroup of the op takes exactly one, whoever holds a lower one has started
    __shared__ uint32_t s_rank;
    if (FW_TICKETS) {
        if (threadIdx.x == 0u) s_rank = atomicAdd(&g.nest_start[op.emit_slot], 1u) - op.ticket_base;
        __syncthreads();
    }
    const uint32_t tile = FW_TICKETS ? op.first_tile + s_rank : wg;
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
        p_life[r] = ((op.parent_nospin & 1u) != 0u && op.parent_life_plane == 0xFFFFFFFFu)
                        ? op.parent_life_const
                        : fw_load_lifetime(op.parent_buf, op.parent_cap, op.parent_life_plane, ci, (op.parent_nospin & 1u) != 0u, (op.parent_nospin & 2u) != 0u);
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
                s_par[wave][1][lane] = fw_ldq(pb + FW_OFF_Q1(PC), PC, ps, (op.parent_nospin & 2u) != 0u);  // (the parent's velocity; a ring's Q1: component planes)
                s_par[wave][2][lane] = (op.parent_nospin & 1u) ? make_float4(op.parent_rot[0], op.parent_rot[1], op.parent_rot[2], op.parent_rot[3])
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
            fw_flag(g, FW_ERR_CAPACITY);
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

// ---- launch wrappers

hipError_t fw_launch_spawn(hipStream_t s, const FwGlobals &g, const FwOp *d_ops, const FwOp *h_ops, uint32_t n_ops,
                           uint32_t total_blocks, uint32_t parity) {
    if (!n_ops || !total_blocks) return hipSuccess;
    FwInlineOps io{};  // (on the stack: contexts may be driven from different threads; the launch copies its arguments)
    if (!d_ops)
        for (uint32_t i = 0; i < n_ops && i < FW_INLINE_OPS; i++) io.ops[i] = h_ops[i];
    hipLaunchKernelGGL(fw_k_spawn, dim3(total_blocks), dim3(FW_BLOCK), 0, s, g, io, d_ops, n_ops, parity);
    return hipGetLastError();
}

hipError_t fw_launch_nested(hipStream_t s, const FwGlobals &g, const FwNestOp *d_ops, const FwNestOp *h_ops, uint32_t n_ops,
                            uint32_t total_tiles, uint32_t parity, uint32_t tag, uint32_t spin_limit, uint32_t dbg) {
    if (!n_ops || !total_tiles) return hipSuccess;
    FwNestInline io{};
    if (!d_ops)
        for (uint32_t i = 0; i < n_ops && i < FW_INLINE_OPS; i++) io.ops[i] = h_ops[i];
    hipLaunchKernelGGL(fw_k_nest, dim3(total_tiles), dim3(FW_BLOCK), 0, s, g, io, d_ops, n_ops, parity, tag, spin_limit, dbg);
    return hipGetLastError();
}

