// fw_engine_mem.cpp -- device / pinned memory of a context, exact counts, the error words of the update kernels
// (host engine of libfirework_hip.so: fw_engine.h lists its translation units; there is no CPU simulation path in this library)
#include "fw_engine.h"

namespace fwh {

fw_status sync(fw_ctx *ctx) {
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side_dirty) {
        FW_HIP(ctx, hipStreamSynchronize(ctx->fifo_stream));
        ctx->side_dirty = false;
    }
    ctx->main_reads_ring = false;
    return FW_OK;
}

// before work that reads ring data is enqueued on the main stream without a synchronisation: the main stream waits for
// the ring launches on the side stream (and the next ring launch will wait for that work)
fw_status join_side(fw_ctx *ctx) {
    if (ctx->side_dirty) {
        FW_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->fifo_stream));
        FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        ctx->side_dirty = false;
    }
    ctx->main_reads_ring = true;
    return FW_OK;
}

// grows the [2][max_seg] bookkeeping arrays and the snapshot ring
fw_status ensure_max_seg(fw_ctx *ctx, uint32_t need) {
    if (need <= ctx->max_seg) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    uint32_t nmax = std::max<uint32_t>(need, ctx->max_seg ? ctx->max_seg * 2 : 1024);
    auto regrow2 = [&](uint32_t *&p) -> fw_status {
        uint32_t *np = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&np, 2ull * nmax * sizeof(uint32_t)));
        FW_HIP(ctx, fw_memset_done(np, 0, 2ull * nmax * sizeof(uint32_t)));
        if (p) {
            for (int r = 0; r < 2; r++)
                FW_HIP(ctx, hipMemcpy(np + (size_t)r * nmax, p + (size_t)r * ctx->max_seg,
                                      ctx->max_seg * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            FW_HIP(ctx, hipFree(p));
        }
        p = np;
        return FW_OK;
    };
    if ((st = regrow2(ctx->g.count))) return st;
    if ((st = regrow2(ctx->g.spawned))) return st;
    if ((st = regrow2(ctx->g.appended))) return st;
    if ((st = regrow2(ctx->g.rold))) return st;
    {
        uint32_t *np = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&np, (size_t)nmax * sizeof(uint32_t)));
        FW_HIP(ctx, fw_memset_done(np, 0, (size_t)nmax * sizeof(uint32_t)));
        if (ctx->g.ndestroyed) {
            FW_HIP(ctx, hipMemcpy(np, ctx->g.ndestroyed, ctx->max_seg * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            FW_HIP(ctx, hipFree(ctx->g.ndestroyed));
        }
        ctx->g.ndestroyed = np;
    }
    {
        unsigned long long *nh = nullptr;
        FW_HIP(ctx, hipHostMalloc((void **)&nh, (size_t)kSnapRing * nmax * sizeof(unsigned long long), hipHostMallocDefault));
        memset(nh, 0, (size_t)kSnapRing * nmax * sizeof(unsigned long long));
        if (ctx->h_snap) FW_HIP(ctx, hipHostFree(ctx->h_snap));
        ctx->h_snap = nh;
        for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;
    }
    {
        uint32_t *np = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&np, (size_t)nmax * sizeof(uint32_t)));
        FW_HIP(ctx, fw_memset_done(np, 0, (size_t)nmax * sizeof(uint32_t)));
        if (ctx->g.range_ticket) {
            FW_HIP(ctx, hipMemcpy(np, ctx->g.range_ticket, ctx->max_seg * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            FW_HIP(ctx, hipFree(ctx->g.range_ticket));
        }
        ctx->g.range_ticket = np;
    }
    if (ctx->d_segids) FW_HIP(ctx, hipFree(ctx->d_segids));
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_segids, (size_t)nmax * sizeof(uint32_t)));
    ctx->max_seg = nmax;
    ctx->g.max_seg = nmax;
    return FW_OK;
}

// tiles the update grid must cover for a segment: the live region in tiles of FW_TILE, plus this frame's new
// particles in tiles of FW_VTILE (fw_k_update's tiling of the index space)
uint32_t seg_live_tiles(const SegHost &s) {
    const uint32_t live_ub = s.nested_fed ? s.capacity : std::min(s.ub - std::min(s.ub, s.frame_spawn), s.capacity);
    return (live_ub + FW_TILE - 1) / FW_TILE;
}
uint32_t seg_tiles(const SegHost &s, uint32_t vt_rounds) {
    if (!s.in_use || s.ring() || s.small) return 0;  // (rings and small types have their own launches: fw_k_update_fifo / _range / _small)
    const uint32_t vtile = vt_rounds * FW_VTILE;
    if (!s.nested_fed && s.frame_spawn <= FW_VTILE) {
        // At most one round of new particles: they ride in the last live tile whenever it has room for them (both
        // update kernels apply the same rule), otherwise they take one tile of their own right behind it -- either
        // way ceil((live + new) / FW_TILE) tiles, for any live count up to the bound.  Thousands of small emitters
        // then cost ONE workgroup each instead of three (every workgroup pays ~5 us of launch-time latencies).
        const uint64_t live_ub = std::min(s.ub - std::min(s.ub, s.frame_spawn), s.capacity);
        return std::max<uint32_t>(1, (uint32_t)((live_ub + s.frame_spawn + FW_TILE - 1) / FW_TILE));
    }
    // (new particles beyond the capacity are dropped by the kernels -- spawn_room -- and need no tiles: a burst of 30 000 queued
    // particles into a type with 4096 caller-given slots used to ask for 118 new-particle tiles, more than the tile scratch of the
    // context -- ensure_tile_arrays, sized from the capacities -- holds: the entries of tiles beyond it were written out of bounds
    // (round 6: found as `check 2` under GPU contention, tools/r06_burst_repro.py))
    const uint32_t spawn_eff = std::min(s.frame_spawn, s.capacity);
    return std::max<uint32_t>(1, seg_live_tiles(s) + (spawn_eff + vtile - 1) / vtile + 1);
}

// (size of the new-particle tiles of a frame, update_tile_table: the smallest -- most parallel -- of 1 or 2 rounds for
// which all ACTIVE tiles of the frame are resident at once (kResidentSlots workgroups: 4 per CU); a second, nearly
// empty round of workgroups would cost a full tile lifetime)

// tile scratch sized for every segment at full capacity
fw_status ensure_tile_arrays(fw_ctx *ctx) {
    size_t tiles = 0, nest_tiles = 0, nest_ops = 0;
    for (auto &s : ctx->segs)
        // (worst case: every slot a new particle -- in one-round tiles -- behind a full buffer of live ones that all die: the bound of
        // update_tile_table's cap_tiles, whatever the frame asks for)
        if (s.in_use && !s.ring()) tiles += (s.capacity + FW_TILE - 1) / FW_TILE + (s.capacity + FW_VTILE - 1) / FW_VTILE + 3;
    for (auto &sp : ctx->spawners) {
        if (!sp.alive) continue;
        for (auto &e : sp.em)
            if (e.es.mode == FW_MODE_NESTED) {
                nest_tiles += (ctx->segs[sp.seg[e.es.target_particle_type]].capacity + FW_NEST_TILE - 1) / FW_NEST_TILE + 1;
                nest_ops++;
            }
    }
    if (tiles > ctx->tiles_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        size_t ncap = tiles * 2;
        if (ctx->g.tile_cnt) hipFree(ctx->g.tile_cnt), hipFree(ctx->g.tile_off), hipFree(ctx->g.tile_status);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_cnt, ncap * sizeof(uint32_t)));
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_off, ncap * sizeof(uint32_t)));
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_status, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.tile_status, 0, ncap * sizeof(unsigned long long)));
        if (ctx->g.tile_box) hipFree(ctx->g.tile_box);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_box, ncap * 8 * sizeof(float)));
        FW_HIP(ctx, fw_memset_done(ctx->g.tile_box, 0, ncap * 8 * sizeof(float)));
        ctx->boxes_epoch = 0;
        if (ctx->g.dbg_ts) hipFree(ctx->g.dbg_ts);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.dbg_ts, (32768 + 8 * ncap) * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.dbg_ts, 0, (32768 + 8 * ncap) * sizeof(unsigned long long)));
        if (ctx->d_fce) hipFree(ctx->d_fce);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_fce, 2 * ncap * sizeof(uint4)));
        FW_HIP(ctx, fw_memset_done(ctx->d_fce, 0, 2 * ncap * sizeof(uint4)));
        if (ctx->d_fc) hipFree(ctx->d_fc);
        ctx->fc_len = ncap + (ncap / 64 + 2) * FW_FC_S2_STRIDE + 8;  // P | P2 | tag (64-bit words)
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_fc, 3 * ctx->fc_len * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->d_fc, 0, 3 * ctx->fc_len * sizeof(unsigned long long)));
        ctx->fc_ok = false, ctx->boxes_epoch = 0;
        ctx->fc_dirty = false;
        ctx->tiles_cap = ncap;
    }
    if (nest_tiles > ctx->nest_tiles_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        size_t ncap = nest_tiles * 2;
        if (ctx->g.nest_status) hipFree(ctx->g.nest_status);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.nest_status, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.nest_status, 0, ncap * sizeof(unsigned long long)));
        ctx->nest_tiles_cap = ncap;
    }
    if (nest_ops > ctx->nest_ops_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        size_t ncap = nest_ops * 2 + 16;
        if (ctx->g.nest_ticket) hipFree(ctx->g.nest_ticket);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.nest_ticket, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.nest_ticket, 0, ncap * sizeof(unsigned long long)));
        ctx->nest_ops_cap = ncap;
    }
    return FW_OK;
}

// device table, look-back words and per-frame pinned records of the range launch, sized for the worst case of every range
// segment (old + young workgroups cover at most the ring, new ones at most a ring of new particles) when a segment is built
// or reallocated: fw_step itself never allocates for them
fw_status ensure_range_arrays(fw_ctx *ctx) {
    size_t tiles = 0;
    for (auto &s : ctx->segs)
        if (s.in_use && s.range)  // OLD tiles of FW_TILE, YOUNG tiles of the build's size, NEW workgroups of FW_BLOCK
            // (... or, a launch on one-round tiles -- fw_ctx::range_small -- OLD and YOUNG tiles of FW_BLOCK)
            tiles += 2 * ((size_t)s.capacity / FW_BLOCK + 4) + (size_t)s.capacity / FW_BLOCK + 2;
    if (tiles > ctx->rdesc_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = tiles + tiles / 2 + 64;
        if (ctx->d_rdesc) hipFree(ctx->d_rdesc);
        if (ctx->h_rdesc) hipHostFree(ctx->h_rdesc);
        if (ctx->d_rstatus) hipFree(ctx->d_rstatus);
        ctx->d_rdesc = nullptr, ctx->h_rdesc = nullptr, ctx->d_rstatus = nullptr, ctx->rdesc_cap = 0;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_rdesc, ncap * sizeof(FwRangeDesc)));
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_rdesc, ncap * sizeof(FwRangeDesc), hipHostMallocDefault));
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_rstatus, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->d_rstatus, 0, ncap * sizeof(unsigned long long)));
        if (ctx->dbg & 8u) {  // per-workgroup timestamps of the last range launch (tools/range_timeline.py)
            if (ctx->d_rts) hipFree(ctx->d_rts);
            ctx->d_rts = nullptr;
            FW_HIP(ctx, hipMalloc((void **)&ctx->d_rts, ncap * 8 * sizeof(unsigned long long)));
            FW_HIP(ctx, fw_memset_done(ctx->d_rts, 0, ncap * 8 * sizeof(unsigned long long)));
        }
        ctx->rdesc_cap = ncap;
        ctx->r_force = true, ctx->rtab_pending = false;
    }
    // one record per segment slot + one op per emission entry of the context
    const size_t need = round_up((uint32_t)(ctx->max_seg * sizeof(FwRangeRec)), 64) + (size_t)(ctx->n_emits + 8) * sizeof(FwOp) + 64;
    if (ctx->n_range && need > ctx->rparam_bytes) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t nb = need * 2;
        for (int i = 0; i < kParamRing; i++) {
            if (ctx->h_rparam[i]) FW_HIP(ctx, ctx->param_bar ? hipFree(ctx->h_rparam[i]) : hipHostFree(ctx->h_rparam[i]));
            ctx->h_rparam[i] = nullptr;
            FW_HIP(ctx, ctx->param_bar ? hipExtMallocWithFlags((void **)&ctx->h_rparam[i], nb, hipDeviceMallocFinegrained)
                                       : hipHostMalloc((void **)&ctx->h_rparam[i], nb, hipHostMallocDefault));
            memset(ctx->h_rparam[i], 0, nb);
            ctx->rslot_frame[i] = 0;
        }
        ctx->rparam_bytes = nb;
    }
    return FW_OK;
}

// slot of particle 0 of a segment whose live count is `count` (exact): 0 unless the segment is a ring
// (a range ring: the old part sits right before the young part; its size is the device's -- rold_seen, refreshed together
// with the exact counts -- for a type that receives Nested children, and count - young_n, the same number, otherwise)
uint32_t ring_head_exact(const SegHost &S, uint32_t count) {
    if (S.fifo) return S.head;
    if (!S.range) return 0u;
    const uint32_t n_old = S.range_dev ? std::min(S.rold_seen, count) : (count > S.young_n ? count - S.young_n : 0u);
    return (uint32_t)(((uint64_t)S.young_lo + S.capacity - (n_old % S.capacity)) % S.capacity);
}

fw_status ensure_param_ring(fw_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->param_bytes) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    FW_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    size_t nb = std::max<size_t>(bytes * 2, 1 << 16);
    for (int i = 0; i < kParamRing; i++) {
        if (ctx->h_param[i]) hipHostFree(ctx->h_param[i]), hipFree(ctx->d_param[i]);
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_param[i], nb, hipHostMallocDefault));
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_param[i], nb));
        ctx->consumed_pending[i] = false;
    }
    ctx->param_bytes = nb;
    return FW_OK;
}

fw_status upload_seg(fw_ctx *ctx, uint32_t si) {
    const SegHost &s = ctx->segs[si];
    FwSeg d{};
    d.buf[0] = s.buf[0], d.buf[1] = s.buf[1];
    d.destroyed = s.destroyed;
    d.capacity = s.capacity;
    d.type_idx = s.type_idx;
    d.n_lplanes = s.n_lplanes;
    d.inst = s.inst, d.inst_cap = s.inst_cap;
    d.cpl = s.ring() ? 1u : 0u;  // (a ring's Q1 / Q3 regions: component planes, fw_device.h)
    d.lplane_emit[0] = d.lplane_emit[1] = 0xFFFFFFFFu;
    if (s.virt_parent && s.spawner >= 0)
        for (uint32_t k = 0; k < s.n_lplanes && k < 2u; k++) {
            const auto &em = ctx->spawners[s.spawner].em;
            const int32_t ei = s.lplane_emission[k];
            if (ei >= 0 && (size_t)ei < em.size() && em[ei].assigned) d.lplane_emit[k] = em[ei].emit_idx;
        }
    FW_HIP(ctx, hipMemcpy(ctx->d_segs.d + si, &d, sizeof d, hipMemcpyHostToDevice));
    return FW_OK;
}

fw_status alloc_seg_buffers(fw_ctx *ctx, SegHost &s, uint32_t capacity, bool want_destroyed) {
    const size_t bytes = FW_BUF_BYTES((size_t)capacity, s.n_lplanes + s.n_xplanes);
    char *b = nullptr;
    hipError_t e = hipMalloc((void **)&b, bytes * (s.ring() ? 1 : 2));  // a ring is updated in place: one buffer
    if (e != hipSuccess) return fail(ctx, FW_ENOMEM, std::string("hipMalloc particle buffers: ") + hipGetErrorString(e));
    s.buf[0] = b;
    s.buf[1] = s.ring() ? b : b + bytes;
    s.capacity = capacity;
    s.destroyed = nullptr;
    if (want_destroyed) {
        e = hipMalloc((void **)&s.destroyed, (size_t)capacity * sizeof(fw_particle));
        if (e != hipSuccess) return fail(ctx, FW_ENOMEM, "hipMalloc destroyed buffer");
    }
    FW_HIP(ctx, fw_launch_fill_colors(ctx->stream, s.buf[0], s.ring() ? nullptr : s.buf[1], capacity, s.fill_bc, s.fill_em));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));  // callers go on with blocking copies on the null stream
    return FW_OK;
}

// the size of the old part of every range ring that receives Nested children, as the device has it (the stream has been
// waited for): what ring_head_exact derives such a ring's first slot from
fw_status refresh_rold(fw_ctx *ctx) {
    bool any = false;
    for (const SegHost &S : ctx->segs) any |= S.in_use && S.range && S.range_dev;
    if (!any) return FW_OK;
    const uint32_t n = (uint32_t)ctx->segs.size();
    std::vector<uint32_t> r(n);
    FW_HIP(ctx, hipMemcpy(r.data(), ctx->g.rold + (size_t)ctx->parity * ctx->max_seg, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++)
        if (ctx->segs[i].in_use && ctx->segs[i].range) ctx->segs[i].rold_seen = r[i];
    return FW_OK;
}

// exact device counts -> host upper bounds (synchronises)
fw_status refresh_counts_exact(fw_ctx *ctx) {
    fw_status st = sync(ctx);
    if (st) return st;
    const uint32_t n = (uint32_t)ctx->segs.size();
    if (!n) return FW_OK;
    std::vector<uint32_t> c(n);
    FW_HIP(ctx, hipMemcpy(c.data(), ctx->g.count + (size_t)ctx->parity * ctx->max_seg, n * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++)
        if (ctx->segs[i].in_use) ctx->segs[i].ub = c[i];
    for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;
    return refresh_rold(ctx);
}

fw_status check_device_errors(fw_ctx *ctx) {
    // (the stream has been waited for: whatever a kernel reported is in the pinned word by now)
    const bool fresh = poll_device_error(ctx);
    // (every kernel that sets a flag says so in the pinned word next to err_host -- fw_flag: nothing there, nothing to fetch)
    if (!*(const volatile unsigned long long *)(ctx->h_err + 1) && !ctx->trace) return FW_OK;
    ctx->h_err[1] = 0ull;
    uint32_t ev[8] = {};
    FW_HIP(ctx, hipMemcpy(ev, ctx->g.err, sizeof ev, hipMemcpyDeviceToHost));
    const uint32_t e = ev[0];
    if (ctx->trace && ctx->d_tile_first) {
        uint32_t t[2] = {77, 77};
        hipMemcpy(t, ctx->d_tile_first, sizeof t, hipMemcpyDeviceToHost);
        fprintf(stderr, "[fw] check: flags=%u table=[%u,%u] ptr=%p fc=%p tiles_cap=%zu\n", e, t[0], t[1],
                (void *)ctx->d_tile_first, (void *)ctx->d_fc, ctx->tiles_cap);
    }
    if (!e) return FW_OK;
    uint32_t zero = 0;
    FW_HIP(ctx, hipMemcpy(ctx->g.err, &zero, sizeof zero, hipMemcpyHostToDevice));
    // (the two bits are looked at independently -- a word with both set used to lose the capacity report when the internal
    // error was not news: ADVICE r04 -- and the more severe status wins)
    fw_status cap = FW_OK;
    if (e & FW_ERR_CAPACITY)
        cap = fail(ctx, FW_ECAPACITY,
                   "a particle type overflowed its device capacity; particles were dropped "
                   "(raise fw_particle_settings.capacity) [device flags " + std::to_string(e) + ", segment " +
                       std::to_string(ev[1]) + ": " + std::to_string(ev[2]) + " particles (" + std::to_string(ev[4]) +
                       " resident), " + std::to_string(ev[3]) + " tiles launched]");
    if ((e & FW_ERR_FORECAST) && fresh) {
        // An internal check of an update kernel failed.  The spawner it names is marked (poll_device_error): its own calls
        // refuse from now on.  Whoever synchronises first is told once; later synchronisations -- of healthy spawners -- are
        // not failed again for an error that has been reported and contained.
        ctx->poison_msg += " [device flags " + std::to_string(e) + ", check " + std::to_string(ev[5]) + ": " + std::to_string(ev[6]) +
                           " " + std::to_string(ev[7]) + "]";
        return poisoned_status(ctx);
    }
    return cap;  // FW_ERR_LOOKBACK_TIMEOUT is informational: the fallback path produced the same result
}

// The pinned error word (FwGlobals::err_host): a kernel's internal check failed.  No synchronisation, no HIP call: fw_step and
// every reader look here first.  The spawner the segment belongs to (every spawner, when the error names none) is marked.
bool poll_device_error(fw_ctx *ctx) {
    const volatile unsigned long long *w = ctx->h_err;
    const unsigned long long v = w ? *w : 0ull;
    if (!v) return false;
    *ctx->h_err = 0ull;
    const uint32_t check = (uint32_t)(v >> 32) & 0x7FFFFFFFu, seg = (uint32_t)v;
    // (frames enqueued before the host looked here repeat the report: only a spawner that was healthy so far is news)
    bool one = false, news = false;
    if (seg < ctx->segs.size() && ctx->segs[seg].in_use && ctx->segs[seg].spawner >= 0 &&
        (size_t)ctx->segs[seg].spawner < ctx->spawners.size()) {
        SpawnerHost &sp = ctx->spawners[ctx->segs[seg].spawner];
        news = !sp.poisoned;
        sp.poisoned = true;
        one = true;
        ctx->n_poisoned++;
    }
    if (!one)
        for (auto &sp : ctx->spawners) news |= sp.alive && !sp.poisoned, sp.poisoned |= sp.alive, ctx->n_poisoned++;
    if (!news) return false;
    ctx->poison_msg = "internal error: check " + std::to_string(check) + " of an update kernel failed" +
                      (one ? " for segment " + std::to_string(seg) : std::string()) +
                      "; the particle state of the spawner is invalid -- rebuild it with fw_spawner_update_settings (drops its "
                      "particles) or destroy it";
    return true;
}
fw_status poisoned_status(fw_ctx *ctx) { return fail(ctx, FW_EHIP, ctx->poison_msg.empty() ? "spawner poisoned by an earlier internal error" : ctx->poison_msg); }
// A check of the HOST half of fw_step failed after the frame's bookkeeping was committed (clocks, RNG serials, cohorts, ring heads
// have advanced; a launch may be out): nothing can be rolled back, so the spawner the segment belongs to (every spawner, when no
// segment is named) is treated like one a kernel's check failed for -- sticky until rebuilt or destroyed
fw_status poison_segment(fw_ctx *ctx, uint32_t si, const std::string &what) {
    bool one = false;
    if (si < ctx->segs.size() && ctx->segs[si].in_use && ctx->segs[si].spawner >= 0 && (size_t)ctx->segs[si].spawner < ctx->spawners.size()) {
        SpawnerHost &sp = ctx->spawners[ctx->segs[si].spawner];
        if (!sp.poisoned) sp.poisoned = true, ctx->n_poisoned++;
        one = true;
    }
    if (!one)
        for (auto &sp : ctx->spawners)
            if (sp.alive && !sp.poisoned) sp.poisoned = true, ctx->n_poisoned++;
    ctx->poison_msg = "internal error: " + what + (one ? " (segment " + std::to_string(si) + ")" : std::string()) +
                      "; the particle state of the spawner is invalid -- rebuild it with fw_spawner_update_settings (drops its "
                      "particles) or destroy it";
    return poisoned_status(ctx);
}

SpawnerHost *get_spawner(fw_ctx *ctx, fw_spawner h) {
    if (!ctx || h < 0 || (size_t)h >= ctx->spawners.size() || !ctx->spawners[h].alive) {
        if (ctx) ctx->err = "invalid spawner handle";
        return nullptr;
    }
    return &ctx->spawners[h];
}

fw_status read_counts(fw_ctx *ctx, std::vector<uint32_t> &out) {
    fw_status st = sync(ctx);
    if (st) return st;
    out.assign(ctx->segs.size(), 0);
    if (!out.empty())
        FW_HIP(ctx, hipMemcpy(out.data(), ctx->g.count + (size_t)ctx->parity * ctx->max_seg,
                              out.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if ((st = refresh_rold(ctx))) return st;
    return check_device_errors(ctx);
}

// ParticleSpawnerData::active (core.rs:288-302) with exact device counts
bool spawner_active(const fw_ctx *ctx, const SpawnerHost &sp, const std::vector<uint32_t> &counts) {
    bool any = false;
    for (uint32_t si : sp.seg) any |= counts[si] != 0;
    bool enabled = false;
    for (const EmissionHost &e : sp.em) enabled |= e.emits_on_other_particles ? (e.enabled && any) : e.enabled;
    (void)ctx;
    return enabled;
}

}  // namespace fwh
