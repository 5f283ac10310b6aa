// fw_engine_step.cpp -- fw_step: one frame = spawn_particles + update_particles for every spawner (core.rs:367-670): lifetime windows, emission clocks, cohort replay, launch assembly
// (host engine of libfirework_hip.so: fw_engine.h lists its translation units; there is no CPU simulation path in this library)
#include "fw_engine.h"

extern "C" {

// ---- the frame ---------------------------------------------------------------------------
fw_status fw_step(fw_ctx *ctx, float dt) {
    if (!ctx) return FW_EINVAL;
    hipSetDevice(ctx->device);
    auto prof_t = std::chrono::steady_clock::now();
    auto prof = [&](int i) {
        if (!ctx->host_prof) return;
        const auto now = std::chrono::steady_clock::now();
        if (ctx->frame < ctx->host_prof_skip) {
            prof_t = now;
            return;
        }
        ctx->prof_ns[i] += std::chrono::duration<double, std::nano>(now - prof_t).count();
        prof_t = now;
    };
    // a spawner whose particle state an internal error invalidated (SpawnerHost::poisoned): no further frame is enqueued on
    // top of it -- for anybody: the frame is all-or-nothing -- until it has been rebuilt or destroyed
    poll_device_error(ctx);
    if (ctx->n_poisoned)  // (counted, not looked for: with thousands of spawners a scan of their records would be a cost of every frame)
        for (const SpawnerHost &sp : ctx->spawners)
            if (sp.alive && sp.poisoned) return poisoned_status(ctx);
    poll_snapshots(ctx);
    if (ctx->derive_ready_any) {  // types whose caller-written particles have all been through an update (see the end of fw_step)
        ctx->derive_ready_any = false;
        for (uint32_t i = 0; i < ctx->segs.size(); i++) {
            SegHost &S = ctx->segs[i];
            if (!S.in_use || !S.derive_ready) continue;
            S.derive_ready = false;
            if (S.colors_dirty || !wants_derived(ctx, S)) continue;  // (detached / rewritten since)
            const fw_status dst = set_derived(ctx, i, true);
            if (dst) return dst;
        }
    }

    // per-frame scratch lives in the context: with thousands of emitters the allocations were a visible part of the
    // host's ~60 ns per emitter
    auto &levels = ctx->levels;
    for (auto &L : levels) L.g.clear(), L.n.clear();
    ctx->fifo_ops.clear(), ctx->fifo_mat_ops.clear(), ctx->range_mat_ops.clear();
    ctx->seg_kind_changed = false;
    if (!std::isfinite(dt))  // 0 * inf = NaN: an angular velocity of zero does not stay zero (core.rs:648-650)
        for (uint32_t si = 0; si < ctx->segs.size(); si++)
            if (ctx->segs[si].in_use && ctx->segs[si].nospin) {
                fw_status nst = leave_nospin(ctx, si);
                if (nst) return nst;
            }
    if (ctx->n_fifo) {
        // the FIFO order rests on ages that never decrease: a negative or non-finite dt ends the mode (as does a dt so
        // small that the cohort list grows without bound)
        const bool dt_ok = dt >= 0.0f && std::isfinite(dt);
        for (uint32_t si = 0; si < ctx->segs.size(); si++) {
            SegHost &S = ctx->segs[si];
            // (a type that receives Nested children: a step as long as its lifetime would destroy children whose number only
            // this frame's Nested pass will know)
            if (!S.in_use || !S.fifo || (dt_ok && S.coh.size() < kMaxCohorts && !(S.fifo_dev && dt >= S.fifo_life))) continue;
            fw_status cst = fifo_to_general(ctx, si);
            if (cst) return cst;
        }
    }
    ctx->range_ops.clear();
    if (ctx->n_range) {
        // The in-place young part of a range ring rests on ages that never decrease and on nobody -- a particle spawned this
        // frame included -- dying before its age reaches lifetime.min: a negative or non-finite dt, a step as long as the
        // shortest lifetime, or a cohort list that grows without bound (a dt thousands of times smaller than the
        // lifetimes) end the mode; the type continues on the compacting path.
        const bool flood = ctx->birth_age.size() > (1u << 22);
        for (uint32_t si = 0; si < ctx->segs.size(); si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.range) continue;
            if (!flood && dt >= 0.0f && dt < S.range_life_lo && S.ycoh.size() < kMaxCohorts && S.dcoh.size() < kMaxCohorts) continue;
            fw_status cst = fifo_to_general(ctx, si);
            if (cst) return cst;
        }
    }
    if (!ctx->n_range) ctx->birth_age.clear();
    // Round 6: rings that receive Nested children need the size of a cohort -- which only the device knows -- in
    // the frame that cohort may start to die; the update of the cohort's own frame left it in a pinned report ring, a lifetime.min
    // ago.  A report that is STILL missing after the stream has been waited for used to be found in the middle of the frame's
    // bookkeeping (clocks advanced, earlier ring launches possibly out): nothing could be rolled back and the spawner was
    // poisoned.  It is looked for HERE, before anything of the frame is committed: such a ring continues on the compacting path
    // -- exact counts from the device, particles and order kept (fifo_to_general) -- and the frame goes ahead.  A failed check
    // of the host's bookkeeping costs a conversion, not the spawner's particles.  (FW_DEBUG 512 in the `ab` build pretends the
    // report of the first due cohort of frame 60 is missing: tests/test_gpu_range.py.)
    if (ctx->n_fifo || ctx->n_range) {
        auto report_missing = [&](const SegHost &S, uint64_t frame) -> bool {
            if (FW_DBG(ctx->dbg, 512u) && ctx->frame == 60u) return true;
            const uint32_t ep = (uint32_t)((frame + 1) & 0x3FFFFFFFu) ? (uint32_t)((frame + 1) & 0x3FFFFFFFu) : 1u;
            const volatile unsigned long long *row = S.h_report + (frame % kReportRing);
            for (int spin = 0; (uint32_t)(*row >> 32) != ep && spin < 100000; spin++) __builtin_ia32_pause();
            if ((uint32_t)(*row >> 32) == ep) return false;
            if (sync(ctx) != FW_OK) return true;
            return (uint32_t)(*row >> 32) != ep;
        };
        for (uint32_t si = 0; si < ctx->segs.size(); si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.h_report) continue;
            bool missing = false;
            if (S.fifo && S.fifo_dev) {
                for (const SegHost::Cohort &c : S.coh) {
                    if (!(c.age + dt >= S.fifo_life)) break;  // (the oldest die first: fw_step's own test, below)
                    if (!c.known && c.frame != ctx->frame && report_missing(S, c.frame)) missing = true;
                }
            } else if (S.range && S.range_dev) {
                for (const SegHost::DCohort &c : S.dcoh) {
                    float age = INFINITY;  // (age_before of the range block below)
                    if (!ctx->birth_age.empty() && c.frame >= ctx->birth_age.front().frame) {
                        const size_t i = (size_t)(c.frame - ctx->birth_age.front().frame);
                        age = i < ctx->birth_age.size() ? ctx->birth_age[i].age : 0.0f;
                    }
                    if (age + dt < S.range_life_lo) break;
                    if (!c.known && report_missing(S, c.frame)) missing = true;
                }
            }
            if (!missing) continue;
            ctx->recovered_rings++;
            fw_status cst = fifo_to_general(ctx, si);
            if (cst) return cst;
        }
    }
    // (frame_spawn is reset in the lifetime-window pass below: one pass over the segments instead of two)
    bool new_static = std::isfinite(dt);  // cleared by any Global op whose particles might not survive this step

    // lifetime windows: drop the spawns that must have expired by now and tighten the bounds with what is left.
    // (ages are fp32 sums of the same dt values on the device; the margin covers the rounding difference)
    if (!(dt >= 0.0f) || !std::isfinite(dt))
        for (auto &S : ctx->segs) S.win_ok = false;  // ages would not grow monotonically
    // (the same pass notes what the rest of the frame asks of every segment: which Nested-fed ones must grow, whether a
    // particle type collides, whether a compacting segment has an instance buffer attached)
    bool any_coll = false, any_inst_general = false;
    struct {
        uint64_t fifo_parts = 0, range_parts = 0;
        bool fifo_dev = false, fifo_coll = false, fifo_inst = false, range_dev = false, range_coll = false, range_inst = false;
    } ring_stats;
    ctx->grow_scratch.clear();
    // the lifetime window of a segment: drop the spawns that must have expired, tighten the bound with what is left
    // (an age is an fp32 running sum of the dt values: up to half an ulp of the age per step taken, i.e. a relative
    // error below steps * 6e-8; the horizon carries that (with a factor 4) on top of a fixed 1e-3)
    auto expire_window = [&](SegHost &S) {
        while (!S.win.empty() &&
               ctx->sim_time - S.win.front().t >=
                   S.life_bound * (1.0 + 1e-3 + 2.4e-7 * (double)(ctx->frame - S.win.front().frame)) + 1e-6) {
            S.win_sum -= S.win.front().n;
            S.win.pop_front();
        }
        if (S.win_sum < S.ub) S.ub = (uint32_t)S.win_sum;
    };
    // (solo segments -- small types with one feeder, thousands of them -- are not visited: the spawner loop below does this for
    // them where it makes their op; fw_ctx::big_list holds everybody else)
    const bool by_list = ctx->n_solo != 0;
    if (by_list && ctx->big_dirty) {
        ctx->big_list.clear();
        for (uint32_t si = 0; si < ctx->segs.size(); si++)
            if (ctx->segs[si].in_use && !ctx->segs[si].solo) ctx->big_list.push_back(si);
        ctx->big_dirty = false;
    }
    const bool solo_ok = ctx->host_fast && ctx->use_small;
    for (size_t k = 0, ns = by_list ? ctx->big_list.size() : ctx->segs.size(); k < ns; k++) {
        const size_t si = by_list ? ctx->big_list[k] : k;
        SegHost &S = ctx->segs[si];
        if (k + 8 < ns) {  // the oldest window entry of a later segment: a heap line of its own, fetched ahead of time
            const SegHost &N = ctx->segs[by_list ? ctx->big_list[k + 8] : k + 8];
            if (N.win.n) __builtin_prefetch(&N.win.v[N.win.head]);
        }
        S.frame_spawn = 0;
        if (!S.in_use) continue;
        S.dead_at_end = false;  // (set again below for the segments this frame updates as range rings)
        any_coll |= S.collides && !S.ring() && !S.small;  // (a colliding type in a ring / on the small kernel is updated by that kernel)
        // (what decides the tile size of the ring launches -- fifo_small / range_small below -- gathered while the record is hot)
        if (S.fifo) ring_stats.fifo_parts += S.ub, ring_stats.fifo_dev |= S.fifo_dev, ring_stats.fifo_coll |= S.collides, ring_stats.fifo_inst |= S.inst != nullptr;
        if (S.range) ring_stats.range_parts += S.ub, ring_stats.range_dev |= S.range_dev, ring_stats.range_coll |= S.collides, ring_stats.range_inst |= S.inst != nullptr;
        any_inst_general |= !S.ring() && S.inst != nullptr;
        if (S.nested_fed && nested_fed_wants_growth(S)) ctx->grow_scratch.push_back((uint32_t)si);
        if (S.win_ok) expire_window(S);
        // a small type with one feeder: solo from here on (this was its last visit by this pass)
        if (S.small && S.one_feeder && solo_ok) S.solo = true, ctx->n_solo++, ctx->big_dirty = true;
    }

    // Nested-fed segments whose count (or its growth) nears their derived capacity: nested_fed_wants_growth.  By half again,
    // not by doubling.  Caller-given capacities are left alone.
    for (uint32_t si : ctx->grow_scratch) {
        SegHost &S = ctx->segs[si];
        // (growing one type's children may already have grown a later entry of the list)
        if (!S.in_use || !nested_fed_wants_growth(S)) continue;
        const uint32_t want = (uint32_t)std::min<uint64_t>((uint64_t)S.capacity * 3 / 2 + (uint64_t)(64.0f * S.dev_rate), 0x70000000ull);
        S.dev_count = 0, S.dev_rate = 0.f, S.dev_epoch = 0;
        fw_status gst = grow_segment(ctx, si, want, false);
        if (!gst) gst = grow_nested_children(ctx, ctx->spawners[S.spawner], (uint32_t)S.type);
        if (gst) return gst;
    }

    // Everything the host half changes before the frame is known to be enqueueable goes through this log; `rollback`
    // restores it (the per-segment live-count bounds only ever get looser, which is harmless).
    ctx->undo_em.clear(), ctx->undo_sp.clear();
    auto rollback = [&](fw_status why) {
        for (const auto &u : ctx->undo_em) {
            EmissionHost &E = ctx->spawners[u.spawner].em[u.entry];
            E.last_emission = u.last_emission, E.time_passed_in_cycle = u.time_passed_in_cycle;
            E.enabled = u.enabled, E.serial = u.serial;
        }
        for (const auto &u : ctx->undo_sp) ctx->spawners[u.spawner].manual_queued_count = u.manual_queued_count;
        // the spawn totals and lifetime windows the ops of this frame were entered into (note_spawned below)
        auto forget = [&](const FwOp &op) {
            SegHost &S = ctx->segs[op.seg];
            S.cum_spawn -= op.n;
            if (S.solo) S.ub -= std::min(S.ub, op.n);  // (its frame_spawn stays 0: the loop over the segments below does nothing for it)
            if (S.fifo || !S.win_ok || S.win.empty() || S.win.back().t != ctx->sim_time) return;
            S.win_sum -= op.n;
            if ((S.win.back().n -= op.n) == 0) S.win.pop_back();
        };
        for (auto &L : ctx->levels)
            for (const FwOp &op : L.g) forget(op);
        for (const FwOp &op : ctx->fifo_ops) forget(op);
        for (const FwOp &op : ctx->range_ops) forget(op);
        for (const auto &io : ctx->fifo_mat_ops) forget(io.second);
        for (const auto &io : ctx->range_mat_ops) forget(io.second);
        for (auto &S : ctx->segs) S.ub -= std::min(S.ub, S.frame_spawn), S.frame_spawn = 0;
        return why;
    };
    // a Global op enters its segment's spawn total and lifetime window as it is made (the segment record is in the cache
    // then; `rollback` takes it out again)
    auto note_spawned = [&](SegHost &S, uint64_t n) {
        S.cum_spawn += n;
        if (S.fifo || !S.win_ok) return;
        if (!S.win.empty() && S.win.back().t == ctx->sim_time) {
            S.win.back().n += n;
        } else {
            if (S.win.size() >= 8192) {  // very long lifetimes: fold the two oldest entries into the newer
                const uint64_t m = S.win.front().n;
                S.win.pop_front();
                S.win.front().n += m;
            }
            S.win.push_back(SegHost::Spawned{ctx->sim_time, n, ctx->frame});
        }
        S.win_sum += n;
    };

    // A parameter slot of the ring (pinned host memory the kernels read in place, or stage from): free once the frame that used
    // it last has been consumed.
    auto acquire_slot = [&](int *out) -> fw_status {
        const int slot = (int)(ctx->ring_seq++ % kParamRing);
        if (ctx->consumed_pending[slot]) {
            FW_HIP(ctx, hipEventSynchronize(ctx->ev_consumed[slot]));
            ctx->consumed_pending[slot] = false;
        }
        if (ctx->slot_frame[slot]) {  // zero-copy use: wait until a launch AFTER that frame has started
            const volatile unsigned long long *tag = ctx->h_done;
            for (int spin = 0; *tag < ctx->slot_frame[slot] && spin < 200000; spin++) __builtin_ia32_pause();
            if (*tag < ctx->slot_frame[slot]) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->slot_frame[slot] = 0;
        }
        *out = slot;
        return FW_OK;
    };
    // Frames of a context with small types hand their Global ops to the kernels as a TABLE in a parameter slot ({first op, one past
    // the last, particles} per segment, then the ops): the spawner loop writes both IN PLACE (OpList::borrow) -- with thousands of
    // emitters the ops are 200-400 KB a frame, and writing them to a list, reading the list and writing the table was a tenth of the
    // host's frame.  The headers are kept while the ops arrive (segments ascending, each one's ops next to each other: what a
    // context whose spawners were built in order delivers); anything else -- an op routed later, a list that outgrows the slot, a
    // frame that turns out to need the separate spawn passes -- falls back to the sort + header pass further down, over the same
    // memory or the list's own.
    struct OpHdr {  // FwUpdateArgs::seg_op_first: per segment {first op, one past its last, particles they spawn in all, 0}
        uint32_t o0, o1, n, pad;
    };
    int pre_slot = -1;
    OpHdr *pre_hdr = nullptr;
    size_t pre_tracked = 0;     // ops of levels[0].g whose headers are up to date
    bool pre_hdr_ok = false;
    // (not a table small enough for host-written device memory -- fw_ctx::b_param: that one is copied there in one sequential pass)
    if (ctx->host_fast && ctx->n_small != 0 && ctx->update_mode == FW_MODE_FUSED && !levels.empty() &&
        !(ctx->param_bar && ctx->segs.size() * sizeof(OpHdr) + ((size_t)ctx->n_in_use + 16) * sizeof(FwOp) <= kBarParamBytes)) {
        const size_t n_seg0 = ctx->segs.size(), off_ops0 = n_seg0 * sizeof(OpHdr);
        const size_t want_ops = (size_t)ctx->n_in_use + 16;
        fw_status pst = FW_OK;
        if (ctx->param_bytes < off_ops0 + want_ops * sizeof(FwOp)) pst = ensure_param_ring(ctx, off_ops0 + (want_ops * 2 + 64) * sizeof(FwOp));
        if (!pst) pst = acquire_slot(&pre_slot);
        if (pst) return pst;  // (nothing of the frame has been entered anywhere yet)
        pre_hdr = (OpHdr *)ctx->h_param[pre_slot];
        memset(pre_hdr, 0, off_ops0);
        levels[0].g.borrow((FwOp *)(ctx->h_param[pre_slot] + off_ops0), (ctx->param_bytes - off_ops0) / sizeof(FwOp));
        pre_hdr_ok = true;
    }
    // (the ring may be re-allocated when a frame needs more than it holds: a list that lives in it moves out first)
    auto ensure_ring = [&](size_t bytes) -> fw_status {
        if (bytes > ctx->param_bytes && pre_slot >= 0) {
            levels[0].g.reserve_own(levels[0].g.size());
            pre_slot = -1, pre_hdr = nullptr, pre_hdr_ok = false;
        }
        return ensure_param_ring(ctx, bytes);
    };
    // a refresh of the exact counts drops this frame's appends from every bound: back in they go
    auto readd_frame_spawns = [&]() {
        for (auto &X : ctx->segs) X.ub += X.frame_spawn;
        if (ctx->n_solo)  // (a solo segment's appends of the frame are in its op)
            for (auto &L : ctx->levels)
                for (const FwOp &op : L.g)
                    if (ctx->segs[op.seg].solo) ctx->segs[op.seg].ub += op.n;
    };

    prof(0);
    // spawn_particles, host half (core.rs:377-428): emission clocks and Global counts
    for (size_t h = 0; h < ctx->spawners.size(); h++) {
        SpawnerHost &sp = ctx->spawners[h];
        if (h + 8 < ctx->spawners.size()) {  // the entries of a later spawner live on the heap: fetch them ahead of time
            const SpawnerHost &nx = ctx->spawners[h + 8];
            if (nx.em.size() > 1) {  // (a single entry is part of the spawner's record: the array is streamed as it is)
                __builtin_prefetch(nx.em.data());
                __builtin_prefetch((const char *)nx.em.data() + 128);
            }
        }
        // (solo segments -- visited here and nowhere else in the frame -- fetched a few spawners ahead of time, record first, then the
        // two ends of the lifetime window: built and measured, 38.5 -> 42 us per frame at 2048 x 200, the loop is bound by its
        // instructions and stores by now, not by the latency of its loads: profiles/r05/host_fast_prefetch.txt)
        if (!sp.alive) continue;
        // `if data.active()` (core.rs:378): an entry that emits on other particles contributes only when
        // some particle exists; with no particle at all the Nested arm below is a no-op anyway, so the
        // host-side gate "any entry enabled" gives the same state transitions without a device round trip.
        bool any_enabled = false;
        for (const EmissionHost &e : sp.em) any_enabled |= e.enabled;
        if (!any_enabled) continue;
        for (size_t i = 0; i < sp.em.size(); i++) {
            EmissionHost &E = sp.em[i];
            if (!E.enabled) continue;
            const fw_emission_settings &es = E.es;
            const uint32_t dst = E.dst_seg;
            if (es.mode == FW_MODE_GLOBAL) {
                uint64_t n = 0;
                ctx->undo_em.push_back(fw_ctx::EmUndo{(uint32_t)h, (uint32_t)i, E.last_emission, E.time_passed_in_cycle,
                                                      E.enabled, E.serial});
                if (es.pacing_kind == FW_PACING_ONESHOT) {
                    E.enabled = false;  // core.rs:397-400
                    n = es.oneshot_count;
                } else if (es.pacing_kind == FW_PACING_ONDEMAND) {
                    n = sp.manual_queued_count;  // core.rs:401-405
                    if (n) ctx->undo_sp.push_back(fw_ctx::SpUndo{(uint32_t)h, n});
                    sp.manual_queued_count = 0;
                } else {
                    E.time_passed_in_cycle = fw_rem_euclid(E.time_passed_in_cycle + dt, es.duration);  // core.rs:412-414
                    float next = 0.f;
                    n = es.duration == 1.0f
                            ? fw_emission_count_unit(E.time_passed_in_cycle, E.last_emission, es.offset_start, es.offset_end, E.between, &next)
                            : fw_emission_count(E.time_passed_in_cycle, E.last_emission, es.duration, es.offset_start, es.offset_end,
                                                es.count, &next);
                    E.last_emission = next;
                }
                if (!n) continue;
                if (n > kMaxSpawnPerOp)
                    return rollback(fail(ctx, FW_ECAPACITY, "emission count exceeds 2^30 particles in one frame"));
                SegHost &S = ctx->segs[dst];
                if (S.solo && S.win_ok) expire_window(S);  // its frame begins here (fw_ctx::n_solo)
                if (S.nested_fed && S.auto_capacity && S.capacity < 0x70000000u) {
                    // A type that receives Nested children too: its children are counted on the device, but its Global
                    // particles are counted right here.  Count of the latest snapshot row + every Global particle since
                    // (at most those the lifetime window still holds) + this op: past half the capacity, the segment grows
                    // NOW, however fast the burst -- the snapshot rule above only follows what the device has seen.
                    uint64_t since = S.cum_spawn - S.snap_cum;  // (includes this frame's earlier ops: note_spawned)
                    if (S.win_ok) since = std::min<uint64_t>(since, S.win_sum);
                    const uint64_t est = (uint64_t)S.snap_count + since + n;
                    if (est > S.capacity / 2) {
                        fw_status st = grow_segment(ctx, dst, (uint32_t)std::min<uint64_t>(est * 2, 0x70000000ull));
                        if (!st) st = grow_nested_children(ctx, sp, (uint32_t)es.particle_index);
                        // (the growth refreshed every bound from the device's exact counts, without this frame's appends)
                        readd_frame_spawns();
                        if (st) return rollback(st);
                        S.snap_count = S.ub - std::min(S.ub, S.frame_spawn), S.snap_cum = S.cum_spawn - S.frame_spawn;
                        S.dev_count = 0;
                    }
                }
                if (!S.nested_fed && (uint64_t)S.ub + n > S.capacity) {
                    fw_status st = refresh_counts_exact(ctx);
                    if (st) return rollback(st);
                    // the refresh dropped this frame's earlier appends from ub: add them back
                    readd_frame_spawns();
                    if ((uint64_t)S.ub + n > S.capacity) {
                        if ((uint64_t)S.ub + n > 0xF0000000ull)
                            return rollback(fail(ctx, FW_ECAPACITY, "particle type too large"));
                        // grow_segment copies `ub - frame_spawn` settled particles; appended ones are not on the
                        // device yet (spawn kernels of this frame have not been enqueued)
                        const uint32_t fs = S.frame_spawn;
                        S.ub -= fs;
                        const uint32_t settled = S.ub;
                        st = grow_segment(ctx, dst, (uint32_t)(settled + fs + n));
                        if (!st) st = grow_nested_children(ctx, sp, (uint32_t)es.particle_index);
                        readd_frame_spawns();
                        if (st) return rollback(st);
                    }
                }
                // does every particle of this op outlive the step?  (TypeHost::life_lo_safe; false for NaN)
                if (!S.ring() && !(dt < E.life_lo_safe)) new_static = false;
                auto fill = [&](FwOp &op) {
                    memset(&op, 0, sizeof op);
                    op.seg = dst, op.emit = E.emit_idx, op.n = (uint32_t)n;
                    op.rel_base = S.frame_spawn;
                    op.serial_base = E.serial;
                    memcpy(op.origin_pos, sp.origin_pos, sizeof sp.origin_pos);
                    memcpy(op.origin_rot, sp.origin_rot, sizeof sp.origin_rot);
                    memcpy(op.parent_vel, sp.parent_vel, sizeof sp.parent_vel);
                    op.speed = sp.mod_speed, op.scale = sp.mod_scale;
                };
                if (!S.ring()) {
                    // a compacting or small segment: the op goes to its emission level, written where it will stay
                    OpList &G = levels[i].g;
                    if (i == 0 && pre_hdr_ok) {  // ... and, in a list that lives in a parameter slot, into the segment's header
                        const uint32_t idx = (uint32_t)G.size();
                        OpHdr &H = pre_hdr[dst];
                        // (the kernels read a segment's ops through its header only: the ops of a segment must sit next to each other,
                        // in emission order -- segments need not ascend, which they do not once slots have been re-used)
                        if (H.o1 == 0) H.o0 = idx, H.o1 = idx + 1u, H.n = (uint32_t)n;
                        else if (H.o1 == idx) H.o1 = idx + 1u, H.n = (uint32_t)std::min<uint64_t>((uint64_t)H.n + n, 0xFFFFFFFFull);
                        else pre_hdr_ok = false;  // (ops of one segment apart: the sort + header pass below)
                        pre_tracked = idx + 1u;
                    }
                    fill(G.push_slot());
                } else {
                    FwOp op;
                    fill(op);
                    if (S.fifo && S.fifo_mat && !S.virt_parent)
                        ctx->fifo_mat_ops.push_back({(uint32_t)i, op});  // routed below, once the frame's Nested ops are known
                    else if (S.fifo)
                        ctx->fifo_ops.push_back(op);  // spawned inside fw_k_update_fifo, whatever else the frame holds
                    else if (S.range_mat && !S.virt_parent)
                        ctx->range_mat_ops.push_back({(uint32_t)i, op});  // routed below, once the frame's Nested ops are known
                    else
                        ctx->range_ops.push_back(op);  // spawned inside fw_k_update_range
                }
                E.serial += n;
                if (!S.solo) S.frame_spawn += (uint32_t)n;  // (a solo segment's one op of the frame carries the count)
                S.ub = (uint32_t)std::min<uint64_t>((uint64_t)S.ub + n, 0xFFFFFFFFull);
                if (S.small && S.ub > ctx->small_max && !S.wide) {
                    // no longer a few hundred particles: a WIDE type (a workgroup of the same kernel) from this frame on -- or, in a
                    // context that does not run wide types (fw_ctx::wide_mid_on), a compacting segment that stays eligible
                    S.wide = true, S.wide_big = false, ctx->small_dirty = true;
                    if (!ctx->wide_mid_on) {
                        const bool was_solo = S.solo;
                        small_suspend(ctx, S);
                        if (was_solo) S.frame_spawn = (uint32_t)n;
                    }
                }
                if (S.small && S.wide && S.ub > std::max(ctx->small_max, 2u * ctx->wide_max)) {
                    // ... no longer a few thousand: a compacting segment
                    const bool was_solo = S.solo;
                    leave_small(ctx, S);
                    if (was_solo) S.frame_spawn = (uint32_t)n;
                }
                note_spawned(S, n);
            } else {
                if (es.pacing_kind != FW_PACING_COUNT_OVER_DURATION) continue;  // warn_once + continue core.rs:474-485
                const SegHost &P = ctx->segs[sp.seg[es.target_particle_type]];
                // children are born with age 0 like any new particle: do they all outlive this step?
                if (!(dt < E.life_lo_safe)) new_static = false;
                FwNestOp op{};
                op.parent_seg = sp.seg[es.target_particle_type], op.child_seg = dst;
                op.emit = E.emit_idx, op.emit_slot = E.emit_slot;
                {  // parent tiles of the launch: from the host's bound of the parent count (exact counts live on the device)
                    const uint32_t par_ub = P.nested_fed ? P.capacity : std::min(P.ub, P.capacity);
                    op.n_tiles = (par_ub + FW_NEST_TILE - 1) / FW_NEST_TILE + 1;
                }
                op.speed = sp.mod_speed, op.scale = sp.mod_scale;
                // (parent_buf / parent_cap are filled in when the launch is built: a later entry of this frame may
                // still grow the parent segment)
                for (uint32_t k = 0; k < P.n_lplanes; k++)
                    if (P.lplane_emission[k] == (int32_t)i) op.parent_lplane = k;
                op.n_count = es.count, op.n_start = es.offset_start, op.n_end = es.offset_end;
                levels[i].n.push_back(op);
            }
        }
    }

    // Rings of spawners with Nested entries: a frame that runs a Nested pass (anywhere in the context: the launches are
    // per emission level) materialises their new particles with fw_k_spawn, at their level, so that the per-parent pass
    // finds them in memory (core.rs:488); any other frame spawns them inside fw_k_update_fifo like every other ring's.
    bool nested_frame = false;
    {
        for (auto &L : levels) nested_frame |= !L.n.empty();
        for (auto &io : ctx->fifo_mat_ops) {
            if (nested_frame) {
                io.second.head = ctx->segs[io.second.seg].head;
                levels[io.first].g.push_back(io.second);
            } else {
                ctx->fifo_ops.push_back(io.second);
            }
        }
        // (range rings other particles' entries emit from: the same choice; materialised at the tail of the ring -- the slot of
        // the list's first particle follows from the old part's size, which the device keeps: FwOp::range_ring)
        for (auto &io : ctx->range_mat_ops) {
            if (nested_frame) {
                io.second.head = ctx->segs[io.second.seg].young_lo, io.second.range_ring = 1u;
                levels[io.first].g.push_back(io.second);
            } else {
                ctx->range_ops.push_back(io.second);
            }
        }
        // every routed op now sits in exactly one list `rollback` walks (levels[].g, fifo_ops or range_ops): forgetting this one
        // too would take its particles out of cum_spawn twice
        ctx->fifo_mat_ops.clear(), ctx->range_mat_ops.clear();
    }
    // A ring that had to grow past its mode's slot limit inside the loop above (realloc_segment) continues as a compacting
    // segment from this very frame: the ops already queued for it as a ring's go where a compacting segment's ops go -- its
    // emission level -- or they would never be spawned while cum_spawn, ub and the lifetime window count them.
    if (ctx->seg_kind_changed) {
        auto reroute = [&](std::vector<FwOp> &list, bool was_fifo) {
            size_t w = 0;
            for (size_t r = 0; r < list.size(); r++) {
                const FwOp &op = list[r];
                const SegHost &S = ctx->segs[op.seg];
                if (was_fifo ? S.fifo : S.range) {
                    list[w++] = op;
                    continue;
                }
                size_t lvl = 0;
                const SpawnerHost &osp = ctx->spawners[S.spawner];
                for (size_t i = 0; i < osp.em.size(); i++)
                    if (osp.em[i].emit_idx == op.emit) lvl = i;
                levels[lvl].g.push_back(op);
                if (!(dt < osp.em[lvl].life_lo_safe)) new_static = false;  // (the test a compacting segment's op gets above)
            }
            list.resize(w);
        };
        reroute(ctx->fifo_ops, true);
        reroute(ctx->range_ops, false);
        for (auto &L : levels)
            for (FwOp &op : L.g)
                if (op.range_ring && !ctx->segs[op.seg].range) op.range_ring = 0u, op.head = 0u;
    }
    // ---- Nested entries run INSIDE the FIFO launch (fw_kernels.h: FwFifoNest; core.rs:471-546).  Every Nested entry of the frame
    // must qualify -- parents in a FIFO ring the host knows the count of (Global-fed), spawned inside the update kernel
    // (virt_parent), no other entry emitting from them; children received by a FIFO ring nothing else feeds and nothing emits from;
    // both in the one FIFO launch of the context, parents first; no instance buffers, no colliders -- and no ring may wait for
    // fw_k_spawn (the ops left at the levels then belong to compacting segments, which no Nested entry of the frame touches: the
    // general launch spawns them itself).  Otherwise the frame runs the separate passes (fw_k_spawn / fw_k_nest), as before.
    FwNestOp fuse_plan[FW_FIFO_NEST_MAX];
    uint32_t n_fuse = 0;
    bool fuse = false;
    if (nested_frame && ctx->nest_fuse && !any_coll && !ctx->seg_kind_changed && ctx->update_mode == FW_MODE_FUSED && ctx->n_fifo != 0 &&
        ctx->n_fifo <= FW_FIFO_PER_LAUNCH && ctx->fifo_ops.size() <= FW_INLINE_OPS && !ring_stats.fifo_coll && !ring_stats.fifo_inst &&
        dt >= 0.0f && std::isfinite(dt)) {
        fuse = true;
        for (auto &L : levels) {
            for (const FwOp &op : L.g) fuse &= !ctx->segs[op.seg].ring();
            for (const FwNestOp &op : L.n) {
                if (!fuse || n_fuse == FW_FIFO_NEST_MAX) {
                    fuse = false;
                    break;
                }
                const SegHost &P = ctx->segs[op.parent_seg], &Cs = ctx->segs[op.child_seg];
                const uint32_t p_in = P.ub - std::min(P.ub, P.frame_spawn);  // (a Global-fed FIFO ring: `ub` is exact)
                fuse = P.fifo && !P.fifo_dev && P.virt_parent && P.n_lplanes == 1 && Cs.fifo && Cs.fifo_dev && Cs.n_lplanes == 0 &&
                       op.parent_seg < op.child_seg && Cs.capacity <= FW_RANGE_MAX_CAPACITY &&
                       // the ring must not wrap into its head tile: the tiles' ranks are then the list order
                       (uint64_t)(P.head % FW_TILE) + p_in <= P.capacity;
                for (uint32_t k = 0; k < n_fuse; k++)
                    fuse &= fuse_plan[k].parent_seg != op.parent_seg && fuse_plan[k].parent_seg != op.child_seg &&
                            fuse_plan[k].child_seg != op.parent_seg && fuse_plan[k].child_seg != op.child_seg;
                if (fuse) fuse_plan[n_fuse++] = op;
            }
            if (!fuse) break;
        }
        fuse = fuse && n_fuse != 0;
    }
    if (nested_frame) (fuse ? ctx->fused_nest_frames : ctx->nest_pass_frames)++;
    // ---- segment -> tile table (device resident, re-uploaded only when a bound moves out of its band)
    prof(1);
    const uint32_t n_seg = (uint32_t)ctx->segs.size();
    fw_status st = update_tile_table(ctx);
    prof(2);
    if (st) return rollback(st);
    const uint32_t total_tiles = ctx->total_tiles_dev;
    // (every per-tile array of the context -- status words, forecast entries, tile boxes -- holds tiles_cap entries: a table that
    // asks for more is a bookkeeping error, and nothing of the frame has been enqueued yet)
    if (total_tiles > ctx->tiles_cap) return rollback(fail(ctx, FW_EHIP, "internal error: the tile table exceeds the tile scratch (frame not enqueued)"));

    size_t n_g = 0, n_n = 0;
    for (auto &L : levels) n_g += L.g.size(), n_n += L.n.size();
    // last thing that can fail before the frame is enqueued: room for the op tables of either form
    if ((st = ensure_ring((size_t)n_seg * 16 + n_g * sizeof(FwOp) + n_n * sizeof(FwNestOp) + 16))) return rollback(st);
    // ---- the frame will run
    prof(3);
    const uint32_t p = ctx->parity;
    // Particle types with collision settings (core.rs:607-624) run the count / scan / update-with-collisions launches
    // (everything materialised first, like FW_UPDATE_MODE=split); the streaming kernels never see a collider.
    if (ctx->seg_kind_changed)  // (a colliding ring that left its mode inside the spawner loop is a compacting segment from this frame on)
        for (const SegHost &S : ctx->segs) any_coll |= S.in_use && S.collides && !S.ring() && !S.small;
    const int frame_mode = any_coll ? FW_MODE_SPLIT_COLL : ctx->update_mode;
    const bool legacy = (n_n != 0 && !fuse) || frame_mode != FW_MODE_FUSED;  // (fuse: the Nested entries run inside the FIFO launch)

    FwUpdateArgs a{};
    a.seg_tile_first = ctx->d_tile_first;
    a.tile_desc = ctx->d_tile_desc;
    a.n_seg = n_seg;
    a.total_tiles = total_tiles;
    a.parity = p;
    a.epoch = (uint32_t)((ctx->frame + 1) & 0x3FFFFFFFu);
    if (!a.epoch) a.epoch = 1;
    a.dt = dt;
    a.spin_limit = ctx->spin_limit;
    a.dbg = ctx->dbg;
    a.vt_rounds = ctx->vt_rounds;
    a.resident_slots = (uint32_t)kResidentSlots;
    a.seg0_type = n_seg ? ctx->segs[0].type_idx | (ctx->segs[0].nospin ? FW_TYPE_IDX_NOSPIN : 0u) : 0;
    a.seg0_keys_off = n_seg ? ctx->segs[0].keys_off : 0;
    a.seg0_keys_len = n_seg ? ctx->segs[0].keys_len : 0;
    a.tile_keys = ctx->d_tile_keys;
    if (n_seg == 1 && ctx->segs[0].in_use && !ctx->segs[0].ring()) {
        const SegHost &S0 = ctx->segs[0];
        a.seg0_ib = S0.buf[p], a.seg0_ob = S0.buf[p ^ 1u];
        a.seg0_destroyed = S0.destroyed, a.seg0_inst = S0.inst;
        a.seg0_capacity = S0.capacity, a.seg0_n_lplanes = S0.n_lplanes, a.seg0_inst_cap = S0.inst_cap;
    }
    if (ctx->live_ring) {
        a.live_out = ctx->live_ring + (ctx->live_ring_frames % ctx->live_ring_n);
        a.live_next = ctx->live_ring + ((ctx->live_ring_frames + 1) % ctx->live_ring_n);
        ctx->live_ring_frames++;
    }
    a.new_static = (new_static && ctx->use_static_new) ? 1u : 0u;
    a.boxes = (ctx->track_aabb && frame_mode == FW_MODE_FUSED) ? 1u : 0u;
    a.force_colors = ctx->colors_dirty ? 1u : 0u;
    // (a ring that had to grow past the ring limit during this frame's spawner loop continues as a compacting segment:
    // look again -- never in a steady-state frame)
    if (ctx->seg_kind_changed)
        for (const SegHost &S : ctx->segs) any_inst_general |= S.in_use && !S.ring() && S.inst != nullptr;
    a.any_inst = any_inst_general ? 1u : 0u;
    a.use_stream = ctx->use_stream ? 1u : 0u;
    uint32_t dt_bits;
    memcpy(&dt_bits, &dt, 4);
    // Frames with Nested entries materialise their new particles before the update; the streaming kernel takes them as
    // loaded new-particle tiles with static slots, which needs every one of them to survive the step (new_static).
    const bool split = frame_mode != FW_MODE_FUSED;
    const bool fc_frame = !split && ctx->use_forecast && ctx->d_fc != nullptr;
    if (fc_frame) {
        for (uint32_t i = 0; i < n_seg; i++) a.fc_sums |= ctx->tiles_dev[i] > FW_FC_DIRECT ? 1u : 0u;
        const bool usable = ctx->fc_ok && ctx->fc_dt_bits == dt_bits && ctx->fc_tab_seq == ctx->tab_seq && a.epoch != 1u &&
                            ctx->fc_sums_prev == a.fc_sums &&  // the previous frame left the other format otherwise
                            (!legacy || (a.use_stream && a.new_static));
        if (ctx->fc_dirty) {
            // the tile indexing changed: sums left at indices of the old table must not leak into the new one
            FW_HIP(ctx, hipMemsetAsync(ctx->d_fc, 0, 3 * ctx->fc_len * sizeof(unsigned long long), ctx->stream));
            ctx->fc_dirty = false;
        }
        a.fc_s2 = (uint32_t)ctx->tiles_cap;
        a.fc_tag = (uint32_t)(ctx->fc_len - 1);
        a.fc_out = ctx->d_fc + (size_t)(ctx->fc_seq % 3u) * ctx->fc_len;
        a.fc_zero = ctx->d_fc + (size_t)((ctx->fc_seq + 1u) % 3u) * ctx->fc_len;
        a.fce_out = ctx->d_fce + (size_t)(ctx->fc_seq & 1u) * ctx->tiles_cap;
        a.fce_in = ctx->d_fce + (size_t)((ctx->fc_seq + 1u) & 1u) * ctx->tiles_cap;
        if (usable) a.fc_in = ctx->d_fc + (size_t)((ctx->fc_seq + 2u) % 3u) * ctx->fc_len;
        // ---- threshold forecast (fw_ctx::use_tf): a dt that differs from the previous frame's, below the theta its lists were made for
        const bool dt_same = ctx->fc_dt_bits == dt_bits;
        const bool tf_size = ctx->use_tf && !a.fc_sums && total_tiles >= ctx->tf_min_tiles && a.use_stream && dt > 0.0f && std::isfinite(dt);
        if (!usable && tf_size && ctx->tf_prev_theta > 0.0f && dt < ctx->tf_prev_theta && !dt_same && ctx->fc_ok &&
            ctx->fc_tab_seq == ctx->tab_seq && a.epoch != 1u && ctx->fc_sums_prev == a.fc_sums && (!legacy || a.new_static) &&
            ctx->tf_cap == ctx->tiles_cap) {
            FwResolveArgs ra{};
            ra.fce = ctx->d_fce + (size_t)((ctx->fc_seq + 1u) & 1u) * ctx->tiles_cap;
            ra.fct = ctx->d_fct + (size_t)((ctx->fc_seq + 1u) & 1u) * ctx->tiles_cap;
            ra.fcl = ctx->d_fcl + (size_t)((ctx->fc_seq + 1u) & 1u) * ctx->tiles_cap * FW_TF_K;
            ra.tile_desc = n_seg == 1 ? nullptr : ctx->d_tile_desc;
            ra.total_tiles = total_tiles, ra.parity = p, ra.epoch = a.epoch, ra.dt = dt;
            FW_HIP(ctx, fw_launch_fc_resolve(ctx->stream, ctx->g, ra));
            a.fc_in = ctx->d_fc + (size_t)((ctx->fc_seq + 2u) % 3u) * ctx->fc_len;  // (per-tile entries: only its non-nullness matters)
            ctx->tf_frames++;
        }
        // producers: lists for the NEXT frame while dt has been seen to vary (64 frames after the last change)
        if (!dt_same && ctx->fc_ok) ctx->tf_armed = 64u;
        ctx->tf_prev_theta = 0.0f;
        if (tf_size && ctx->tf_armed) {
            ctx->tf_armed--;
            if (ctx->tf_cap != ctx->tiles_cap) {  // (first use, or the tile scratch grew: ensure_tile_arrays)
                if (ctx->d_fct) (void)hipFree(ctx->d_fct), ctx->d_fct = nullptr;
                if (ctx->d_fcl) (void)hipFree(ctx->d_fcl), ctx->d_fcl = nullptr;
                ctx->tf_cap = 0;
                if (hipMalloc((void **)&ctx->d_fct, 2 * ctx->tiles_cap * sizeof(uint4)) == hipSuccess &&
                    hipMalloc((void **)&ctx->d_fcl, 2 * ctx->tiles_cap * FW_TF_K * sizeof(float2)) == hipSuccess)
                    ctx->tf_cap = ctx->tiles_cap;
                else
                    (void)hipGetLastError();  // (no memory for the lists: the look-back schedule stays)
            }
            if (ctx->tf_cap == ctx->tiles_cap) {
                a.fc_theta = dt * 1.25f;
                a.fct_out = ctx->d_fct + (size_t)(ctx->fc_seq & 1u) * ctx->tiles_cap;
                a.fcl_out = ctx->d_fcl + (size_t)(ctx->fc_seq & 1u) * ctx->tiles_cap * FW_TF_K;
                ctx->tf_prev_theta = a.fc_theta;
            }
        }
        ctx->fc_seq++;
        ctx->fc_sums_prev = a.fc_sums;
        if (ctx->trace)
            fprintf(stderr, "[fw] frame %llu epoch %u fc: usable=%d ok=%d dt_same=%d tab %llu/%llu sums %u legacy=%d new_static=%u seq %llu tiles %u\n",
                    (unsigned long long)ctx->frame, a.epoch, (int)usable, (int)ctx->fc_ok, (int)(ctx->fc_dt_bits == dt_bits),
                    (unsigned long long)ctx->fc_tab_seq, (unsigned long long)ctx->tab_seq, a.fc_sums, (int)legacy, a.new_static,
                    (unsigned long long)ctx->fc_seq, total_tiles);
    }
    // a snapshot row stays armed until its stores have been seen (a free-running host can be hundreds of frames
    // ahead of the device; re-arming by frame number would never catch one)
    int snap = -1;
    if ((ctx->frame % ctx->snap_every) == 0)
        for (int k = 0; k < kSnapRing && snap < 0; k++)
            if (!ctx->snap_pending[k]) snap = k;
    // (small types are bounded by their lifetime windows and have no grid to size: a context of nothing else takes no snapshots --
    // a row is a pass over every segment record on the host and a store over the bus per segment on the device)
    if (ctx->n_in_use == ctx->n_small) snap = -1;
    const bool take_snap = snap >= 0;
    a.host_counts = take_snap ? ctx->h_snap + (size_t)snap * ctx->max_seg : nullptr;

    FwInlineOps inl;
    int spawn_form = FW_SPAWN_NONE;
    int slot = -1;
    // the op table of the small launch (fw_k_update_small): the frame's table in Global-only frames, one of its own in frames that
    // run the separate passes
    const uint4 *small_hdr = nullptr;
    const FwOp *small_ops = nullptr;
    int small_tslot = -1;  // ... the parameter slot it lives in, in the latter case
    prof(4);

    if (legacy) {
        // Frames with Nested entries: parents spawned earlier in the frame must exist in memory before the
        // per-parent pass reads them (core.rs:488), so Global ops are materialised by fw_k_spawn, level by level.
        // Not those of SMALL types (round 5): no Nested entry reads them, their kernel spawns them itself (virtual particles) from a
        // table as in any other frame -- one spawner with a Nested entry among hundreds of small emitters used to send every op of
        // the context through fw_k_spawn and its staged table: 512 small emitters 16.7 us per frame, with examples/textures.rs next
        // to them 56.3 (profiles/r05/nested_among_small.txt).
        const bool small_virtual = ctx->n_small != 0 && ctx->ops_zerocopy && ctx->host_fast;
        auto stays_virtual = [&](const FwOp &op) { return small_virtual && ctx->segs[op.seg].small; };
        size_t n_virtual = 0;
        if (small_virtual)
            for (auto &L : levels)
                for (const FwOp &op : L.g) n_virtual += stays_virtual(op) ? 1u : 0u;
        n_g -= n_virtual;
        const size_t off_nops = n_g * sizeof(FwOp);
        const size_t bytes = off_nops + n_n * sizeof(FwNestOp) + 16;
        if ((st = ensure_ring(bytes))) return st;
        if ((st = acquire_slot(&slot))) return st;
        char *hp = ctx->h_param[slot];
        char *dp = ctx->d_param[slot];  // staged copy here: the few workgroups of the small spawn / nest kernels would
                                        // wait for the bus on their critical path (measured: no gain from reading in place)
        struct Launch {
            bool nested;
            size_t first, count;
            uint32_t blocks;
        };
        std::vector<Launch> launches;
        FwOp *h_ops = (FwOp *)hp;
        FwNestOp *h_nops = (FwNestOp *)(hp + off_nops);
        size_t gi = 0, ni = 0, pend_first = 0;
        uint32_t pend_blocks = 0;
        auto flush_global = [&]() {
            if (gi > pend_first) launches.push_back(Launch{false, pend_first, gi - pend_first, pend_blocks});
            pend_first = gi;
            pend_blocks = 0;
        };
        for (auto &L : levels) {
            for (FwOp op : L.g) {
                if (stays_virtual(op)) continue;
                op.first_block = pend_blocks;
                pend_blocks += (op.n + FW_BLOCK - 1) / FW_BLOCK;
                h_ops[gi++] = op;
            }
            if (!L.n.empty()) {
                flush_global();
                const size_t first = ni;
                uint32_t tiles = 0;
                for (FwNestOp op : L.n) {
                    op.first_tile = tiles;
                    tiles += op.n_tiles;
                    op.parent_buf = ctx->segs[op.parent_seg].buf[p];
                    op.parent_cap = ctx->segs[op.parent_seg].capacity;
                    // (a range ring: the slot of its first YOUNG particle as of the last update -- this frame's cohorts join the
                    // old part further down, after these launches have been enqueued -- and the device subtracts the old part)
                    const SegHost &PS = ctx->segs[op.parent_seg], &CS = ctx->segs[op.child_seg];
                    op.parent_head = PS.fifo ? PS.head : (PS.range ? PS.young_lo : 0u);
                    op.child_head = CS.fifo ? CS.head : (CS.range ? CS.young_lo : 0u);
                    op.parent_range = PS.range ? 1u : 0u, op.child_range = CS.range ? 1u : 0u;
                    op.parent_nospin = (ctx->segs[op.parent_seg].nospin ? 1u : 0u) | (ctx->segs[op.parent_seg].ring() ? 2u : 0u);
                    memcpy(op.parent_rot, ctx->segs[op.parent_seg].const_rot, sizeof op.parent_rot);
                    // (its lifetimes: the lifetime plane of a compacting segment, one value for a ring)
                    op.parent_life_plane = ctx->segs[op.parent_seg].fifo ? 0xFFFFFFFFu : ctx->segs[op.parent_seg].n_lplanes;
                    op.parent_life_const = ctx->segs[op.parent_seg].fifo_life;
                    // (START tickets, fw_kernels.h: every workgroup of the op takes one)
                    op.ticket_base = ctx->nest_ticket_base[op.emit_slot], ctx->nest_ticket_base[op.emit_slot] += op.n_tiles;
                    h_nops[ni++] = op;
                }
                launches.push_back(Launch{true, first, ni - first, tiles});
            }
        }
        flush_global();
        // launches with at most FW_INLINE_OPS ops carry them in their kernel arguments; only longer lists (many
        // spawners with Nested entries) are staged through the copy stream
        bool staged = false;
        for (const Launch &L : launches) staged |= L.count > FW_INLINE_OPS;
        if (staged) {
            FW_HIP(ctx, hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
            FW_HIP(ctx, hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
            FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
        }
        for (const Launch &L : launches) {
            const bool inl_ops = L.count <= FW_INLINE_OPS;
            if (!L.nested) {
                FW_HIP(ctx, fw_launch_spawn(ctx->stream, ctx->g, inl_ops ? nullptr : (const FwOp *)dp + L.first,
                                            h_ops + L.first, (uint32_t)L.count, L.blocks, p));
            } else {
                ctx->nest_seq = (ctx->nest_seq + 1u) & 0x3FFFFFFFu;
                if (!ctx->nest_seq) ctx->nest_seq = 1u;
                FW_HIP(ctx, fw_launch_nested(ctx->stream, ctx->g,
                                             inl_ops ? nullptr : (const FwNestOp *)(dp + off_nops) + L.first,
                                             h_nops + L.first, (uint32_t)L.count, L.blocks, p, ctx->nest_seq,
                                             ctx->spin_limit, ctx->dbg));
            }
        }
        if (!staged) slot = -1;  // nothing in the ring slot is read by the device: no consumed-event needed
        if (n_virtual) {
            // ---- the table of the small launch: {first op, one past the last, particles} per segment, then the ops
            const size_t off_ops = (size_t)n_seg * sizeof(OpHdr);
            bool others = false;  // small types' ops at an emission level other than the first
            for (size_t li = 1; li < levels.size(); li++)
                for (const FwOp &op : levels[li].g) others |= stays_virtual(op);
            OpList &g0 = levels[0].g;
            int tslot = -1;
            char *hp = nullptr;
            if (pre_slot >= 0 && g0.lent && !others && pre_hdr_ok && pre_tracked == g0.size()) {
                // the list of the first level lives in the slot taken before the spawner loop, headers and all: the ops of the other
                // types next to the small ones' are simply not looked at (fw_k_update_small reads its segments' headers only)
                tslot = pre_slot, hp = ctx->h_param[tslot];
            } else {
                OpList &ops = ctx->ops_scratch;
                ops.clear();
                ops.reserve_own(n_virtual);
                for (auto &L : levels)
                    for (const FwOp &op : L.g)
                        if (stays_virtual(op)) ops.push_back(op);
                if (!std::is_sorted(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; }))
                    std::stable_sort(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; });
                const size_t tbytes = off_ops + ops.size() * sizeof(FwOp);
                if ((st = ensure_ring(tbytes))) return st;
                if ((st = acquire_slot(&tslot))) return st;
                const bool bar = ctx->param_bar && tbytes <= kBarParamBytes;
                hp = bar ? ctx->b_param[tslot] : ctx->h_param[tslot];
                OpHdr *hdr = (OpHdr *)hp;
                size_t oi = 0;
                for (uint32_t sgi = 0; sgi < n_seg; sgi++) {  // (ops are sorted by segment)
                    const size_t b = oi;
                    uint64_t n = 0;
                    while (oi < ops.size() && ops[oi].seg == sgi) n += ops[oi].n, oi++;
                    hdr[sgi] = OpHdr{(uint32_t)b, (uint32_t)oi, (uint32_t)std::min<uint64_t>(n, 0xFFFFFFFFull), 0u};
                }
                memcpy(hp + off_ops, ops.data(), ops.size() * sizeof(FwOp));
                if (bar) {
                    _mm_sfence();
                    (void)*(volatile const uint32_t *)hp;
                }
            }
            small_hdr = (const uint4 *)hp, small_ops = (const FwOp *)(hp + off_ops);
            small_tslot = tslot;  // (read in place: recycled where the small launch is enqueued, below)
        }
    } else {
        // Global-only frame: spawn is fused into the update kernel (virtual particles).  Ops sorted by segment;
        // the order inside a segment stays the emission order (rel_base was assigned in that order).
        // (one emission level holds all of them most of the time: its list is used as it is)
        OpList *one = nullptr;
        size_t n_lists = 0;
        for (auto &L : levels)
            if (!L.g.empty()) one = &L.g, n_lists++;
        if (n_lists != 1) {
            one = &ctx->ops_scratch;
            one->clear();
            one->reserve_own(n_g);
            for (auto &L : levels) one->append(L.g.begin(), L.g.end());
        }
        OpList &ops = *one;
        // the list lives in the parameter slot taken before the spawner loop (`pre_slot`), and so may its headers
        const bool in_place = pre_slot >= 0 && one == &levels[0].g && ops.lent;
        const bool hdr_done = in_place && pre_hdr_ok && pre_tracked == ops.size();
        // sorted by segment, emission order kept inside a segment; a single emission level is already in spawner
        // (= segment creation) order most of the time: skip the sort then
        if (!hdr_done && !std::is_sorted(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; }))
            std::stable_sort(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; });
        a.n_ops = (uint32_t)ops.size();
        if (ops.size() <= FW_INLINE_OPS && (ops.empty() || !ctx->n_small)) {  // (small types read their ops from the table: fw_k_update_small)
            spawn_form = ops.empty() ? FW_SPAWN_NONE : FW_SPAWN_INLINE;
            for (size_t i = 0; i < ops.size(); i++) inl.ops[i] = ops[i];
        } else {
            spawn_form = FW_SPAWN_TABLE;
            const size_t off_ops = (size_t)n_seg * sizeof(OpHdr);
            const size_t bytes = off_ops + ops.size() * sizeof(FwOp);
            if (in_place) {
                slot = pre_slot;  // (taken, waited for and large enough since before the spawner loop; the ops are where they belong)
            } else {
                if ((st = ensure_ring(bytes))) return st;
                if ((st = acquire_slot(&slot))) return st;
            }
            // (a small table goes to device memory the host writes through the large BAR: fw_ctx::b_param)
            const bool bar = !in_place && ctx->param_bar && ctx->ops_zerocopy && bytes <= kBarParamBytes;
            char *hp = bar ? ctx->b_param[slot] : ctx->h_param[slot];
            char *dp = ctx->d_param[slot];
            if (!hdr_done) {
                OpHdr *hdr = (OpHdr *)hp;
                size_t oi = 0;
                for (uint32_t sgi = 0; sgi < n_seg; sgi++) {  // (ops are sorted by segment)
                    const size_t b = oi;
                    uint64_t n = 0;
                    while (oi < ops.size() && ops[oi].seg == sgi) n += ops[oi].n, oi++;
                    hdr[sgi] = OpHdr{(uint32_t)b, (uint32_t)oi, (uint32_t)std::min<uint64_t>(n, 0xFFFFFFFFull), 0u};
                }
            }
            if (!in_place) memcpy(hp + off_ops, ops.data(), ops.size() * sizeof(FwOp));
            if (bar) {  // (write-combining stores: drained, and known to have arrived -- a read may not pass posted writes)
                _mm_sfence();
                (void)*(volatile const uint32_t *)hp;
            }
            if (ctx->ops_zerocopy) {
                // pinned host memory is device-visible: the tiles read their few ops over the bus (tens of bytes each)
                a.seg_op_first = (const uint4 *)hp;
                a.ops = (const FwOp *)(hp + off_ops);
                ctx->slot_frame[slot] = ctx->frame + 1;  // free once done_tag >= frame + 1
                slot = -1;                                // no consumed-event for this slot
            } else {
                FW_HIP(ctx, hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
                FW_HIP(ctx, hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
                FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
                a.seg_op_first = (const uint4 *)dp;
                a.ops = (const FwOp *)(dp + off_ops);
            }
            small_hdr = a.seg_op_first, small_ops = a.ops;
        }
    }
    if (ctx->h_done) a.done_tag = ctx->h_done, a.done_value = ctx->frame;
    prof(5);

    // update_particles + compaction (core.rs:577-670)
    // timing: the events ride on the dispatch packet (its begin / end timestamps), no marker packets in the stream
    bool timed_frame = false;
    auto next_timing_pair = [&](hipEvent_t *e0, hipEvent_t *e1) {
        *e0 = *e1 = nullptr;
        if (!ctx->timing || ctx->tev_used + 2 > ctx->tev.size()) return;
        *e0 = ctx->tev[ctx->tev_used], *e1 = ctx->tev[ctx->tev_used + 1];
        ctx->tev_used += 2;
        timed_frame = true;
    };
    // ---- FIFO segments: in place, everything they need in the kernel arguments (fw_kernels.h: FwFifoSeg)
    bool fifo_launched = false;
    if (ctx->n_fifo) {
        // the ring launch(es) of this frame: on the side stream when a general launch runs next to them (fw_ctx: fifo_stream)
        // (never on a caller-supplied stream: work the caller orders behind fw_step on ITS stream must cover the whole
        // frame, as it did before the side stream existed)
        // (the wave-per-type launch of small types counts as "a general launch" here: it touches none of the rings either)
        bool side = ctx->use_fifo_stream && ctx->own_stream && (total_tiles != 0 || ctx->n_small != 0) && ctx->live_ring == nullptr;
        // (... nor with a colliding ring: a new collider set travels in the MAIN stream, fw_ctx_set_colliders)
        for (const SegHost &S : ctx->segs) side &= !(S.in_use && S.fifo && (S.fifo_mat || S.inst != nullptr || S.collides));
        if (side && (!ctx->fifo_last_side || ctx->main_reads_ring)) {
            // the previous ring launch, or a reader of ring data, sits on the main stream: this launch comes after it
            FW_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
            FW_HIP(ctx, hipStreamWaitEvent(ctx->fifo_stream, ctx->ev_main, 0));
            ctx->main_reads_ring = false;
        } else if (!side && ctx->fifo_last_side && ctx->side_dirty) {
            FW_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->fifo_stream));
            FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
            ctx->side_dirty = false;
        }
        ctx->fifo_last_side = side;
        const hipStream_t fstream = side ? ctx->fifo_stream : ctx->stream;
        FwFifoArgs fa{};
        FwInlineOps fio;
        // (with a colliding ring in the context every FIFO launch of it runs the COLL instantiation, whose workgroups cover one
        // round -- FW_FIFO_COLL_TILE slots -- each: the tile grid of all of them follows)
        bool fifo_coll = false;
        // ... and so do the launches of a context whose rings hold too few particles to fill the chip with four-round workgroups
        // (the reference's own stress_test: 157k particles = 154 of them on 256 CUs, each lane working through four particles one
        // after the other): below FW_FIFO_SMALL four-round tiles in all, one round per workgroup.  Not with a ring whose count
        // only the device knows (its grid covers its capacity: four times the idle workgroups).
        // (counted in the first pass over the segments of this frame; a ring that changed its kind since then: next frame)
        const bool fifo_small = !ring_stats.fifo_dev && ring_stats.fifo_parts < (uint64_t)ctx->fifo_small_tiles * FW_TILE;
        const bool fifo_coll_real = ring_stats.fifo_coll;
        fifo_coll = fifo_coll_real || fifo_small;  // (the same tile grid; which instantiation runs: FwFifoArgs::any_coll / small_tiles)
        uint32_t f_ops = 0, f_tiles = 0;
        uint64_t f_bytes = 0;  // what the launch streams, roughly: its tiles x the bytes a particle of the type moves
        uint32_t nest_status_next = 0;  // look-back words handed to the Nested entries of this launch so far (FwFifoNest::status_first)
        if (fuse) {
            ctx->nest_seq = (ctx->nest_seq + 1u) & 0x3FFFFFFFu;
            if (!ctx->nest_seq) ctx->nest_seq = 1u;
        }
        auto flush = [&]() -> hipError_t {
            if (!fa.n_segs) return hipSuccess;
            fa.parity = p, fa.epoch = a.epoch, fa.dbg = ctx->dbg, fa.dt = dt;
            // (the pinned "frame F has started" word recycles host buffers the GENERAL launch reads: when the two launches
            // run on different streams only that one reports)
            fa.done_tag = side ? nullptr : a.done_tag, fa.done_value = a.done_value;
            fa.host_counts = a.host_counts;
            fa.live_out = a.live_out, fa.live_next = a.live_next;
            hipEvent_t e0, e1;
            next_timing_pair(&e0, &e1);
            const hipError_t e = fw_launch_update_fifo(fstream, ctx->g, fa, fio, f_tiles, f_bytes > ctx->nt_bytes ? 2 : f_bytes > ctx->nt_wo_bytes ? 1 : 0, e0, e1);
            if (side) ctx->side_dirty = true;
            fa = FwFifoArgs{};
            f_ops = f_tiles = 0, f_bytes = 0;
            fifo_launched = true;
            return e;
        };
        for (uint32_t si = 0; si < n_seg; si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.fifo) continue;
            uint32_t k_ops = 0;
            for (const FwOp &op : ctx->fifo_ops) k_ops += op.seg == si ? 1u : 0u;
            if (k_ops > FW_INLINE_OPS)  // (build_spawner never makes such a type a ring)
                return poison_segment(ctx, si, "a FIFO ring with more spawn ops than its launch can carry");
            if (fa.n_segs == FW_FIFO_PER_LAUNCH || f_ops + k_ops > FW_INLINE_OPS) FW_HIP(ctx, flush());
            const int32_t wm = S.derived ? 0 : S.fifo_wm;  // (FW_TYPE_DERIVED: none of the optional planes is stored)
            if (!fa.n_segs) fa.write_mask = wm;
            else if (fa.write_mask != wm) fa.write_mask = -1;
            // frames that materialise (Nested pass): the segment's Global particles of this frame already sit in the ring
            const bool mat_frame = S.fifo_mat && nested_frame && !S.virt_parent;
            const bool mat = S.fifo_dev || mat_frame;
            const uint32_t n_spawn = mat_frame ? 0u : S.frame_spawn;  // spawned by fw_k_update_fifo itself
            // live particles before fw_k_update_fifo's own spawns (a type that receives children: only the device knows)
            const uint32_t n_in = S.fifo_dev ? 0xFFFFFFFFu : S.ub - n_spawn;
            // the cohorts age by this dt exactly as their particles do (fp32 additions, fw_survives); the oldest die first
            if (S.fifo_dev) {
                S.coh.push_back(SegHost::Cohort{0u, 0.0f, ctx->frame, false});  // size: whatever the device appends
            } else if (S.frame_spawn) {
                if (!S.coh.empty() && S.coh.back().age == 0.0f && !std::signbit(S.coh.back().age))
                    S.coh.back().n += S.frame_spawn;
                else
                    S.coh.push_back(SegHost::Cohort{S.frame_spawn, 0.0f, ctx->frame, true});
            }
            for (auto &c : S.coh) c.age = c.age + dt;
            uint32_t dead = 0;
            while (!S.coh.empty() && S.coh.front().age >= S.fifo_life) {
                SegHost::Cohort &c = S.coh.front();
                if (!c.known && c.frame != ctx->frame) {
                    // children added `lifetime` ago: the update of that frame left their number in the pinned ring
                    const uint32_t ep = (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) ? (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) : 1u;
                    const volatile unsigned long long *row = S.h_report + (c.frame % kReportRing);
                    for (int spin = 0; (uint32_t)(*row >> 32) != ep && spin < 100000; spin++) __builtin_ia32_pause();
                    if ((uint32_t)(*row >> 32) != ep) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if ((uint32_t)(*row >> 32) != ep) return poison_segment(ctx, si, "cohort report missing");
                    c.n = (uint32_t)*row, c.known = true;
                }
                dead += c.n;
                S.coh.pop_front();
            }
            FwFifoSeg &F = fa.s[fa.n_segs++];
            F.buf = S.buf[0], F.destroyed = S.destroyed, F.inst = S.inst;
            F.inst_cap = S.inst_cap, F.capacity = S.capacity, F.seg = si;
            F.type_idx = S.type_idx | (S.nospin ? FW_TYPE_IDX_NOSPIN : 0u), F.life = S.fifo_life;
            F.keys_off = S.keys_off, F.keys_len = S.keys_len;
            F.head = S.head, F.n_in = n_in, F.n_spawn = n_spawn, F.dead = dead;
            F.mat = mat ? 1u : 0u;
            F.n_lplanes = S.n_lplanes;
            F.report = S.fifo_dev ? S.h_report + (ctx->frame % kReportRing) : nullptr;
            F.op0 = f_ops;
            if (!mat_frame)
                for (const FwOp &op : ctx->fifo_ops)
                    if (op.seg == si) fio.ops[f_ops++] = op;
            F.op1 = f_ops;
            // workgroups: the new particles first, FW_BLOCK each, in two groups of consecutive slots (up to the end of the
            // buffer / from slot 0); then the ring tiles from the first slot the update touches (the first destroyed particle
            // when their records are wanted, the first survivor otherwise) to the last old particle (a type whose count
            // only the device knows: the whole ring, empty tiles leave at once); at least one in all (it publishes the counts)
            const uint32_t n_old = S.fifo_dev ? S.capacity : n_in;
            // (... or a ring whose particles a Nested entry of this launch emits from: the ones about to die still emit, and the
            // tiles' ranks count from the ring's head)
            int nest_parent = -1, nest_child = -1;
            for (uint32_t k = 0; fuse && k < n_fuse; k++) {
                if (fuse_plan[k].parent_seg == si) nest_parent = (int)k;
                if (fuse_plan[k].child_seg == si) nest_child = (int)k;
            }
            const uint32_t lo = std::min((S.destroyed || nest_parent >= 0) ? 0u : dead, n_old), cnt = n_old - lo;
            const uint32_t ftile = fifo_coll ? FW_FIFO_COLL_TILE : FW_TILE;
            const uint32_t ps = (uint32_t)(((uint64_t)S.head + lo) % S.capacity), ring_tiles = S.capacity / ftile;
            const uint32_t ns0 = (uint32_t)(((uint64_t)S.head + (S.fifo_dev ? 0u : n_in)) % S.capacity);  // slot of the first new particle
            F.spawn_a = std::min(n_spawn, S.capacity - ns0);
            F.n_vt_a = (F.spawn_a + FW_BLOCK - 1) / FW_BLOCK, F.n_vt_b = (n_spawn - F.spawn_a + FW_BLOCK - 1) / FW_BLOCK;
            F.tile0 = ps / ftile;
            const uint32_t live_tiles = cnt ? std::min<uint32_t>(ring_tiles, (ps % ftile + cnt + ftile - 1) / ftile) : 0u;
            F.n_tiles = std::max(1u, F.n_vt_a + F.n_vt_b + live_tiles);
            F.tile_first = f_tiles;
            f_tiles += F.n_tiles;
            F.nest = 0u;
            if (nest_parent >= 0) {
                const FwNestOp &op = fuse_plan[nest_parent];
                FwFifoNest &N = fa.nest[nest_parent];
                F.nest = (uint32_t)nest_parent + 1u;
                N.parent = fa.n_segs - 1u;
                N.emit = op.emit, N.emit_slot = op.emit_slot, N.parent_lplane = op.parent_lplane;
                N.n_count = op.n_count, N.n_start = op.n_start, N.n_end = op.n_end, N.speed = op.speed, N.scale = op.scale;
                N.status_first = nest_status_next, N.n_ptiles = F.n_tiles - (F.n_vt_a + F.n_vt_b);
                nest_status_next += N.n_ptiles;
                N.ticket_base = ctx->nest_ticket_base[op.emit_slot], ctx->nest_ticket_base[op.emit_slot] += N.n_ptiles;
                N.tag = ctx->nest_seq, N.spin_limit = ctx->spin_limit;
                fa.n_nest = n_fuse;
            }
            if (nest_child >= 0) {
                F.nest = ((uint32_t)nest_child + 1u) | FW_FIFO_NEST_CHILD;
                fa.nest[nest_child].child = fa.n_segs - 1u;
            }
            f_bytes += (uint64_t)live_tiles * ftile * (S.nospin ? 104u : 164u);
            fa.any_inst |= S.inst != nullptr ? 1u : 0u;
            fa.any_coll |= fifo_coll_real ? 1u : 0u;
            fa.small_tiles = (fifo_coll && !fifo_coll_real) ? 1u : 0u;
            S.head = (uint32_t)(((uint64_t)S.head + dead) % S.capacity);
            if (!S.fifo_dev) S.ub = n_in + n_spawn - std::min(dead, n_in + n_spawn);  // exact
        }
        FW_HIP(ctx, flush());
    }
    // ---- range rings: in place, one launch for all of them (fw_kernels.h: FwRangeRec)
    bool range_launched = false;
    if (ctx->n_range) {
        const int rslot = (int)(ctx->rring_seq++ % kParamRing);
        if (ctx->rslot_frame[rslot]) {  // the kernel reads the slot in place: free once a launch AFTER that frame has started
            const volatile unsigned long long *tag = ctx->h_done;
            for (int spin = 0; *tag < ctx->rslot_frame[rslot] && spin < 200000; spin++) __builtin_ia32_pause();
            if (*tag < ctx->rslot_frame[rslot]) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->rslot_frame[rslot] = 0;
        }
        FwRangeRec *recs = (FwRangeRec *)ctx->h_rparam[rslot];
        FwOp *rops = (FwOp *)(ctx->h_rparam[rslot] + round_up((uint32_t)(ctx->max_seg * sizeof(FwRangeRec)), 64));
        std::vector<FwOp> &ops = ctx->range_ops;
        if (!std::is_sorted(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; }))
            std::stable_sort(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; });
        if (!ops.empty()) memcpy(rops, ops.data(), ops.size() * sizeof(FwOp));
        // age of a particle born in frame f before this frame's update (fw_ctx::birth_age: one entry per frame, contiguous)
        auto age_before = [&](uint64_t f) -> float {
            if (ctx->birth_age.empty() || f < ctx->birth_age.front().frame) return INFINITY;  // (long graduated)
            const size_t i = (size_t)(f - ctx->birth_age.front().frame);
            return i < ctx->birth_age.size() ? ctx->birth_age[i].age : 0.0f;  // this frame's own cohort: born with age 0
        };
        bool dirty = ctx->r_force, all_nospin = true, range_inst = false, range_coll = false;
        {  // the tile size of this launch (fw_ctx::range_small)
            const uint64_t parts = ring_stats.range_parts;
            const bool any_dev = ring_stats.range_dev, any_coll_r = ring_stats.range_coll;
            const uint64_t lim = (uint64_t)ctx->range_small_tiles * FW_TILE;
            const bool small = any_coll_r || (!any_dev && parts < (ctx->range_small ? lim + lim / 4 : lim));
            if (small != ctx->range_small) ctx->range_small = small, dirty = true;
            // ... and of its YOUNG workgroups (fw_ctx::range_young_rounds)
            const uint64_t mean = parts / std::max<uint32_t>(1u, ctx->n_range), big = ctx->range_young_big;
            uint32_t yr = ctx->range_young_rounds;
            if (small || any_dev || ring_stats.range_inst || big == 0) yr = 4;
            else if (yr == 4 && mean >= big) yr = 2;
            else if (yr == 2 && mean < big - big / 4) yr = 4;
            if (yr != ctx->range_young_rounds) ctx->range_young_rounds = yr, dirty = true;
        }
        const uint32_t OT = ctx->range_small ? (uint32_t)FW_BLOCK : (uint32_t)FW_TILE;  // slots an OLD workgroup covers
        uint64_t r_bytes = 0;  // what the launch streams, roughly (the non-temporal form of the kernel: fw_ctx::nt_bytes)
        size_t oi = 0;
        for (uint32_t si = 0; si < n_seg; si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.range) continue;
            all_nospin &= S.nospin;
            range_inst |= S.inst != nullptr;
            range_coll |= S.collides;
            r_bytes += (uint64_t)(S.range_dev ? S.capacity / 2 : S.ub) * (S.nospin ? 104u : 164u);
            S.dead_at_end = true;
            // cohorts that are no longer provably too young to die join the old part: the boundary moves, nothing is copied
            const bool mat_frame = S.range_mat && nested_frame && !S.virt_parent;  // its Global particles of this frame already sit in the ring
            uint32_t grad = 0;
            if (!S.range_dev) {
                while (!S.ycoh.empty()) {
                    const float an = age_before(S.ycoh.front().frame) + dt;  // the device's own addition (core.rs:594)
                    if (an < S.range_life_lo) break;
                    grad += S.ycoh.front().n;
                    S.ycoh.pop_front();
                }
                if (S.frame_spawn) S.ycoh.push_back(SegHost::YCohort{ctx->frame, S.frame_spawn});
            } else {
                // a type that receives Nested children: a cohort's size is whatever the device appended in its frame; the update
                // of that frame left it in the pinned ring, and it is only needed now, a lifetime.min later
                while (!S.dcoh.empty()) {
                    SegHost::DCohort &c = S.dcoh.front();
                    const float an = age_before(c.frame) + dt;
                    if (an < S.range_life_lo) break;
                    if (!c.known) {
                        const uint32_t ep = (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) ? (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) : 1u;
                        const volatile unsigned long long *row = S.h_report + (c.frame % kReportRing);
                        for (int spin = 0; (uint32_t)(*row >> 32) != ep && spin < 100000; spin++) __builtin_ia32_pause();
                        if ((uint32_t)(*row >> 32) != ep) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        if ((uint32_t)(*row >> 32) != ep) return poison_segment(ctx, si, "cohort report missing (range ring)");
                        c.n = (uint32_t)*row, c.known = true;
                    }
                    grad += c.n;
                    if (c.n) S.gcoh.push_back(SegHost::YCohort{c.frame, c.n}), S.gcoh_sum += c.n;
                    S.dcoh.pop_front();
                }
                S.dcoh.push_back(SegHost::DCohort{ctx->frame, 0u, false});
                // graduated cohorts whose every particle an EARLIER update has destroyed (age >= lifetime.max before this frame)
                while (!S.gcoh.empty() && !(age_before(S.gcoh.front().frame) < (float)S.life_bound)) {
                    S.gcoh_sum -= S.gcoh.front().n;
                    S.gcoh.pop_front();
                }
            }
            {  // (young_lo + grad) mod capacity without a 64-bit division: both terms are below the capacity
                uint64_t nb = (uint64_t)S.young_lo + grad;
                while (nb >= S.capacity) nb -= S.capacity;
                S.young_lo = (uint32_t)nb;
            }
            const uint32_t y_exist = S.range_dev ? 0u : S.young_n - std::min(S.young_n, grad);
            FwRangeRec Rc{};  // (built here, stored once below: the slot is written, never read, by the host)
            Rc.b = S.young_lo, Rc.y_exist = y_exist, Rc.n_spawn = (mat_frame || S.range_dev) ? 0u : S.frame_spawn;
            Rc.grad = grad, Rc.flags = (mat_frame ? FW_RREC_MAT : 0u) | (S.range_dev ? (FW_RREC_MAT | FW_RREC_DEV) : 0u);
            Rc.report = S.range_dev ? S.h_report + (ctx->frame % kReportRing) : nullptr, Rc.pad2 = 0;
            while (oi < ops.size() && ops[oi].seg < si) oi++;
            Rc.op0 = (uint32_t)oi, Rc.op_n = 0;
            while (oi < ops.size() && ops[oi].seg == si) oi++, Rc.op_n++;
            S.young_n = S.range_dev ? 0u : y_exist + S.frame_spawn;
            // workgroups of each role (bands: the table is re-sent only when a need leaves its band)
            const uint32_t YT = ctx->range_small ? (uint32_t)FW_BLOCK : ctx->range_young_rounds * (uint32_t)FW_BLOCK;
            const uint32_t ysh = (uint32_t)__builtin_ctz(YT), osh = (uint32_t)__builtin_ctz(OT);  // (powers of two: shifts, not divisions, per segment)
            uint32_t need_old, need_new, need_young;
            if (S.range_dev) {
                // the old part: at most the cohorts that have joined it and may still hold survivors (all sizes known); the young
                // part: somewhere behind b -- the grid covers the ring, a tile without young particles leaves at once
                need_old = std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>((S.gcoh_sum + OT - 1) >> osh, (S.capacity >> osh) + 1));
                need_new = 0u;
                need_young = S.capacity >> ysh;
            } else {
                const uint32_t live_before_ub = std::min(S.ub - std::min(S.ub, S.frame_spawn), S.capacity);
                const uint32_t old_ub = live_before_ub - std::min(live_before_ub, y_exist);
                need_old = std::max(1u, (old_ub + OT - 1) >> osh);
                need_new = mat_frame ? 0u : (S.frame_spawn + FW_BLOCK - 1) / FW_BLOCK;
                need_young = std::min(S.capacity >> ysh, ((S.young_lo & (YT - 1u)) + y_exist + (mat_frame ? S.frame_spawn : 0u) + YT - 1) >> ysh);
            }
            // every provisioned workgroup is dispatched every frame, active or not (~3 us of a slot each): small needs get
            // one spare, large ones an eighth -- a re-sent table is a copy in the stream, an idle workgroup a cost in every frame
            // (the bound of the old part follows the snapshots in a sawtooth: a role grows at once, and shrinks only after its
            // need has stayed far below what is provided for 64 frames in a row -- otherwise the table would be re-sent on
            // every tooth)
            auto fit = [&](uint32_t &have, uint32_t need, uint32_t spare, uint32_t &low) {
                if (need > have) {
                    have = need + spare, low = 0, dirty = true;
                } else if (have > need + need / 4 + spare + 2) {
                    if (++low > 64) have = need + spare, low = 0, dirty = true;
                } else {
                    low = 0;
                }
            };
            S.r_need[0] = need_old, S.r_need[1] = need_new, S.r_need[2] = need_young;
            if (S.range_dev) {
                // the grid covers the ring, but how far behind b the young part reaches is roughly known: the count of the latest
                // snapshot row (+ a fifth, + what a few frames add).  Tiles beyond that are "probably idle" and go to the end of
                // the table, where they run while the launch drains (a tile that does hold particles simply updates them
                // there); the split follows the count in steps of an eighth
                const uint64_t est = (uint64_t)((double)S.dev_count * 1.2 + 8.0 * (double)S.dev_rate) + 2 * YT;
                const uint32_t likely = (uint32_t)std::min<uint64_t>(need_young, (est + YT - 1) >> ysh);
                if (likely > S.r_young_main || likely + likely / 4 + 8 < S.r_young_main) S.r_young_main = std::min(need_young, likely + likely / 8 + 2), dirty = true;
                S.r_need[2] = S.r_young_main;
            }
            fit(S.r_old, need_old, need_old >= 8 ? need_old / 4 : 1u, S.r_low[0]);
            fit(S.r_new, need_new, need_new ? (need_new >= 8 ? need_new / 8 : 1u) : 0u, S.r_low[1]);
            fit(S.r_young, need_young, need_young >= 16 ? need_young / 8 : 1u, S.r_low[2]);
            S.r_young = std::min(S.r_young, S.capacity >> ysh);
            // (START tickets, fw_kernels.h: every OLD workgroup the table provides for the segment takes one per launch -- r_old of
            // them, whether the table is re-sent this frame or not: fit() changes the number only together with `dirty`)
            Rc.ticket_base = S.ticket_base, S.ticket_base += S.r_old;
            recs[si] = Rc;
        }
        if (dirty) {
            if (ctx->rtab_pending) {  // (one staging buffer: the previous upload must have left it)
                FW_HIP(ctx, hipEventSynchronize(ctx->ev_rtab));
                ctx->rtab_pending = false;
            }
            // Workgroup order of a segment: its OLD workgroups (k ascending: whoever an old tile waits for has a lower workgroup
            // index), then its NEW ones, then the YOUNG ones; segment after segment, so the latency-bound old tiles of one
            // segment overlap the streaming of its neighbours.  A segment with many NEW workgroups (one large emitter: a
            // thousand of them, each ~5x the arithmetic of a YOUNG one and no memory traffic to speak of) gets them spread
            // over the first three quarters of its YOUNG ones instead of as a block: 374-384 -> 364-375 us at 1 x 16M,
            // nothing elsewhere (profiles/r03/range_spread_new.txt).  (Old and young workgroups interleaved within a segment --
            // so that a context with ONE large segment would not start with a front of old tiles -- was measured: 381 -> 384 us
            // at 1 x 16M, 96 -> 100 us at 512 x 8192: no.)  Look-back words are indexed per segment (old_first + k).
            size_t t = 0;
            bool ok = true;
            auto put = [&](uint32_t si, uint32_t role, uint32_t k) {
                if (t >= ctx->rdesc_cap) {
                    ok = false;
                    return;
                }
                const SegHost &S = ctx->segs[si];
                FwRangeDesc &D = ctx->h_rdesc[t++];
                D.seg = si, D.role_k = (role << 30) | k, D.old_first = S.r_status_base;
                D.type_idx = S.type_idx | (S.nospin ? FW_TYPE_IDX_NOSPIN : 0u) | ((S.derived && S.inst == nullptr) ? FW_TYPE_IDX_NOLIFE : 0u);
                D.keys_off = S.keys_off, D.keys_len = S.keys_len, D.n_old = S.r_old, D.pad = 0;
            };
            // Workgroups that are provisioned but probably idle -- the spares of every role, and the upper part of the OLD range
            // (its bound counts everybody older than lifetime.min as alive) -- go to the END of the table, behind every
            // segment's probably-active ones: an idle workgroup still holds a slot for ~2.5 us (descriptor, record, count),
            // and there it does so while the launch drains and slots are free anyway.  An OLD tile stays behind the lower
            // tiles of its segment, so the look-back order holds; a "probably idle" workgroup that does have work simply
            // does it there.
            auto main_old = [&](const SegHost &S) { return ctx->range_idle_last ? std::min(S.r_old, std::max(1u, (S.r_need[0] * 5u + 7u) / 8u)) : S.r_old; };
            auto main_new = [&](const SegHost &S) { return ctx->range_idle_last ? std::min(S.r_new, S.r_need[1]) : S.r_new; };
            auto main_young = [&](const SegHost &S) { return ctx->range_idle_last ? std::min(S.r_young, S.r_need[2]) : S.r_young; };
            auto put_old = [&](uint32_t si) {
                const uint32_t n = main_old(ctx->segs[si]);
                for (uint32_t k = 0; k < n && ok; k++) put(si, FW_RANGE_OLD, k);
            };
            auto put_rest = [&](uint32_t si) {
                const SegHost &S = ctx->segs[si];
                const uint32_t n_new = main_new(S), n_young = main_young(S);
                if (!ctx->range_spread_new || n_new <= 8) {
                    for (uint32_t k = 0; k < n_new && ok; k++) put(si, FW_RANGE_NEW, k);
                    for (uint32_t k = 0; k < n_young && ok; k++) put(si, FW_RANGE_YOUNG, k);
                } else {  // many NEW workgroups (one large segment): spread over the first three quarters of the YOUNG ones
                    const uint64_t span = (uint64_t)n_new + (uint64_t)n_young * 3 / 4;
                    uint32_t kn = 0, ky = 0;
                    for (uint64_t i = 0; i < span && ok; i++) {
                        if (kn < n_new && (uint64_t)kn * span / n_new <= i) put(si, FW_RANGE_NEW, kn++);
                        else if (ky < n_young) put(si, FW_RANGE_YOUNG, ky++);
                    }
                    while (kn < n_new && ok) put(si, FW_RANGE_NEW, kn++);
                    while (ky < n_young && ok) put(si, FW_RANGE_YOUNG, ky++);
                }
            };
            auto put_tail = [&](uint32_t si) {
                const SegHost &S = ctx->segs[si];
                for (uint32_t k = main_old(S); k < S.r_old && ok; k++) put(si, FW_RANGE_OLD, k);
                for (uint32_t k = main_new(S); k < S.r_new && ok; k++) put(si, FW_RANGE_NEW, k);
                for (uint32_t k = main_young(S); k < S.r_young && ok; k++) put(si, FW_RANGE_YOUNG, k);
            };
            auto &rs = ctx->range_scratch;  // the range segments, in segment order
            rs.clear();
            uint32_t status_base = 0;
            for (uint32_t si = 0; si < n_seg; si++) {
                SegHost &S = ctx->segs[si];
                if (!S.in_use || !S.range) continue;
                S.r_status_base = status_base, status_base += S.r_old;
                rs.push_back(si);
            }
            // (the OLD workgroups of a segment dispatched n segments ahead of its other ones: measured, no gain --
            // profiles/r03/range_old_ahead.txt)
            const size_t nr = rs.size();
            // (XCD-aware order -- the runs of eight consecutive segments interleaved entry by entry, so that the workgroups of one
            // segment share an XCD and its L2 -- was built and measured in round 4: nothing at one GPU's share of configs[4]
            // (86.4 against 86.5 us), 2.5 % slower at configs[2]: profiles/r04/range_xcd_order_ab.txt)
            for (size_t i = 0; i < nr && ok; i++) put_old(rs[i]), put_rest(rs[i]);
            for (size_t i = 0; i < nr && ok; i++) put_tail(rs[i]);
            if (!ok) return poison_segment(ctx, kNoSeg, "range table overflow");
            ctx->r_total = (uint32_t)t;
            if (t) FW_HIP(ctx, hipMemcpyAsync(ctx->d_rdesc, ctx->h_rdesc, t * sizeof(FwRangeDesc), hipMemcpyHostToDevice, ctx->stream));
            FW_HIP(ctx, hipEventRecord(ctx->ev_rtab, ctx->stream));
            ctx->rtab_pending = true;
            ctx->r_force = false;
            ctx->r_uploads++;
        }
        // the ages every later frame starts from
        for (auto &e : ctx->birth_age) e.age = e.age + dt;
        ctx->birth_age.push_back(fw_ctx::BirthAge{ctx->frame, 0.0f + dt});
        // (kept while some ring may still ask for the age: until lifetime.min for every ring, until lifetime.max for those that
        // bound their old part by the cohorts in it -- range_age_keep)
        while (!ctx->birth_age.empty() && !(ctx->birth_age.front().age < std::max(ctx->range_life_max, ctx->range_age_keep))) ctx->birth_age.pop_front();
        if (ctx->r_total) {
            if (ctx->param_bar) {
                // the records went through write-combining stores into device memory: drained, and -- a read may not pass posted
                // writes on the bus -- known to have arrived before the launch is announced
                _mm_sfence();
                (void)*(volatile const uint32_t *)recs;
            }
            FwRangeArgs ra{};
            ra.desc = ctx->d_rdesc, ra.recs = recs, ra.ops = rops, ra.status = ctx->d_rstatus;
            ra.total_tiles = ctx->r_total, ra.parity = p, ra.epoch = a.epoch, ra.spin_limit = ctx->spin_limit, ra.dbg = ctx->dbg;
            ra.dt = dt;
            ra.any_inst = range_inst ? 1u : 0u;
            ra.any_coll = range_coll ? 1u : 0u;
            ra.small_tiles = ctx->range_small ? 1u : 0u;
            ra.young_rounds = ctx->range_young_rounds;
            ra.done_tag = a.done_tag, ra.done_value = a.done_value;
            ra.host_counts = a.host_counts, ra.live_out = a.live_out, ra.live_next = a.live_next;
            ra.ts = ctx->d_rts;
            hipEvent_t e0, e1;
            next_timing_pair(&e0, &e1);
            FW_HIP(ctx, fw_launch_update_range(ctx->stream, ctx->g, ra, all_nospin, r_bytes > ctx->nt_bytes ? 2 : r_bytes > ctx->nt_wo_bytes_range ? 1 : 0, e0, e1));
            ctx->rslot_frame[rslot] = ctx->frame + 1;
            range_launched = true;
        }
    }
    // ---- small types: one wave each (fw_k_small.hip)
    bool small_launched = false;
    if (ctx->n_small) {
        const bool list_resent = ctx->small_dirty;
        if (ctx->small_dirty) {  // the list changed (a spawner built or destroyed, a type that outgrew the mode): re-sent through the stream
            ctx->small_list.clear();
            for (int wide = 0; wide < 2; wide++) {  // narrow types first, then the wide ones (FwSmallArgs::n_narrow)
                for (uint32_t si = 0; si < n_seg; si++)
                    if (ctx->segs[si].in_use && ctx->segs[si].small && (int)(ctx->segs[si].wide && !ctx->wave_all_on) == wide) ctx->small_list.push_back(si);
                if (!wide) ctx->n_narrow = (uint32_t)ctx->small_list.size();
            }
            if (ctx->small_list.size() > ctx->small_cap) return poison_segment(ctx, kNoSeg, "small-type list overflow");
            if (ctx->small_pending) {
                FW_HIP(ctx, hipEventSynchronize(ctx->ev_small));
                ctx->small_pending = false;
            }
            if (ctx->small_last_side && ctx->side_dirty) {
                // the previous frame's small launch ran on the ring stream and may not have read the OLD list yet: the copy below
                // (main stream) comes after it (the join used to be enqueued only further down, after the copy)
                FW_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->fifo_stream));
                FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
                ctx->side_dirty = false;
            }
            memcpy(ctx->h_small, ctx->small_list.data(), ctx->small_list.size() * sizeof(uint32_t));
            FW_HIP(ctx, hipMemcpyAsync(ctx->d_small, ctx->h_small, ctx->small_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            FW_HIP(ctx, hipEventRecord(ctx->ev_small, ctx->stream));
            ctx->small_pending = true, ctx->small_dirty = false;
        }
        FwSmallArgs sa{};
        sa.list = ctx->d_small, sa.n = (uint32_t)ctx->small_list.size(), sa.n_narrow = ctx->n_narrow, sa.parity = p, sa.epoch = a.epoch, sa.dt = dt;
        sa.seg_op_first = small_hdr, sa.ops = small_ops;
        sa.force_colors = a.force_colors, sa.dbg = ctx->dbg;
        sa.any_inst = ctx->n_inst != 0 ? 1u : 0u;
        sa.any_coll = ctx->n_small_coll != 0 ? 1u : 0u;
        sa.done_tag = a.done_tag, sa.done_value = a.done_value;
        sa.host_counts = a.host_counts, sa.live_out = a.live_out, sa.live_next = a.live_next;
        // next to the collision passes of the frame, on the ring stream (fw_ctx::small_last_side)
        const bool small_side = legacy && frame_mode != FW_MODE_FUSED && sa.n != 0 && ctx->host_fast && ctx->use_fifo_stream && ctx->own_stream && ctx->live_ring == nullptr &&
                                ctx->n_small_coll == 0 && !list_resent && ctx->ops_zerocopy;  // (materialised ops: fw_k_spawn writes small segments on the main stream)
        if (sa.n) {
            if (small_side && (!ctx->small_last_side || ctx->main_reads_ring)) {
                // the previous small launch, or a reader of its types' data, sits on the main stream: this launch comes after it
                FW_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
                FW_HIP(ctx, hipStreamWaitEvent(ctx->fifo_stream, ctx->ev_main, 0));
                ctx->main_reads_ring = false;
            } else if (!small_side && ctx->small_last_side && ctx->side_dirty) {
                FW_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->fifo_stream));
                FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
                ctx->side_dirty = false;
            }
            ctx->small_last_side = small_side;
            if (small_side) sa.done_tag = nullptr;  // (the "frame has started" word recycles buffers the MAIN stream's launches read)
            hipEvent_t e0, e1;
            next_timing_pair(&e0, &e1);
            FW_HIP(ctx, fw_launch_update_small(small_side ? ctx->fifo_stream : ctx->stream, ctx->g, sa, e0, e1));
            if (small_side) ctx->side_dirty = true;
            if (small_tslot >= 0) {
                if (small_side) {
                    FW_HIP(ctx, hipEventRecord(ctx->ev_consumed[small_tslot], ctx->fifo_stream));
                    ctx->consumed_pending[small_tslot] = true;
                } else {
                    ctx->slot_frame[small_tslot] = ctx->frame + 1;  // free once a launch after this frame has started
                }
            }
            small_launched = true;
        }
    }
    if (total_tiles || !(fifo_launched || range_launched || small_launched)) {
        hipEvent_t e0, e1;
        next_timing_pair(&e0, &e1);
        FW_HIP(ctx, fw_launch_update(ctx->stream, ctx->g, a, spawn_form == FW_SPAWN_INLINE ? &inl : nullptr, spawn_form,
                                     frame_mode, e0, e1));
    }
    if (timed_frame) ctx->tev_frames++;
    if (ctx->colors_dirty) {
        // that update wrote every colour of its output; the buffer it read (next frame's output) may still hold the
        // caller's colours past the survivors: back to the fill value, after which constant planes are skipped again
        for (uint32_t i = 0; i < n_seg; i++) {
            SegHost &S = ctx->segs[i];
            if (!S.in_use || !S.colors_dirty) continue;
            FW_HIP(ctx, fw_launch_fill_colors(ctx->stream, S.buf[ctx->parity], nullptr, S.capacity, S.fill_bc, S.fill_em));
            S.colors_dirty = false;
        }
        ctx->colors_dirty = false;
        // every particle has been through an update since the caller's write: scale and colours are functions of the age
        // again, a type with an attached instance buffer can stop storing them (FW_TYPE_DERIVED; waits for this frame: rare)
        // (the flag flips at the START of the next fw_step, before anything of that frame is enqueued: flipping it waits for
        // the stream -- the frame just enqueued still stores the planes -- and may fail; fw_step itself only enqueues, and
        // by now this frame must finish its bookkeeping whatever happens)
        for (uint32_t i = 0; i < n_seg; i++)
            if (ctx->segs[i].in_use && ctx->segs[i].derive_pending) {
                ctx->segs[i].derive_pending = false;
                if (wants_derived(ctx, ctx->segs[i])) ctx->segs[i].derive_ready = true, ctx->derive_ready_any = true;
            }
    }
    prof(6);
    if (slot >= 0) {
        FW_HIP(ctx, hipEventRecord(ctx->ev_consumed[slot], ctx->stream));
        ctx->consumed_pending[slot] = true;
    }
    if (take_snap) {
        ctx->snap_pending[snap] = true;
        ctx->snap_seen[snap] = false;
        ctx->snap_epoch[snap] = a.epoch;
        ctx->snap_cum[snap].resize(n_seg);
        for (uint32_t i = 0; i < n_seg; i++) ctx->snap_cum[snap][i] = ctx->segs[i].cum_spawn;
    }

    ctx->fc_ok = fc_frame;
    ctx->boxes_epoch = a.boxes ? a.epoch : 0u;
    ctx->fc_dt_bits = dt_bits;
    ctx->fc_tab_seq = ctx->tab_seq;
    ctx->parity ^= 1u;
    ctx->frame++;
    ctx->sim_time += (double)dt;
    prof(7);
    if (ctx->frame > ctx->host_prof_skip) ctx->prof_frames++;
    return FW_OK;
}

}  // extern "C"
