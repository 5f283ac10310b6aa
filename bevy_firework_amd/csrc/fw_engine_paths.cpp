// fw_engine_paths.cpp -- which update path a particle type is on: transitions with live particles, capacity policy, the tile table, live-count snapshots
// (host engine of libfirework_hip.so: fw_engine.h lists its translation units; there is no CPU simulation path in this library)
#include "fw_engine.h"

namespace fwh {

// moves a segment into freshly allocated buffers of `ncap` slots: its live particles, in order, from slot 0 (a FIFO ring
// is unwrapped); make_general: a FIFO segment leaves that mode (two buffers, compacting update) on the way
fw_status realloc_segment(fw_ctx *ctx, uint32_t si, uint32_t ncap, bool make_general) {
    SegHost &s = ctx->segs[si];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    fw_status st = refresh_counts_exact(ctx);
    if (st) return st;
    SegHost old = s;
    if (s.fifo && ncap >= 0x40000000u) make_general = true;  // ring slots are computed in 32 bits: head + index < 2^32
    if (s.range && ncap > FW_RANGE_MAX_CAPACITY) make_general = true;  // (32-bit byte offsets into a plane)
    if (make_general && s.fifo) {
        s.fifo = false, s.fifo_mat = s.fifo_dev = false, s.coh.clear();
        ctx->n_fifo--;
        ctx->tab_force = true;
        ctx->seg_kind_changed = true;
    }
    if (make_general && s.range) {
        s.range = false, s.ycoh.clear(), s.dcoh.clear(), s.gcoh.clear(), s.gcoh_sum = 0, s.young_lo = s.young_n = 0;
        s.range_mat = s.range_dev = false;
        ctx->n_range--;
        if (s.few_ring) s.few_ring = false, ctx->n_few--;
        if (s.spilled) s.spilled = false, ctx->n_spilled--;
        ctx->tab_force = true, ctx->r_force = true;
        ctx->seg_kind_changed = true;
    }
    st = alloc_seg_buffers(ctx, s, ncap, old.destroyed != nullptr);
    if (st) {
        if (old.fifo && !s.fifo) ctx->n_fifo++;
        if (old.range && !s.range) ctx->n_range++;
        if (old.few_ring && !s.few_ring) ctx->n_few++;
        if (old.spilled && !s.spilled) ctx->n_spilled++;
        s = old;
        return st;
    }
    if (((old.fifo && !s.fifo) || (old.range && !s.range)) && s.h_report) hipHostFree(s.h_report), s.h_report = nullptr;
    s.head = 0;
    const uint32_t p = ctx->parity;
    const uint32_t n = old.ub;  // exact after the refresh
    const uint32_t h = ring_head_exact(old, n);
    if (s.range) {  // the list now starts in slot 0: old part first, the young part right behind it
        s.young_lo = old.range_dev ? std::min(old.rold_seen, n) : (n > old.young_n ? n - old.young_n : 0u);
        ctx->r_force = true;
    }
    const uint32_t n1 = std::min<uint32_t>(n, old.capacity - h);  // up to the end of the old buffer, then from its slot 0
    auto cp = [&](size_t noff, size_t ooff, size_t elem) -> hipError_t {
        hipError_t e = hipSuccess;
        if (n1) e = hipMemcpy(s.buf[p] + noff, old.buf[p] + ooff + (size_t)h * elem, (size_t)n1 * elem, hipMemcpyDeviceToDevice);
        if (e == hipSuccess && n > n1)
            e = hipMemcpy(s.buf[p] + noff + (size_t)n1 * elem, old.buf[p] + ooff, (size_t)(n - n1) * elem, hipMemcpyDeviceToDevice);
        return e;
    };
    const size_t OC = old.capacity, NC = ncap;
    FW_HIP(ctx, cp(FW_OFF_Q0(NC), FW_OFF_Q0(OC), 16));
    // Q1 / Q3 of a RING are four component planes of 4-byte elements each (fw_device.h; their distance follows the capacity), of a segment
    // of the compacting path float4 planes: a ring that grows keeps its layout, one that leaves for that path is transposed as it is
    // unwrapped (hipMemcpy2D: rows of 4 bytes, 4 apart in the source, 16 apart in the destination)
    const bool was_cpl = old.ring(), is_cpl = s.ring();
    auto cpq = [&](size_t noff, size_t ooff) -> hipError_t {
        if (!was_cpl) return cp(noff, ooff, 16);  // (float4 -> float4; nothing becomes a ring here)
        hipError_t e = hipSuccess;
        for (size_t c = 0; c < 4 && e == hipSuccess; c++) {
            if (is_cpl) {
                e = cp(noff + c * 4 * NC, ooff + c * 4 * OC, 4);
            } else {
                if (n1) e = hipMemcpy2D(s.buf[p] + noff + c * 4, 16, old.buf[p] + ooff + c * 4 * OC + (size_t)h * 4, 4, 4, n1, hipMemcpyDeviceToDevice);
                if (e == hipSuccess && n > n1)
                    e = hipMemcpy2D(s.buf[p] + noff + (size_t)n1 * 16 + c * 4, 16, old.buf[p] + ooff + c * 4 * OC, 4, 4, n - n1, hipMemcpyDeviceToDevice);
            }
        }
        return e;
    };
    FW_HIP(ctx, cpq(FW_OFF_Q1(NC), FW_OFF_Q1(OC)));
    FW_HIP(ctx, cp(FW_OFF_Q2(NC), FW_OFF_Q2(OC), 16));
    FW_HIP(ctx, cpq(FW_OFF_Q3(NC), FW_OFF_Q3(OC)));
    FW_HIP(ctx, cp(FW_OFF_Q5(NC), FW_OFF_Q5(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_Q6(NC), FW_OFF_Q6(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_S4(NC), FW_OFF_S4(OC), 4));
    for (uint32_t k = 0; k < s.n_lplanes + s.n_xplanes; k++) FW_HIP(ctx, cp(FW_OFF_L(NC, k), FW_OFF_L(OC, k), 4));
    if (old.destroyed) {  // the records of the last update stay readable (fw_spawner_read_destroyed)
        const size_t m = std::min(old.capacity, ncap);  // (from the start of the buffer, or -- a range ring's -- up to its end)
        const size_t so = old.dead_at_end ? (size_t)old.capacity - m : 0, dof = old.dead_at_end ? (size_t)ncap - m : 0;
        FW_HIP(ctx, hipMemcpy(s.destroyed + dof * sizeof(fw_particle), old.destroyed + so * sizeof(fw_particle), m * sizeof(fw_particle),
                              hipMemcpyDeviceToDevice));
    }
    FW_HIP(ctx, hipFree(old.buf[0]));
    if (old.destroyed) FW_HIP(ctx, hipFree(old.destroyed));
    if (old.fifo && !s.fifo) {
        s.win_ok = false;  // no lifetime window was kept: the bound follows the snapshots from here on
        if (s.nospin) {  // a ring keeps no lifetime plane (one value); the compacting kernels read it
            FW_HIP(ctx, fw_launch_fill_plane1(ctx->stream, s.buf[0], s.buf[1], FW_OFF_L((size_t)s.capacity, s.n_lplanes), s.capacity,
                                              s.fifo_life));
            FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if ((st = upload_seg(ctx, si))) return st;
    if ((st = ensure_range_arrays(ctx))) return st;
    return ensure_tile_arrays(ctx);
}

// (at_least_double: the amortised growth of a Vec -- `need` with a quarter of slack, never less than twice the capacity;
// false: exactly what the caller asks for)
fw_status grow_segment(fw_ctx *ctx, uint32_t si, uint32_t need, bool at_least_double) {
    const SegHost &s = ctx->segs[si];
    const uint32_t ncap = round_up(at_least_double ? std::max<uint32_t>((uint32_t)std::min<uint64_t>((uint64_t)need * 5 / 4, 0xFFFF0000ull),
                                                                        s.capacity * 2)
                                                   : std::max<uint32_t>(need, s.capacity),
                                   std::max<uint32_t>(FW_TILE, fw_range_young_tile()));
    return realloc_segment(ctx, si, ncap, false);
}

// Types fed by Nested entries cannot be bounded by the host (children are counted per parent on the device), so they cannot
// grow exactly when needed the way Global-fed ones do (the reference's Vec::push, core.rs:523).  Their DERIVED capacity
// (parent capacity x children per parent x lifetime ratio x 1.25) is an upper estimate of the steady state already: such a
// segment grows when the count seen in the snapshot rows passes 85 % of it, or when at the rate it was last seen growing
// it would fill up within 64 frames (the rows a free-running host looks at are up to a dozen frames old) -- long before the
// device-side clamp (FW_ECAPACITY) could drop a particle, and without the 2x over-allocation and the ~30 ms reallocation
// the old "half full" rule cost a steady configs[3].
bool nested_fed_wants_growth(const SegHost &S) {
    if (!S.nested_fed || !S.auto_capacity || S.capacity >= 0x70000000u) return false;
    return (double)S.dev_count > 0.85 * (double)S.capacity || (double)S.dev_count + 64.0 * (double)S.dev_rate > (double)S.capacity;
}

// a FIFO segment whose premise no longer holds (the caller wrote particles, dt went negative or non-finite, ...)
// continues as an ordinary segment
fw_status fifo_to_general(fw_ctx *ctx, uint32_t si) {
    if (!ctx->segs[si].ring()) return FW_OK;
    return realloc_segment(ctx, si, ctx->segs[si].capacity, true);
}

// SegHost::small: may this compacting segment be updated by the wave-per-type kernel?
bool small_eligible(const fw_ctx *ctx, const SegHost &S) {
    // (a colliding type qualifies -- the kernel's COLL instantiation, destroy_on_collision included: the stable compaction removes a
    // destroyed particle like one that died of age --; one whose curve keys exceed the LDS staging area does not: SegHost::bigkeys)
    return ctx->use_small && S.in_use && !S.ring() && !S.nested_fed && S.n_lplanes == 0 && !S.bigkeys &&
           !S.colors_dirty && (S.expect_live * 2.0f <= (float)ctx->small_max || S.expect_live <= (float)ctx->wide_max);
}
// on the kernel / off it: a flag (the same buffers, the same layout; the tile table is re-sent)
static void small_activate(fw_ctx *ctx, SegHost &S) {
    if (S.small) return;
    S.small = true, ctx->n_small++, ctx->small_dirty = true;
    if (S.collides) ctx->n_small_coll++;
    ctx->tab_force = true, ctx->fc_ok = false, ctx->boxes_epoch = 0;
}
static void small_deactivate(fw_ctx *ctx, SegHost &S) {
    if (!S.small) return;
    S.small = false, ctx->n_small--, ctx->small_dirty = true;
    if (S.collides) ctx->n_small_coll--;
    if (S.solo) S.solo = false, ctx->n_solo--;  // (fw_step's frame-begin pass visits it again: fw_ctx::big_list)
    ctx->big_dirty = true;
    ctx->tab_force = true, ctx->fc_ok = false, ctx->boxes_epoch = 0;
}
// the segment qualifies (small_eligible): on the kernel at once if the context runs it (fw_ctx::small_on), else with the others
// when there are enough of them (update_small_mode)
void enter_small(fw_ctx *ctx, SegHost &S) {
    if (!S.small_ok)
        S.small_ok = true, ctx->n_small_ok++, S.wide = S.expect_live * 2.0f > (float)ctx->small_max, S.wide_big = S.expect_live > (float)ctx->wide_mid;
    if (S.wide ? (S.wide_big ? ctx->wide_on : ctx->wide_mid_on) : ctx->small_on) small_activate(ctx, S);
}
void small_suspend(fw_ctx *ctx, SegHost &S) { small_deactivate(ctx, S); }
// ... and no longer does: a compacting segment from here on
void leave_small(fw_ctx *ctx, SegHost &S) {
    if (S.small_ok) S.small_ok = false, ctx->n_small_ok--;
    small_deactivate(ctx, S);
}
// fw_ctx::small_min, with hysteresis; called where spawners are built and destroyed (the context is synchronised)
void update_small_mode(fw_ctx *ctx) {
    const uint32_t on_at = ctx->small_min, off_below = ctx->small_min - ctx->small_min / 4;
    const bool want = ctx->use_small && (ctx->small_on ? ctx->n_small_ok >= off_below : ctx->n_small_ok >= on_at);
    // ... and its wide role from wide_min of them on (fw_ctx::wide_min)
    const bool want_wide = ctx->use_small && ctx->wide_max != 0 &&
                           (ctx->wide_on ? ctx->n_small_ok >= ctx->wide_min - ctx->wide_min / 4 : ctx->n_small_ok >= ctx->wide_min);
    const uint32_t mid_at = ctx->wide_min / 3;  // (fw_ctx::wide_mid)
    const bool want_mid = ctx->use_small && ctx->wide_max != 0 &&
                          (ctx->wide_mid_on ? ctx->n_small_ok >= mid_at - mid_at / 4 : ctx->n_small_ok >= mid_at);
    // ... and from wave_all_min types on EVERY type of the kernel -- wide ones included -- is walked by a wave (fw_ctx::wave_all_min)
    const bool want_wave_all = ctx->use_small && ctx->wave_all_min != 0 &&
                               (ctx->wave_all_on ? ctx->n_small_ok >= ctx->wave_all_min - ctx->wave_all_min / 6 : ctx->n_small_ok >= ctx->wave_all_min);
    if (want_wave_all != ctx->wave_all_on) ctx->wave_all_on = want_wave_all, ctx->small_dirty = true;  // (the list is laid out by role)
    if (want == ctx->small_on && want_wide == ctx->wide_on && want_mid == ctx->wide_mid_on) return;
    ctx->small_on = want, ctx->wide_on = want_wide, ctx->wide_mid_on = want_mid;
    for (auto &S : ctx->segs)
        if (S.in_use && S.small_ok) (S.wide ? (S.wide_big ? want_wide : want_mid) : want) ? small_activate(ctx, S) : small_deactivate(ctx, S);
}

// every SegHost::few_ring segment leaves its ring (fw_ctx::range_few), particles and order kept
fw_status drop_few_rings(fw_ctx *ctx) {
    for (uint32_t si = 0; si < ctx->segs.size() && ctx->n_few; si++) {
        if (!ctx->segs[si].in_use || !ctx->segs[si].few_ring) continue;
        const fw_status st = fifo_to_general(ctx, si);  // (realloc_segment clears the flag and the count)
        if (st) return st;
        if (small_eligible(ctx, ctx->segs[si])) enter_small(ctx, ctx->segs[si]);  // (a small type: the wave-per-type kernel from here on)
    }
    return FW_OK;
}

// can this FIFO ring continue as a RANGE ring (build_spawner's rule for range rings, for a type that is a FIFO ring already)
bool fifo_may_become_range(const fw_ctx *ctx, const SegHost &S) {
    if (!S.in_use || !S.fifo || !ctx->use_range || S.spawner < 0) return false;
    const SpawnerHost &sp = ctx->spawners[S.spawner];
    return !sp.no_rings && S.n_lplanes <= 2 && sp.types[S.type].life_lo_safe > 0.0f && S.capacity <= FW_RANGE_MAX_CAPACITY &&
           S.capacity % std::max<uint32_t>(FW_TILE, fw_range_young_tile()) == 0u && !(S.inst != nullptr && !S.inst_window);
}

// A FIFO ring becomes a RANGE ring where it stands (fw_ctx::n_spilled): the same buffer, the same slots, nothing copied.  A FIFO
// ring is a range ring whose particles are all "young" -- nobody has been told yet that it may die -- with an empty old part: the
// slot of the first young particle is the head, the young cohorts are the FIFO cohorts (frame of birth + size; sizes the device
// alone knows stay in the pinned report ring), their ages -- one table for all range rings of the context, fw_ctx::birth_age -- are
// the ages the FIFO replay kept per cohort: the same fp32 additions, bit for bit.  The next fw_step moves the cohorts that may
// die in it to the old part as for any range ring.  The context is synchronised (build time).
fw_status fifo_to_range(fw_ctx *ctx, uint32_t si) {
    if (!fifo_may_become_range(ctx, ctx->segs[si])) return FW_OK;
    fw_status st = refresh_counts_exact(ctx);  // (a ring that receives Nested children: only the device knows its count)
    if (st) return st;
    SegHost &S = ctx->segs[si];
    const TypeHost &T = ctx->spawners[S.spawner].types[S.type];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    // ---- the ages of the frames its cohorts were born in: fw_ctx::birth_age holds one entry per frame, contiguous up to the
    // current frame; it is extended backwards to the oldest cohort.  Frames in which this ring received nothing get the age of
    // the next older cohort (never asked for by it; at least as old as the true one, so the pruning order holds) -- and are
    // overwritten with the exact value by whichever ring does hold a cohort of that frame.
    if (!S.coh.empty()) {
        auto &B = ctx->birth_age;
        const uint64_t have_from = B.empty() ? ctx->frame : B.front().frame;
        if (S.coh.front().frame < have_from) {
            std::vector<fw_ctx::BirthAge> pre;
            size_t ci = 0;
            float age = S.coh.front().age;
            for (uint64_t f = S.coh.front().frame; f < have_from; f++) {
                while (ci < S.coh.size() && S.coh[ci].frame < f) ci++;
                if (ci < S.coh.size() && S.coh[ci].frame == f) age = S.coh[ci].age;
                pre.push_back(fw_ctx::BirthAge{f, age});
            }
            B.insert(B.begin(), pre.begin(), pre.end());
        }
        for (const SegHost::Cohort &c : S.coh)
            if (!B.empty() && c.frame >= B.front().frame && c.frame - B.front().frame < B.size()) B[(size_t)(c.frame - B.front().frame)].age = c.age;
    }
    // ---- the ring's own bookkeeping
    S.fifo = false, ctx->n_fifo--;
    S.range = true, ctx->n_range++;
    S.spilled = true, ctx->n_spilled++;
    S.young_lo = S.head, S.head = 0;
    S.range_life_lo = T.life_lo_safe;
    ctx->range_life_max = std::max(ctx->range_life_max, S.range_life_lo);
    S.range_mat = S.n_lplanes != 0, S.range_dev = S.nested_fed;
    S.ycoh.clear(), S.dcoh.clear(), S.gcoh.clear(), S.gcoh_sum = 0, S.rold_seen = 0, S.r_young_main = 0;
    S.r_old = S.r_new = S.r_young = 0;
    S.win.clear(), S.win_sum = 0, S.win_ok = false;
    if (S.range_dev) {
        S.young_n = 0;
        ctx->range_age_keep = std::max(ctx->range_age_keep, (float)(S.life_bound * 1.01 + 1e-3));
        for (const SegHost::Cohort &c : S.coh) S.dcoh.push_back(SegHost::DCohort{c.frame, c.known ? c.n : 0u, c.known});
        // (h_report stays: the update of a range ring that receives children leaves each frame's cohort size in the same ring)
    } else {
        uint64_t sum = 0;
        for (const SegHost::Cohort &c : S.coh) {
            if (!c.n) continue;
            S.ycoh.push_back(SegHost::YCohort{c.frame, c.n});
            // the lifetime window (the bound of the old part follows it): the time of the spawn from the cohort's age -- an fp32
            // running sum, whose distance from the exact time the window's horizon allows for (fw_step)
            S.win.push_back(SegHost::Spawned{ctx->sim_time - (double)c.age, c.n, c.frame});
            sum += c.n;
        }
        S.young_n = (uint32_t)sum;  // (= the exact live count: everybody is young)
        S.win_sum = sum, S.win_ok = std::isfinite(S.life_bound);
        if (S.h_report) hipHostFree(S.h_report), S.h_report = nullptr;
    }
    S.coh.clear();
    S.fifo_mat = S.fifo_dev = false;
    // ---- the device's share: the old part is empty; a type that cannot turn keeps its lifetimes in a plane of their own on
    // this path (a FIFO ring has one value and no plane)
    const uint32_t zero = 0;
    for (int r = 0; r < 2; r++) FW_HIP(ctx, hipMemcpy(ctx->g.rold + (size_t)r * ctx->max_seg + si, &zero, 4, hipMemcpyHostToDevice));
    if (S.nospin) {
        FW_HIP(ctx, fw_launch_fill_plane1(ctx->stream, S.buf[0], nullptr, FW_OFF_L((size_t)S.capacity, S.n_lplanes), S.capacity, S.fifo_life));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->tab_force = true, ctx->r_force = true;
    return ensure_range_arrays(ctx);
}

// every FIFO ring of the context that may becomes a range ring (fw_ctx::n_spilled)
fw_status spill_fifo_rings(fw_ctx *ctx) {
    for (uint32_t si = 0; si < ctx->segs.size() && ctx->n_fifo; si++) {
        const fw_status st = fifo_to_range(ctx, si);
        if (st) return st;
    }
    return FW_OK;
}

// A type stops being FW_TYPE_NOSPIN (the caller rewrites its particles, a non-finite dt is stepped): the rotation plane,
// which nobody maintained, gets the constant rotation in every slot, then the flag goes.
fw_status leave_nospin(fw_ctx *ctx, uint32_t si) {
    SegHost &s = ctx->segs[si];
    if (!s.nospin) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    FW_HIP(ctx, fw_launch_fill_rotation(ctx->stream, s.buf[0], s.ring() ? nullptr : s.buf[1], s.capacity, s.const_rot));
    FW_HIP(ctx, fw_launch_restore_q3(ctx->stream, s.buf[0], s.ring() ? nullptr : s.buf[1], s.capacity,
                                     s.fifo ? 0xFFFFFFFFu : s.n_lplanes, s.fifo_life, s.ring()));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t flags = s.derived ? FW_TYPE_DERIVED : 0u;
    FW_HIP(ctx, hipMemcpy((char *)(ctx->d_types.d + s.type_idx) + offsetof(FwType, flags), &flags, sizeof flags, hipMemcpyHostToDevice));
    s.nospin = false;
    ctx->tab_force = true, ctx->r_force = true;  // (the tile descriptors carry the flag)
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    return FW_OK;
}

// FW_TYPE_DERIVED on / off (SegHost::derived).  Off: the planes nobody maintained are filled from age / lifetime / initial_scale
// first (`refill` false when the caller is about to overwrite the particles anyway).
fw_status set_derived(fw_ctx *ctx, uint32_t si, bool on, bool refill) {
    SegHost &s = ctx->segs[si];
    if (s.derived == on) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    if (!on && refill) {
        FW_HIP(ctx, fw_launch_rederive(ctx->stream, s.buf[ctx->parity], s.capacity, ctx->d_types.d + s.type_idx, ctx->d_keys.d, s.nospin,
                                       s.life_plane(), s.fifo_life, s.ring()));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    uint32_t flags = (s.nospin ? FW_TYPE_NOSPIN : 0u) | (on ? FW_TYPE_DERIVED : 0u);
    FW_HIP(ctx, hipMemcpy((char *)(ctx->d_types.d + s.type_idx) + offsetof(FwType, flags), &flags, sizeof flags, hipMemcpyHostToDevice));
    s.derived = on;
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    if (s.range) ctx->r_force = true;  // (the range descriptors carry FW_TYPE_IDX_NOLIFE)
    return FW_OK;
}

// A parent type grew: the types its particles emit onto (Nested entries targeting it) were sized from the parent's
// capacity (derive_capacity) and cannot grow on demand themselves -- their counts are only known on the device -- so
// they follow the parent now, by the same rule.  Types with a caller-given capacity are left alone.
fw_status grow_nested_children(fw_ctx *ctx, SpawnerHost &sp, uint32_t parent_type, int depth) {
    if (depth > (int)sp.types.size()) return FW_OK;
    const double pcap = ctx->segs[sp.seg[parent_type]].capacity;
    for (const EmissionHost &E : sp.em) {
        const fw_emission_settings &es = E.es;
        if (es.mode != FW_MODE_NESTED || (uint32_t)es.target_particle_type != parent_type) continue;
        if (es.pacing_kind != FW_PACING_COUNT_OVER_DURATION || !(es.count > 0)) continue;
        const uint32_t ct = (uint32_t)es.particle_index;
        if (ct == parent_type) continue;
        const fw_particle_settings &cp = sp.types[ct].ps, &pp = sp.types[parent_type].ps;
        if (cp.capacity) continue;
        const double life = std::max(0.0, (double)std::max(cp.lifetime.min, cp.lifetime.max));
        const double plife = std::max(1e-3, (double)std::min(pp.lifetime.min, pp.lifetime.max));
        double need = pcap * (double)es.count * std::max(1.0, life / plife + 0.1) * 1.25 + kMinCapacity;
        if (need > 3.0e9) need = 3.0e9;
        SegHost &C = ctx->segs[sp.seg[ct]];
        if (need > (double)C.capacity) {
            fw_status st = grow_segment(ctx, sp.seg[ct], (uint32_t)need);
            if (st) return st;
            if ((st = grow_nested_children(ctx, sp, ct, depth + 1))) return st;
        }
    }
    return FW_OK;
}

// The update grid covers ceil(bound / FW_TILE) tiles per segment, where `bound` is the host's upper
// bound of the live count.  The table lives on the device and is re-sent only when a segment's
// need leaves the band [need, need * 5/4 + 8], so steady-state frames upload nothing.
fw_status update_tile_table(fw_ctx *ctx) {
    const uint32_t n_seg = (uint32_t)ctx->segs.size();
    // (a context of rings and small types only -- thousands of small emitters: nothing for the compacting launch, whose table is
    // empty already: not a pass over every segment record per frame)
    if (!ctx->tab_force && ctx->total_tiles_dev == 0 && ctx->d_tile_first && ctx->tiles_dev.size() == n_seg &&
        ctx->n_in_use == ctx->n_fifo + ctx->n_range + ctx->n_small) {
        ctx->vt_rounds = 1u;
        return FW_OK;
    }
    bool dirty = ctx->tiles_dev.size() != n_seg || ctx->tab_force;  // descriptors carry per-segment type indices
    ctx->tab_force = false;
    ctx->tiles_dev.resize(n_seg, 0);
    // (the same pass picks the size of the new-particle tiles: one round of FW_VTILE, or two -- at most FW_TILE / 2, Q1/Q2
    // of new particles live in the upper half of the LDS planes -- when only that keeps the whole frame resident)
    uint64_t act1 = 0, act2 = 0;
    for (uint32_t i = 0; i < n_seg; i++) {
        const SegHost &S = ctx->segs[i];
        if (S.in_use && !S.ring() && !S.small) {
            const uint32_t live = seg_live_tiles(S);
            const uint32_t spawn_eff = std::min(S.frame_spawn, S.capacity);  // (what exceeds the capacity is dropped: seg_tiles)
            act1 += live + (spawn_eff + FW_VTILE - 1) / FW_VTILE;
            act2 += live + (spawn_eff + 2 * FW_VTILE - 1) / (2 * FW_VTILE);
        }
        // provision for one-round new-particle tiles whatever vt_rounds says: a lone segment picks its tile size on
        // the device from exact counts and may use the smaller tiles when the host, with looser bounds, would not
        const uint32_t need = seg_tiles(S, 1);
        uint32_t &have = ctx->tiles_dev[i];
        if (!S.in_use || S.ring() || S.small) {
            if (have) have = 0, dirty = true;
            continue;
        }
        const uint32_t cap_tiles =
            (S.capacity + FW_TILE - 1) / FW_TILE + (std::min(S.frame_spawn, S.capacity) + FW_VTILE - 1) / FW_VTILE + 1;
        // slack: an eighth for large segments; a small segment (thousands of small emitters) gets one spare tile --
        // idle workgroups are cheap one by one, but two per segment doubled such a grid
        if (need > have || have > need + need / 4 + (need >= 16 ? 8u : 2u) || have > cap_tiles) {
            have = std::min(cap_tiles, need + (need >= 16 ? std::max<uint32_t>(2, need / 8) : (need >= 4 ? 1u : 0u)));
            dirty = true;
        }
    }
    // neither size keeps the frame resident: the smaller (more parallel) one
    ctx->vt_rounds = (act2 > kResidentSlots || act1 <= kResidentSlots) ? 1u : 2u;
    if (ctx->trace)
        fprintf(stderr, "[fw] frame %llu tile table dirty=%d n_seg=%u have0=%u ub0=%u\n",
                (unsigned long long)ctx->frame, (int)dirty, n_seg, n_seg ? ctx->tiles_dev[0] : 0u,
                n_seg ? ctx->segs[0].ub : 0u);
    if (!dirty && ctx->d_tile_first) return FW_OK;
    ctx->fc_dirty = true;  // forecast sums are indexed by global tile: a new table invalidates whatever they hold
    if (n_seg + 1 > ctx->tile_first_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = (size_t)(n_seg + 1) * 2 + 64;
        if (ctx->d_tile_first) hipFree(ctx->d_tile_first);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_tile_first, ncap * sizeof(uint32_t)));
        if (ctx->d_tile_keys) hipFree(ctx->d_tile_keys);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_tile_keys, ncap * sizeof(uint2)));
        for (int i = 0; i < kTabRing; i++) {
            if (ctx->h_tab[i]) hipHostFree(ctx->h_tab[i]);
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_tab[i], ncap * sizeof(uint32_t), hipHostMallocDefault));
            if (ctx->h_keys[i]) hipHostFree(ctx->h_keys[i]);
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_keys[i], ncap * sizeof(uint2), hipHostMallocDefault));
            ctx->tab_pending[i] = false;
        }
        ctx->tile_first_cap = ncap;
    }
    uint32_t total_new = 0;
    for (uint32_t i = 0; i < n_seg; i++) total_new += ctx->tiles_dev[i];
    if (total_new > ctx->tile_desc_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = (size_t)total_new * 2 + 256;
        if (ctx->d_tile_desc) hipFree(ctx->d_tile_desc);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_tile_desc, ncap * sizeof(uint4)));
        for (int i = 0; i < kTabRing; i++) {
            if (ctx->h_desc[i]) hipHostFree(ctx->h_desc[i]);
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_desc[i], ncap * sizeof(uint4), hipHostMallocDefault));
            ctx->tab_pending[i] = false;
        }
        ctx->tile_desc_cap = ncap;
    }
    const int slot = (int)(ctx->tab_seq++ % kTabRing);
    if (ctx->tab_pending[slot]) {
        FW_HIP(ctx, hipEventSynchronize(ctx->ev_tab[slot]));
        ctx->tab_pending[slot] = false;
    }
    uint32_t *h = ctx->h_tab[slot];
    uint32_t total = 0;
    for (uint32_t i = 0; i < n_seg; i++) {
        h[i] = total;
        total += ctx->tiles_dev[i];
    }
    h[n_seg] = total;
    ctx->total_tiles_dev = total;
    uint2 *hk = ctx->h_keys[slot];
    for (uint32_t i = 0; i < n_seg; i++) hk[i] = make_uint2(ctx->segs[i].keys_off, ctx->segs[i].keys_len);
    if (n_seg)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_tile_keys, hk, (size_t)n_seg * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    uint4 *hd = ctx->h_desc[slot];
    for (uint32_t i = 0; i < n_seg; i++)
        for (uint32_t t = 0; t < ctx->tiles_dev[i]; t++)
            hd[h[i] + t] = make_uint4(i, h[i], ctx->tiles_dev[i],
                                      ctx->segs[i].type_idx | (ctx->segs[i].nospin ? FW_TYPE_IDX_NOSPIN : 0u));
    FW_HIP(ctx, hipMemcpyAsync(ctx->d_tile_first, h, (size_t)(n_seg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                               ctx->stream));
    if (total)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_tile_desc, hd, (size_t)total * sizeof(uint4), hipMemcpyHostToDevice,
                                   ctx->stream));
    FW_HIP(ctx, hipEventRecord(ctx->ev_tab[slot], ctx->stream));
    ctx->tab_pending[slot] = true;
    return FW_OK;
}

// consume finished live-count snapshots to tighten the host upper bounds (no sync, no HIP call)
void poll_snapshots(fw_ctx *ctx) {
    for (int k = 0; k < kSnapRing; k++) {
        if (!ctx->snap_pending[k]) continue;
        const volatile unsigned long long *snap = ctx->h_snap + (size_t)k * ctx->max_seg;
        const auto &cum = ctx->snap_cum[k];
        const size_t n = std::min(ctx->segs.size(), cum.size());
        if (!ctx->snap_seen[k]) {
            // the row is complete once every segment's last tile has stored; look at one segment first and give
            // the rest one more step
            size_t probe = n;
            for (size_t i = 0; i < n && probe == n; i++)
                if (ctx->segs[i].in_use) probe = i;
            if (probe == n) {
                ctx->snap_pending[k] = false;
            } else if ((uint32_t)(snap[probe] >> 32) == ctx->snap_epoch[k]) {
                ctx->snap_seen[k] = true;
            }
            continue;
        }
        ctx->snap_pending[k] = false;
        for (size_t i = 0; i < n; i++) {
            SegHost &S = ctx->segs[i];
            if (!S.in_use) continue;
            const unsigned long long v = snap[i];
            if ((uint32_t)(v >> 32) != ctx->snap_epoch[k]) continue;  // that segment's store has not landed yet
            if (S.fifo && !S.fifo_dev) continue;  // the host's count is exact
            if (S.nested_fed) {
                {   // no host-side bound exists; the count and its growth rate only drive capacity growth
                    const uint32_t ep = ctx->snap_epoch[k], c = (uint32_t)v;
                    if (S.dev_epoch && ep > S.dev_epoch) S.dev_rate = c > S.dev_count ? (float)(c - S.dev_count) / (float)(ep - S.dev_epoch) : 0.f;
                    S.dev_count = c, S.dev_epoch = ep;
                }
                S.snap_count = (uint32_t)v, S.snap_cum = cum[i];
                continue;
            }
            const uint64_t b = (uint64_t)(uint32_t)v + (S.cum_spawn - cum[i]);
            if (b < S.ub) S.ub = (uint32_t)b;
        }
    }
}

}  // namespace fwh
