// fw_engine_build.cpp -- sync_spawner_data (core.rs:343-365): descriptors -> device tables of one spawner (build_spawner), and back (release_spawner_segments)
// (host engine of libfirework_hip.so: fw_engine.h lists its translation units; there is no CPU simulation path in this library)
#include "fw_engine.h"

namespace fwh {

void copy_curve(CurveCopy &dst, int32_t kind, int32_t n, const float *times, const float *values, int stride) {
    dst.kind = kind;
    dst.n = n;
    dst.values.assign(values, values + (size_t)n * stride);
    dst.times.assign((size_t)n, 0.f);
    if (times && kind == FW_CURVE_UNEVEN) dst.times.assign(times, times + n);
    if (kind == FW_CURVE_UNEVEN && n >= 2) {
        // bevy_math UnevenCore::new: drop non-finite times, stable sort by time, dedup keeping the first
        std::vector<int> idx;
        for (int i = 0; i < n; i++)
            if (std::isfinite(dst.times[i])) idx.push_back(i);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return dst.times[a] < dst.times[b]; });
        std::vector<float> t, v;
        for (int i : idx) {
            if (!t.empty() && t.back() == dst.times[i]) continue;
            t.push_back(dst.times[i]);
            v.insert(v.end(), dst.values.begin() + (size_t)i * stride, dst.values.begin() + (size_t)(i + 1) * stride);
        }
        dst.times = t;
        dst.values = v;
        dst.n = (int32_t)t.size();
    }
    if (dst.n == 1) dst.kind = FW_CURVE_CONSTANT;  // curve.rs:46-49: one sample -> ConstantCurve
}

fw_status validate_desc(fw_ctx *ctx, const fw_spawner_desc *d) {
    if (!d) return fail(ctx, FW_EINVAL, "null descriptor");
    // (Vec<ParticleSettings> / Vec<EmissionSettings> of any length, core.rs:178-185; the counts are 32-bit here)
    if ((d->n_particle_settings && !d->particle_settings) || (d->n_emission_settings && !d->emission_settings))
        return fail(ctx, FW_EINVAL, "null settings array");
    for (uint32_t i = 0; i < d->n_particle_settings; i++) {
        const fw_particle_settings &p = d->particle_settings[i];
        const int32_t ns[3] = {p.scale_curve.n, p.base_color.n, p.emissive_color.n};
        const int32_t ks[3] = {p.scale_curve.kind, p.base_color.kind, p.emissive_color.kind};
        const void *vs[3] = {p.scale_curve.values, p.base_color.rgba, p.emissive_color.rgba};
        const void *ts[3] = {p.scale_curve.times, p.base_color.times, p.emissive_color.times};
        for (int k = 0; k < 3; k++) {
            if (ns[k] < 1) return fail(ctx, FW_EINVAL, "Cannot create curve from 0 samples");  // curve.rs:45,61,211,227
            if (ks[k] < 0 || ks[k] > 2 || !vs[k]) return fail(ctx, FW_EINVAL, "bad curve kind / null values");
            if (ks[k] == FW_CURVE_UNEVEN && !ts[k]) return fail(ctx, FW_EINVAL, "uneven curve without times");
            if (ks[k] == FW_CURVE_UNEVEN && ns[k] >= 2) {
                // UnevenCore::new drops non-finite times and duplicates; with fewer than two left it returns
                // Err(NotEnoughSamples) and the reference's `.unwrap()` panics (curve.rs:50,67,217,232)
                const float *tt = (const float *)ts[k];
                int distinct = 0;
                for (int a = 0; a < ns[k]; a++) {
                    if (!std::isfinite(tt[a])) continue;
                    bool dup = false;
                    for (int b = 0; b < a; b++) dup |= std::isfinite(tt[b]) && tt[b] == tt[a];
                    distinct += dup ? 0 : 1;
                }
                if (distinct < 2) return fail(ctx, FW_EINVAL, "uneven curve needs at least 2 distinct finite times");
            }
        }
    }
    for (uint32_t i = 0; i < d->n_emission_settings; i++) {
        const fw_emission_settings &e = d->emission_settings[i];
        if (e.particle_index < 0 || (uint32_t)e.particle_index >= d->n_particle_settings)
            return fail(ctx, FW_EINVAL, "emission_settings.particle_index out of range");  // index panic core.rs:392
        if (e.mode == FW_MODE_NESTED &&
            (e.target_particle_type < 0 || (uint32_t)e.target_particle_type >= d->n_particle_settings))
            return fail(ctx, FW_EINVAL, "target_particle_type out of range");  // index panic core.rs:488
        if (e.mode != FW_MODE_GLOBAL && e.mode != FW_MODE_NESTED) return fail(ctx, FW_EINVAL, "bad emission mode");
        if (e.pacing_kind < 0 || e.pacing_kind > 2) return fail(ctx, FW_EINVAL, "bad pacing kind");
        if (e.shape_kind < 0 || e.shape_kind > 2) return fail(ctx, FW_EINVAL, "bad shape kind");
    }
    return FW_OK;
}

// capacity heuristic: expected live count from the emitters feeding a type, x1.25 + slack
uint32_t derive_capacity(const fw_spawner_desc *d, uint32_t t, const std::vector<uint32_t> &caps, double *expect_live) {
    const fw_particle_settings &p = d->particle_settings[t];
    const uint32_t FW_CAP_ROUND = std::max<uint32_t>(FW_TILE, fw_range_young_tile());  // every kernel's tile divides a capacity
    if (p.capacity && !expect_live) return round_up(std::max<uint32_t>(p.capacity, FW_TILE), FW_CAP_ROUND);
    const double life = std::max(0.0, (double)std::max(p.lifetime.min, p.lifetime.max));
    double need = 0;
    for (uint32_t i = 0; i < d->n_emission_settings; i++) {
        const fw_emission_settings &e = d->emission_settings[i];
        if ((uint32_t)e.particle_index != t) continue;
        if (e.mode == FW_MODE_GLOBAL) {
            if (e.pacing_kind == FW_PACING_ONESHOT)
                need += (double)e.oneshot_count;
            else if (e.pacing_kind == FW_PACING_COUNT_OVER_DURATION && e.duration > 0 && e.count > 0)
                need += (double)e.count / e.duration * (life + 0.05) + 2 * (double)e.count / e.duration / 30.0;
        } else if (e.pacing_kind == FW_PACING_COUNT_OVER_DURATION && e.count > 0) {
            const fw_particle_settings &pp = d->particle_settings[e.target_particle_type];
            const double plife = std::max(1e-3, (double)std::min(pp.lifetime.min, pp.lifetime.max));
            const double pcap = caps[e.target_particle_type] ? caps[e.target_particle_type] : kMinCapacity;
            need += pcap * (double)e.count * std::max(1.0, life / plife + 0.1);
        }
    }
    if (expect_live) *expect_live = need;  // (what the emitters sustain: SegHost::expect_live)
    if (p.capacity) return round_up(std::max<uint32_t>(p.capacity, FW_TILE), FW_CAP_ROUND);
    need = need * 1.25 + kMinCapacity;
    if (need > 3.0e9) need = 3.0e9;
    return round_up((uint32_t)need, FW_CAP_ROUND);
}

void fill_randvec3(const fw_rand_vec3 &r, float &mn, float &mx, float &spread, float dir[4], float arc[4]) {
    mn = r.magnitude.min, mx = r.magnitude.max, spread = r.spread;
    dir[0] = r.direction[0], dir[1] = r.direction[1], dir[2] = r.direction[2], dir[3] = 0.f;
    fw_q4 q = fw_quat_from_rotation_arc(fw_v3{0.f, 1.f, 0.f}, fw_v3{r.direction[0], r.direction[1], r.direction[2]});
    arc[0] = q.x, arc[1] = q.y, arc[2] = q.z, arc[3] = q.w;
}

uint32_t pad4(uint32_t n) { return (n + 3u) & ~3u; }

// builds the device tables (types, keys, emits, segments) of one spawner
fw_status build_spawner(fw_ctx *ctx, int h, const fw_spawner_desc *d, const std::vector<uint64_t> *carry_serial) {
    SpawnerHost &sp = ctx->spawners[h];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    ctx->tab_force = true;
    const uint32_t nt = d->n_particle_settings, ne = d->n_emission_settings;
    sp.uid = d->uid;
    sp.starts_enabled = d->starts_enabled;
    if (ctx->levels.size() < d->n_emission_settings) ctx->levels.resize(d->n_emission_settings);
    sp.types.assign(nt, TypeHost{});
    sp.em.assign(ne, EmissionHost{});
    sp.seg.assign(nt, kNoSeg);

    fw_status st;
    if ((st = dev_reserve(ctx, ctx->d_types, ctx->n_types + nt, ctx->n_types))) return st;
    if ((st = dev_reserve(ctx, ctx->d_type_coll, ctx->n_types + nt, ctx->n_types))) return st;
    if ((st = dev_reserve(ctx, ctx->d_emits, ctx->n_emits + ne, ctx->n_emits))) return st;
    if ((st = dev_reserve(ctx, ctx->d_emit_serial, ctx->n_emit_slots + ne, ctx->n_emit_slots))) return st;
    if ((st = dev_reserve(ctx, ctx->d_nest_start, ctx->n_emit_slots + ne, ctx->n_emit_slots))) return st;
    if (ctx->nest_ticket_base.size() < ctx->n_emit_slots + ne) ctx->nest_ticket_base.resize(ctx->n_emit_slots + ne, 0u);
    if ((st = dev_reserve(ctx, ctx->d_segs, ctx->segs.size() + nt, ctx->segs.size()))) return st;
    if ((st = ensure_max_seg(ctx, (uint32_t)ctx->segs.size() + nt))) return st;
    ctx->g.type_coll = ctx->d_type_coll.d;
    ctx->g.types = ctx->d_types.d, ctx->g.emits = ctx->d_emits.d, ctx->g.keys = ctx->d_keys.d;
    ctx->g.segs = ctx->d_segs.d, ctx->g.emit_serial = ctx->d_emit_serial.d, ctx->g.nest_start = ctx->d_nest_start.d;

    std::vector<uint32_t> caps(nt, 0);
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t t = 0; t < nt; t++) caps[t] = derive_capacity(d, t, caps);

    for (uint32_t t = 0; t < nt; t++) {
        TypeHost &T = sp.types[t];
        const fw_particle_settings &p = d->particle_settings[t];
        T.ps = p;
        copy_curve(T.scale, p.scale_curve.kind, p.scale_curve.n, p.scale_curve.times, p.scale_curve.values, 1);
        copy_curve(T.base, p.base_color.kind, p.base_color.n, p.base_color.times, p.base_color.rgba, 4);
        copy_curve(T.emis, p.emissive_color.kind, p.emissive_color.n, p.emissive_color.times, p.emissive_color.rgba, 4);
        T.life_lo_safe = (std::isfinite(p.lifetime.min) && std::isfinite(p.lifetime.max))
                             ? std::nextafterf(std::nextafterf(std::min(p.lifetime.min, p.lifetime.max), -INFINITY), -INFINITY)
                             : NAN;
        T.ps.scale_curve.times = T.ps.scale_curve.values = nullptr;  // descriptors are copied, never kept
        T.ps.base_color.times = T.ps.base_color.rgba = nullptr;
        T.ps.emissive_color.times = T.ps.emissive_color.rgba = nullptr;

        FwType dt{};
        memcpy(dt.acc, p.acceleration, sizeof dt.acc);
        memcpy(dt.angacc, p.angular_acceleration, sizeof dt.angacc);
        dt.lin_drag = p.linear_drag, dt.ang_drag = p.angular_drag;
        dt.sc_kind = T.scale.kind, dt.sc_n = T.scale.n;
        dt.bc_kind = T.base.kind, dt.bc_n = T.base.n;
        dt.em_kind = T.emis.kind, dt.em_n = T.emis.n;
        dt.pbr = p.pbr, dt.report_destroyed = p.report_destroyed;
        // (angular_drag must be finite: 0 * inf = NaN, core.rs:648-650 would turn a zero angular velocity into NaN)
        bool nospin = ctx->use_nospin && p.angular_acceleration[0] == 0.f && p.angular_acceleration[1] == 0.f &&
                      p.angular_acceleration[2] == 0.f && std::isfinite(p.angular_drag);
        {  // FW_TYPE_NOSPIN: every entry that feeds the type spawns with zero angular velocity and the same rotation
            const fw_emission_settings *first = nullptr;
            for (uint32_t i = 0; i < ne && nospin; i++) {
                const fw_emission_settings &e = d->emission_settings[i];
                if ((uint32_t)e.particle_index != t) continue;
                nospin = e.initial_angular_velocity.magnitude.min == 0.f && e.initial_angular_velocity.magnitude.max == 0.f &&
                         (!first || memcmp(first->initial_rotation, e.initial_rotation, sizeof e.initial_rotation) == 0);
                if (!first) first = &e;
            }
            nospin = nospin && first != nullptr;
            if (nospin) {
                dt.flags |= FW_TYPE_NOSPIN;
                // (+ 0.0f: a negative zero component becomes +0, which is what from_scaled_axis(0) * rotation makes of it in
                // all but contrived cases; every reader then sees the same bits)
                for (int c = 0; c < 4; c++) dt.const_rot[c] = first->initial_rotation[c] + 0.0f;
            }
        }
        FwTypeColl dc{};
        dc.coll_flags = (p.collision.enabled ? FW_COLL_ENABLED : 0u) |
                        (p.collision.enabled && p.collision.destroy_on_collision ? FW_COLL_DESTROY : 0u);
        dc.coll_mask = p.collision.filter_mask;
        dc.coll_restitution = p.collision.restitution, dc.coll_friction = p.collision.friction;
        std::vector<float> keys;
        auto put = [&](const std::vector<float> &v, uint32_t padded) {
            uint32_t off = (uint32_t)keys.size();
            keys.insert(keys.end(), v.begin(), v.end());
            keys.resize(off + padded, 0.f);
            return off;
        };
        put(T.scale.times, pad4(T.scale.n));
        dt.o_sc_v = put(T.scale.values, pad4(T.scale.n));
        dt.o_bc_t = put(T.base.times, pad4(T.base.n));
        dt.o_bc_v = put(T.base.values, 4 * T.base.n);
        dt.o_em_t = put(T.emis.times, pad4(T.emis.n));
        dt.o_em_v = put(T.emis.values, 4 * T.emis.n);
        if (keys.size() > 0x3FFFFFFFu) return fail(ctx, FW_EINVAL, "curve keys exceed 2^30 floats");
        const bool bigkeys = keys.size() > FW_KEYS_MAX;  // beyond the LDS staging area of the streaming kernels
        uint32_t type_idx;
        if (!ctx->free_types.empty()) {
            type_idx = ctx->free_types.back();
            ctx->free_types.pop_back();
        } else {
            type_idx = ctx->n_types++;
        }
        // a window of the key pool: the first released one that is large enough, or fresh floats at the end
        uint32_t kwin_off = 0, kwin_cap = 0;
        for (size_t fi = 0; fi < ctx->free_keys.size(); fi++)
            if (ctx->free_keys[fi].second >= keys.size()) {
                kwin_off = ctx->free_keys[fi].first, kwin_cap = ctx->free_keys[fi].second;
                ctx->free_keys.erase(ctx->free_keys.begin() + (long)fi);
                break;
            }
        if (!kwin_cap) {
            kwin_cap = std::max<uint32_t>(64u, pad4((uint32_t)keys.size()));
            if ((st = dev_reserve(ctx, ctx->d_keys, ctx->keys_end + kwin_cap, ctx->keys_end))) return st;
            ctx->g.keys = ctx->d_keys.d;
            kwin_off = (uint32_t)ctx->keys_end;
            ctx->keys_end += kwin_cap;
        }
        dt.keys_off = kwin_off;
        dt.keys_len = (uint32_t)keys.size();
        // segment: the slot, its type index and the spawner's reference to it are recorded BEFORE anything that can
        // fail, so that release_spawner_segments undoes a build that stops half-way (nothing leaks, nothing dangles)
        uint32_t si = (uint32_t)ctx->segs.size();
        for (uint32_t k = 0; k < ctx->segs.size(); k++)
            if (!ctx->segs[k].in_use) {
                si = k;
                break;
            }
        if (si == ctx->segs.size()) ctx->segs.push_back(SegHost{});
        SegHost &S = ctx->segs[si];
        S = SegHost{};
        S.in_use = true, S.spawner = h, S.type = (int)t, S.type_idx = type_idx;
        ctx->n_in_use++, ctx->big_dirty = true;
        S.keys_off = dt.keys_off, S.keys_len = dt.keys_len, S.keys_cap = kwin_cap, S.bigkeys = bigkeys;
        S.nospin = nospin, S.n_xplanes = nospin ? 1u : 0u;
        // scale and colours are left to the readers from the first frame on (fw_ctx::derive_all; wants_derived: not for a type that
        // runs on the collision kernels)
        S.derived = ctx->use_derived && ctx->derive_all && !(p.collision.enabled != 0 || bigkeys);
        if (S.derived) dt.flags |= FW_TYPE_DERIVED;
        memcpy(S.const_rot, dt.const_rot, sizeof S.const_rot);
        sp.seg[t] = si;
        FW_HIP(ctx, hipMemcpy(ctx->d_keys.d + dt.keys_off, keys.data(), keys.size() * sizeof(float),
                              hipMemcpyHostToDevice));
        FW_HIP(ctx, hipMemcpy(ctx->d_types.d + type_idx, &dt, sizeof dt, hipMemcpyHostToDevice));
        FW_HIP(ctx, hipMemcpy(ctx->d_type_coll.d + type_idx, &dc, sizeof dc, hipMemcpyHostToDevice));
        S.lplane_emission.clear();
        for (uint32_t i = 0; i < ne; i++) {
            const fw_emission_settings &e = d->emission_settings[i];
            if (e.mode == FW_MODE_NESTED && (uint32_t)e.target_particle_type == t)
                S.lplane_emission.push_back((int32_t)i), S.n_lplanes++;
            if (e.mode == FW_MODE_NESTED && (uint32_t)e.particle_index == t) S.nested_fed = true;
        }
        S.auto_capacity = p.capacity == 0;
        S.collides = p.collision.enabled != 0 || bigkeys;
        S.coll_inplace = p.collision.enabled != 0 && p.collision.destroy_on_collision == 0 && !bigkeys;
        S.life_bound = (double)std::max(p.lifetime.min, p.lifetime.max);  // lifetime = lerp(min, max, u), u in [0, 1)
        S.win_ok = !S.nested_fed && std::isfinite(S.life_bound);
        {  // FIFO ring (SegHost::fifo): one lifetime value, no collisions; spawners whose particles emit onto their own
            // type stay on the general path (a parent would see this frame's children as parents)
            // ... and so does a type that receives Nested children AND Global particles (its Global particles would have to be
            // placed behind a live count only the device knows)
            bool any_nested = false, self_nested = false, mixed_feed = false;
            uint32_t n_global_feed = 0;  // Global entries that feed the type: each may add one op to a frame
            for (uint32_t i = 0; i < ne; i++) {
                const fw_emission_settings &e = d->emission_settings[i];
                n_global_feed += (e.mode == FW_MODE_GLOBAL && (uint32_t)e.particle_index == t) ? 1u : 0u;
                any_nested |= e.mode == FW_MODE_NESTED;
                self_nested |= e.mode == FW_MODE_NESTED && e.target_particle_type == e.particle_index;
                mixed_feed |= S.nested_fed && e.mode == FW_MODE_GLOBAL && (uint32_t)e.particle_index == t;
            }
            // (a ring's spawn ops of a frame travel in the kernel arguments of its launch -- FwInlineOps, FW_INLINE_OPS of
            // them: a type fed by more Global entries than that takes the range or the compacting path, whose tiles read op
            // tables from memory)
            S.fifo = ctx->use_fifo && !sp.no_rings && !self_nested && !mixed_feed && (!S.collides || S.coll_inplace) && p.lifetime.min == p.lifetime.max &&
                     n_global_feed <= FW_INLINE_OPS &&
                     std::isfinite(p.lifetime.min) &&
                     caps[t] >= ctx->fifo_min && caps[t] < 0x40000000u &&  // (head + index stays far from 2^32)
                     (!any_nested || ctx->fifo_nested);
            // more such types than one FIFO launch holds (fw_ctx::n_spilled): this one takes a range ring -- if it qualifies for
            // one: the rule below -- and the FIFO rings of the context follow it at the end of the build
            const bool spill = S.fifo && (ctx->n_spilled != 0 || ctx->n_fifo >= kMaxFifoSegs);
            if (spill) S.fifo = false;
            // (fw_ctx::range_few) the capacity of a type that receives Nested children is derived from its parents' CAPACITY -- the
            // host cannot bound their number -- and passes fifo_min for a handful of parents already (examples/textures.rs: 55
            // bullet cases, 110 puffs, 32 768 slots): in a context of few segments such a type stays with its small parent type on
            // range rings (one kind of launch per frame) unless its derived capacity is really large
            bool few_nested = false;  // ... a range ring only because of that: it leaves with the other small rings (drop_few_rings)
            if (S.fifo && S.nested_fed && ctx->use_range && ctx->range_few != 0 && !ctx->few_blocked && ctx->n_in_use <= ctx->range_few && ctx->n_fifo == 0 &&
                caps[t] < 8u * ctx->fifo_min && caps[t] < ctx->range_min * 32u &&
                S.n_lplanes <= 2 && T.life_lo_safe > 0.0f && caps[t] <= FW_RANGE_MAX_CAPACITY)  // (it does qualify for a range ring)
                S.fifo = false, few_nested = true;
            if (S.fifo) {
                ctx->n_fifo++;
                S.win_ok = false;
                S.fifo_mat = any_nested;
                S.fifo_dev = S.nested_fed;
                if (S.fifo_dev) {
                    FW_HIP(ctx, hipHostMalloc((void **)&S.h_report, (size_t)kReportRing * sizeof(unsigned long long), hipHostMallocDefault));
                    memset(S.h_report, 0, (size_t)kReportRing * sizeof(unsigned long long));
                }
                S.fifo_life = 0.0f * (p.lifetime.max - p.lifetime.min) + p.lifetime.min;  // u * (max - min) + min, any u
                S.fifo_wm = (T.base.kind != 0 ? 1 : 0) | (T.emis.kind != 0 ? 2 : 0) | (T.scale.kind != 0 ? 4 : 0);
            }
            // Range ring (SegHost::range): any finite lifetime range -- a single value included, for the types the eight
            // FIFO records of a launch have no room for -- in a spawner without Nested entries; the young part of the
            // list is updated in place, only the part that can lose particles this frame is compacted
            // In a spawner WITH Nested entries (round 4): types other particles' entries emit FROM (range_mat: fw_k_spawn /
            // fw_k_nest address their particles by list index through the size of the old part, which the device keeps --
            // FwGlobals::rold) and types that RECEIVE children (range_dev: the device alone knows their count) qualify too;
            // as for FIFO rings, not a type that emits onto itself, nor one that receives children AND Global particles; at
            // most two last_emitted_age planes (the old tiles carry them in registers).
            S.range = ctx->use_range && !sp.no_rings && !S.fifo && !self_nested && !mixed_feed && S.n_lplanes <= 2 &&
                      (!S.collides || S.coll_inplace) && std::isfinite(p.lifetime.min) &&
                      std::isfinite(p.lifetime.max) && T.life_lo_safe > 0.0f &&
                      (caps[t] >= ctx->range_min || (ctx->range_few != 0 && !ctx->few_blocked && ctx->n_in_use <= ctx->range_few && ctx->n_fifo == 0)) &&
                      caps[t] <= FW_RANGE_MAX_CAPACITY;
            if (S.range) {
                if (caps[t] < ctx->range_min || few_nested) S.few_ring = true, ctx->n_few++;  // (fw_ctx::range_few)
                if (spill) S.spilled = true, ctx->n_spilled++;
                ctx->n_range++;
                S.range_life_lo = T.life_lo_safe;
                ctx->range_life_max = std::max(ctx->range_life_max, S.range_life_lo);
                ctx->r_force = true;
                S.range_mat = S.n_lplanes != 0;
                S.range_dev = S.nested_fed;
                if (S.range_dev) {
                    S.win_ok = false;
                    ctx->range_age_keep = std::max(ctx->range_age_keep, (float)(S.life_bound * 1.01 + 1e-3));
                    FW_HIP(ctx, hipHostMalloc((void **)&S.h_report, (size_t)kReportRing * sizeof(unsigned long long), hipHostMallocDefault));
                    memset(S.h_report, 0, (size_t)kReportRing * sizeof(unsigned long long));
                }
            }
        }
        for (int c = 0; c < 4; c++) {  // the first key is the colour at age 0 (and, for one key, at every age)
            S.fill_bc[c] = T.base.values.empty() ? 0.f : T.base.values[c];
            S.fill_em[c] = T.emis.values.empty() ? 0.f : T.emis.values[c];
        }
        S.colors_dirty = false;
        {
            double expect = 0.0;
            derive_capacity(d, t, caps, &expect);
            S.expect_live = (float)std::min(expect, 3.0e9);
            uint32_t feeders = 0, global_feeders = 0;
            for (uint32_t i = 0; i < d->n_emission_settings; i++)
                if ((uint32_t)d->emission_settings[i].particle_index == t) feeders++, global_feeders += d->emission_settings[i].mode == FW_MODE_GLOBAL ? 1u : 0u;
            S.one_feeder = feeders == 1 && global_feeders == 1;
        }
        if ((st = alloc_seg_buffers(ctx, S, caps[t], p.report_destroyed != 0))) return st;
        if (small_eligible(ctx, S)) enter_small(ctx, S);  // (fw_ctx::n_small: the wave-per-type kernel)
        if ((st = upload_seg(ctx, si))) return st;
        const uint32_t zero2[2] = {0, 0};
        for (int r = 0; r < 2; r++) {
            FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
            FW_HIP(ctx, hipMemcpy(ctx->g.spawned + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
            FW_HIP(ctx, hipMemcpy(ctx->g.appended + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
            FW_HIP(ctx, hipMemcpy(ctx->g.rold + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
        }
        FW_HIP(ctx, hipMemcpy(ctx->g.ndestroyed + si, zero2, 4, hipMemcpyHostToDevice));
        FW_HIP(ctx, hipMemcpy(ctx->g.range_ticket + si, zero2, 4, hipMemcpyHostToDevice));  // (S.ticket_base is 0: a fresh SegHost)
    }

    for (uint32_t i = 0; i < ne; i++) {
        EmissionHost &E = sp.em[i];
        const fw_emission_settings &e = d->emission_settings[i];
        E.es = e;
        E.between = (e.offset_end - e.offset_start) / e.count;
        E.last_emission = 0.f, E.time_passed_in_cycle = 0.f;  // sync_spawner_data core.rs:350-358
        E.enabled = d->starts_enabled != 0;
        E.emits_on_other_particles = e.mode == FW_MODE_NESTED;
        E.dst_seg = sp.seg[e.particle_index];
        E.life_lo_safe = sp.types[e.particle_index].life_lo_safe;
        E.serial = carry_serial && i < carry_serial->size() ? (*carry_serial)[i] : 0;
        const fw_particle_settings &p = d->particle_settings[e.particle_index];
        FwEmit de{};
        de.shape_kind = e.shape_kind, de.shape_radius = e.shape_radius;
        de.uid = d->uid, de.emission_index = i;
        fw_q4 sa = fw_quat_from_rotation_arc(fw_v3{0.f, 1.f, 0.f},
                                             fw_v3{e.shape_normal[0], e.shape_normal[1], e.shape_normal[2]});
        de.shape_arc[0] = sa.x, de.shape_arc[1] = sa.y, de.shape_arc[2] = sa.z, de.shape_arc[3] = sa.w;
        fill_randvec3(e.initial_velocity, de.v_mag_min, de.v_mag_max, de.v_spread, de.v_dir, de.v_arc);
        fill_randvec3(e.initial_angular_velocity, de.w_mag_min, de.w_mag_max, de.w_spread, de.w_dir, de.w_arc);
        de.inherit = e.inherit_parent_velocity;
        de.type_idx = ctx->segs[sp.seg[e.particle_index]].type_idx;
        memcpy(de.init_rot, e.initial_rotation, sizeof de.init_rot);
        de.radial_min = e.initial_velocity_radial.min, de.radial_max = e.initial_velocity_radial.max;
        de.iscale_min = p.initial_scale.min, de.iscale_max = p.initial_scale.max;
        de.life_min = p.lifetime.min, de.life_max = p.lifetime.max;
        de.n_count = e.count, de.n_start = e.offset_start, de.n_end = e.offset_end;
        de.n_lplane = 0;
        if (e.mode == FW_MODE_NESTED) {
            const SegHost &P = ctx->segs[sp.seg[e.target_particle_type]];
            for (uint32_t k = 0; k < P.n_lplanes; k++)
                if (P.lplane_emission[k] == (int32_t)i) de.n_lplane = k;
        }
        if (!ctx->free_emits.empty()) {
            E.emit_idx = ctx->free_emits.back();
            ctx->free_emits.pop_back();
        } else {
            E.emit_idx = ctx->n_emits++;
        }
        if (!ctx->free_emit_slots.empty()) {
            E.emit_slot = ctx->free_emit_slots.back();
            ctx->free_emit_slots.pop_back();
        } else {
            E.emit_slot = ctx->n_emit_slots++;
        }
        E.assigned = true;
        FW_HIP(ctx, hipMemcpy(ctx->d_emits.d + E.emit_idx, &de, sizeof de, hipMemcpyHostToDevice));
        const unsigned long long s0 = E.serial;
        FW_HIP(ctx, hipMemcpy(ctx->d_emit_serial.d + E.emit_slot, &s0, sizeof s0, hipMemcpyHostToDevice));
        const uint32_t t0 = 0u;
        FW_HIP(ctx, hipMemcpy(ctx->d_nest_start.d + E.emit_slot, &t0, sizeof t0, hipMemcpyHostToDevice));
        ctx->nest_ticket_base[E.emit_slot] = 0u;
    }
    // ring types other particles' entries emit from that need no materialisation (SegHost::virt_parent)
    for (uint32_t t = 0; t < nt; t++) {
        SegHost &S = ctx->segs[sp.seg[t]];
        if (!S.ring() || S.nested_fed || S.n_lplanes > 2) continue;
        if (S.n_lplanes == 0) {  // (no entry emits from it or onto it: nothing in a Nested pass ever looks at its particles)
            S.virt_parent = true;
            continue;
        }
        const fw_particle_settings &p = d->particle_settings[t];
        bool ok = std::min(p.lifetime.min, p.lifetime.max) > 0.0f;
        for (uint32_t k = 0; k < S.n_lplanes && ok; k++) {
            const fw_emission_settings &e = d->emission_settings[S.lplane_emission[k]];
            ok = e.pacing_kind == FW_PACING_COUNT_OVER_DURATION && e.count > 0.0f && e.offset_start >= 0.0f && e.offset_end >= e.offset_start &&
                 std::isfinite(e.count) && std::isfinite(e.offset_end);
        }
        S.virt_parent = ok;
        if (ok && (st = upload_seg(ctx, sp.seg[t]))) return st;
    }
    sp.initialized = true;
    if ((st = ensure_range_arrays(ctx))) return st;
    if ((st = ensure_tile_arrays(ctx))) return st;
    if (ctx->segs.size() > ctx->small_cap) {  // the small-type list (fw_ctx::d_small): room for every segment slot; fw_step never allocates
        if ((st = sync(ctx))) return st;
        const size_t ncap = ctx->segs.size() * 2 + 256;
        if (ctx->d_small) hipFree(ctx->d_small);
        if (ctx->h_small) hipHostFree(ctx->h_small);
        ctx->d_small = nullptr, ctx->h_small = nullptr, ctx->small_cap = 0, ctx->small_pending = false;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_small, ncap * sizeof(uint32_t)));
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_small, ncap * sizeof(uint32_t), hipHostMallocDefault));
        ctx->small_cap = ncap, ctx->small_dirty = true;
    }
    // the context is no longer one of few segments without a FIFO ring: its small range rings continue on the compacting path
    // (fw_ctx::range_few; callers of build_spawner have synchronised the context)
    if (ctx->n_spilled && ctx->n_fifo && (st = spill_fifo_rings(ctx))) return st;  // (fw_ctx::n_spilled)
    if (ctx->n_in_use > ctx->range_few) ctx->few_blocked = true;
    if (ctx->n_few && (ctx->n_fifo != 0 || ctx->n_in_use > ctx->range_few)) {
        // (the hysteresis covers the arrival of a FIFO ring as well: a context in which one comes and goes would otherwise
        // convert its small rings at every arrival -- ADVICE r04)
        ctx->few_blocked = true;
        if ((st = drop_few_rings(ctx))) return st;
    }
    update_small_mode(ctx);
    return FW_OK;
}

fw_status release_spawner_segments(fw_ctx *ctx, SpawnerHost &sp) {
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    ctx->tab_force = true;
    for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;  // rows in flight describe the old segments
    for (const EmissionHost &e : sp.em) {
        if (!e.assigned) continue;  // a build that failed half-way
        ctx->free_emits.push_back(e.emit_idx);
        ctx->free_emit_slots.push_back(e.emit_slot);
    }
    sp.em.clear();
    for (uint32_t si : sp.seg) {
        if (si == kNoSeg) continue;
        SegHost &S = ctx->segs[si];
        if (!S.in_use) continue;
        ctx->free_types.push_back(S.type_idx);
        if (S.keys_cap) ctx->free_keys.push_back({S.keys_off, S.keys_cap});
        if (S.fifo) ctx->n_fifo--;
        if (S.range) ctx->n_range--, ctx->r_force = true;
        if (S.few_ring) ctx->n_few--;
        if (S.spilled) ctx->n_spilled--;
        if (S.small) ctx->n_small--, ctx->small_dirty = true, ctx->n_small_coll -= S.collides ? 1u : 0u;
        if (S.small_ok) ctx->n_small_ok--;
        if (S.inst) ctx->n_inst--;
        if (S.solo) ctx->n_solo--;
        ctx->n_in_use--, ctx->big_dirty = true;
        if (ctx->n_in_use <= ctx->range_few / 2) ctx->few_blocked = false;
        if (S.h_report) hipHostFree(S.h_report);
        if (S.buf[0]) FW_HIP(ctx, hipFree(S.buf[0]));
        if (S.destroyed) FW_HIP(ctx, hipFree(S.destroyed));
        S = SegHost{};
        const uint32_t zero = 0;
        for (int r = 0; r < 2; r++)
            FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)r * ctx->max_seg + si, &zero, 4, hipMemcpyHostToDevice));
    }
    sp.seg.clear();
    update_small_mode(ctx);
    return FW_OK;
}


}  // namespace fwh
