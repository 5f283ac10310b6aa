// fw_engine_api.cpp -- the C ABI of include/firework_hip.h other than fw_step (+ the debug hooks of firework_hip_debug.h)
// (host engine of libfirework_hip.so: fw_engine.h lists its translation units; there is no CPU simulation path in this library)
#include "fw_engine.h"

thread_local std::string fwh::g_create_error;  // (fw_last_error(nullptr): per calling thread, like errno)

// Is `p` inside a mapping of this process that the host may write?  (The runtime reserves the device's address range in the host's
// address space whether or not the memory behind an address is reachable through the BAR: a reserved-only page is mapped PROT_NONE,
// and a store to it would be a SIGSEGV, not an error code -- so the page's permissions are read from /proc/self/maps, once, when a
// context is created.)
static bool host_maps_writable(const void *p) {
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return false;
    char line[512];
    bool ok = false;
    const unsigned long long a = (unsigned long long)(uintptr_t)p;
    while (fgets(line, sizeof line, f)) {
        unsigned long long lo = 0, hi = 0;
        char perms[8] = {0};
        if (sscanf(line, "%llx-%llx %7s", &lo, &hi, perms) == 3 && a >= lo && a < hi) {
            ok = perms[0] == 'r' && perms[1] == 'w';
            break;
        }
    }
    fclose(f);
    return ok;
}

extern "C" {


int fw_abi_version(void) { return FW_ABI_VERSION; }

const char *fw_last_error(const fw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

uint64_t fw_compute_emission_count(float t, float last, float dur, float start, float end, float per_cycle,
                                   float *next_last) {
    float nl = 0.f;
    const uint64_t n = fw_emission_count(t, last, dur, start, end, per_cycle, &nl);
    if (next_last) *next_last = nl;
    return n;
}

fw_status fw_ctx_create(int device, uint32_t seed, void *stream, fw_ctx **out) {
    if (!out) return FW_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) +
                         "); this backend has no CPU fallback";
        return FW_ENODEV;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return FW_EINVAL;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return FW_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName;
        return FW_ENODEV;
    }
    fw_ctx *ctx = new fw_ctx();
    ctx->device = device;
    ctx->seed = seed;
    {  // fw_ctx::param_bar: device memory the host can write (large BAR) -- asked of the runtime, then tried once
        int large_bar = 0;
        void *probe = nullptr;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large_bar != 0 &&
            hipExtMallocWithFlags(&probe, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
            hipPointerAttribute_t at{};
            ctx->param_bar = hipPointerGetAttributes(&at, probe) == hipSuccess && at.type == hipMemoryTypeDevice && host_maps_writable(probe);
            hipFree(probe);
        }
        (void)hipGetLastError();
    }
    auto bail = [&](const char *what, hipError_t he) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(he);
        delete ctx;
        return FW_EHIP;
    };
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e);
        ctx->own_stream = true;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate(copy)", e);
    if ((e = hipStreamCreateWithFlags(&ctx->fifo_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate(rings)", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_rtab, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_small, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming)) != hipSuccess)
        return bail("hipEventCreate", e);
    for (int i = 0; i < kParamRing; i++) {
        if ((e = hipEventCreateWithFlags(&ctx->ev_copied[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&ctx->ev_consumed[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    }
    for (int i = 0; i < kTabRing; i++)
        if ((e = hipEventCreateWithFlags(&ctx->ev_tab[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    if ((e = hipHostMalloc((void **)&ctx->h_done, 64, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    *ctx->h_done = 0ull;
    ctx->h_err = ctx->h_done + 4, ctx->h_err[0] = ctx->h_err[1] = 0ull, ctx->g.err_host = ctx->h_err;
    if ((e = hipMalloc((void **)&ctx->g.err, 64)) != hipSuccess) return bail("hipMalloc", e);
    fw_memset_done(ctx->g.err, 0, 64);
    if ((e = hipMalloc((void **)&ctx->g.stats, FW_STAT_SLOTS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    fw_memset_done(ctx->g.stats, 0, FW_STAT_SLOTS * sizeof(unsigned long long));
    if ((e = hipMalloc((void **)&ctx->d_aabb, 256 * 8 * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_total, 64)) != hipSuccess) return bail("hipMalloc", e);
    ctx->g.seed = seed;
    // A/B and debugging switches (firework_hip_debug.h): read only when FW_ENABLE_KNOBS=1 -- a product process does not change
    // behaviour because of a stray environment variable
    const char *knobs_on = getenv("FW_ENABLE_KNOBS");
    auto getenv = [&](const char *name) -> const char * { return (knobs_on && atoi(knobs_on) != 0) ? ::getenv(name) : nullptr; };
    // -- path selectors: every one of them names a path the product takes by itself under some workload; the tests force each
    if (const char *m = getenv("FW_UPDATE_MODE")) ctx->update_mode = !strcmp(m, "split") ? FW_MODE_SPLIT : FW_MODE_FUSED;
    if (const char *m = getenv("FW_FORECAST")) ctx->use_forecast = atoi(m) != 0;
    if (const char *m = getenv("FW_TF")) ctx->use_tf = atoi(m) != 0;
    if (const char *m = getenv("FW_TF_MIN_TILES")) ctx->tf_min_tiles = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_STREAM")) ctx->use_stream = atoi(m) != 0;
    if (const char *m = getenv("FW_STATIC_NEW")) ctx->use_static_new = atoi(m) != 0;  // 0: always count + look back
    if (const char *m = getenv("FW_SPIN_LIMIT")) ctx->spin_limit = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_FIFO")) ctx->use_fifo = atoi(m) != 0;
    if (const char *m = getenv("FW_FIFO_MIN")) ctx->fifo_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_FIFO_SMALL")) ctx->fifo_small_tiles = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_FIFO_STREAM")) ctx->use_fifo_stream = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE")) ctx->use_range = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE_MIN")) ctx->range_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_RANGE_FEW")) ctx->range_few = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_RANGE_SMALL")) ctx->range_small_tiles = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_RANGE_YOUNG_BIG")) ctx->range_young_big = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_NOSPIN")) ctx->use_nospin = atoi(m) != 0;
    // 0: scale / colour planes always stored; 1: not stored for types with an attached instance buffer; 2 (default): for no type
    if (const char *m = getenv("FW_DERIVED")) ctx->use_derived = atoi(m) != 0, ctx->derive_all = atoi(m) >= 2;
    if (const char *m = getenv("FW_NEST_FUSE")) ctx->nest_fuse = atoi(m) != 0;
    if (const char *m = getenv("FW_SMALL")) ctx->use_small = atoi(m) != 0;
    if (const char *m = getenv("FW_SMALL_MAX")) ctx->small_max = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_SMALL_MIN")) ctx->small_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_WIDE_MAX")) ctx->wide_max = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_WIDE_MIN")) ctx->wide_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_WIDE_MID")) ctx->wide_mid = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_WAVE_ALL_MIN")) ctx->wave_all_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_HOST_FAST")) ctx->host_fast = atoi(m) != 0;
    if (const char *m = getenv("FW_PARAM_BAR")) ctx->param_bar = ctx->param_bar && atoi(m) != 0;
    for (int i = 0; i < kParamRing && ctx->param_bar; i++)
        if ((e = hipExtMallocWithFlags((void **)&ctx->b_param[i], kBarParamBytes, hipDeviceMallocFinegrained)) != hipSuccess)
            return bail("hipExtMallocWithFlags(op tables)", e);
    if (const char *m = getenv("FW_NT_MB")) ctx->nt_bytes = (uint64_t)atoll(m) << 20;
    if (const char *m = getenv("FW_NT_WO_MB")) ctx->nt_wo_bytes = ctx->nt_wo_bytes_range = (uint64_t)atoll(m) << 20;
#ifdef FW_AB
    // -- the experiment surface: only in the `make ab` build (libfirework_hip_ab.so; the tools load it through FW_LIB_PATH)
    if (const char *m = getenv("FW_DEBUG")) ctx->dbg = (uint32_t)atoi(m);
    if (const char *m = getenv("FW_FIFO_NESTED")) ctx->fifo_nested = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE_IDLE_LAST")) ctx->range_idle_last = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE_SPREAD_NEW")) ctx->range_spread_new = atoi(m) != 0;
    if (const char *m = getenv("FW_AABB")) ctx->track_aabb = atoi(m) != 0;  // same as fw_ctx_track_aabbs(ctx, 1)
    if (const char *m = getenv("FW_OPS_ZEROCOPY")) ctx->ops_zerocopy = atoi(m) != 0;
    if (const char *m = getenv("FW_SNAP_EVERY")) ctx->snap_every = std::max(1, atoi(m));
    ctx->trace = getenv("FW_TRACE") != nullptr;
    if (const char *m = getenv("FW_HOST_PROF")) ctx->host_prof = atoi(m) != 0, ctx->host_prof_skip = atoi(m) > 1 ? (uint64_t)atoi(m) : 0;
#endif
    if (ensure_max_seg(ctx, 1024) != FW_OK) {
        g_create_error = ctx->err;
        delete ctx;
        return FW_EHIP;
    }
    *out = ctx;
    return FW_OK;
}

fw_status fw_ctx_destroy(fw_ctx *ctx) {
    if (!ctx) return FW_EINVAL;
    if (ctx->host_prof && ctx->prof_frames) {
        static const char *names[10] = {"windows+reset", "spawner loop", "tile table", "commit", "args", "op tables", "launch", "post", "", ""};
        fprintf(stderr, "[fw] host half of fw_step over %llu frames (ns per frame):", (unsigned long long)ctx->prof_frames);
        for (int i = 0; i < 8; i++) fprintf(stderr, "  %s %.0f", names[i], ctx->prof_ns[i] / (double)ctx->prof_frames);
        fprintf(stderr, "\n[fw] table uploads: general %llu, range %llu over %llu frames\n", (unsigned long long)ctx->tab_seq,
                (unsigned long long)ctx->r_uploads, (unsigned long long)ctx->frame);
        fprintf(stderr, "[fw] host records: SegHost %zu B, SpawnerHost %zu B (entry at %zu), EmissionHost %zu B, FwOp %zu B\n", sizeof(SegHost),
                sizeof(SpawnerHost), (size_t)((const char *)&ctx->spawners.data()->em.one_ - (const char *)ctx->spawners.data()),
                sizeof(EmissionHost), sizeof(FwOp));
    }
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->copy_stream);
    if (ctx->fifo_stream) hipStreamSynchronize(ctx->fifo_stream);
    for (auto &S : ctx->segs) {
        if (S.buf[0]) hipFree(S.buf[0]);
        if (S.destroyed) hipFree(S.destroyed);
        if (S.h_report) hipHostFree(S.h_report);
    }
    void *frees[] = {ctx->d_type_coll.d, ctx->d_segs.d,       ctx->d_types.d,       ctx->d_keys.d,        ctx->d_emits.d,
                     ctx->d_emit_serial.d, ctx->d_nest_start.d, ctx->g.range_ticket, ctx->g.count,        ctx->g.spawned,       ctx->g.appended,     ctx->g.rold,
                     ctx->g.ndestroyed,   ctx->g.tile_cnt,      ctx->g.tile_off,      ctx->g.tile_status,
                     ctx->g.err,          ctx->g.stats,         ctx->g.nest_status,   ctx->g.nest_ticket,
                     ctx->d_aabb,         ctx->d_total,         ctx->d_segids,        ctx->g.dbg_ts,
                     ctx->d_colliders,   ctx->g.tile_box,      ctx->d_stage};
    for (void *p : frees)
        if (p) hipFree(p);
    for (int i = 0; i < kParamRing; i++) {
        if (ctx->h_param[i]) hipHostFree(ctx->h_param[i]);
        if (ctx->d_param[i]) hipFree(ctx->d_param[i]);
        if (ctx->b_param[i]) hipFree(ctx->b_param[i]);
        hipEventDestroy(ctx->ev_copied[i]);
        hipEventDestroy(ctx->ev_consumed[i]);
    }
    for (int i = 0; i < kTabRing; i++) {
        hipEventDestroy(ctx->ev_tab[i]);
        if (ctx->h_tab[i]) hipHostFree(ctx->h_tab[i]);
        if (ctx->h_desc[i]) hipHostFree(ctx->h_desc[i]);
    }
    if (ctx->d_tile_first) hipFree(ctx->d_tile_first);
    if (ctx->d_tile_desc) hipFree(ctx->d_tile_desc);
    if (ctx->d_tile_keys) hipFree(ctx->d_tile_keys);
    for (int i = 0; i < kTabRing; i++)
        if (ctx->h_keys[i]) hipHostFree(ctx->h_keys[i]);
    if (ctx->d_fc) hipFree(ctx->d_fc);
    if (ctx->d_fce) hipFree(ctx->d_fce);
    if (ctx->d_fct) hipFree(ctx->d_fct);
    if (ctx->d_fcl) hipFree(ctx->d_fcl);
    if (ctx->h_snap) hipHostFree(ctx->h_snap);
    if (ctx->h_aabb) hipHostFree(ctx->h_aabb);
    if (ctx->h_done) hipHostFree(ctx->h_done);
    for (hipEvent_t ev : ctx->tev) hipEventDestroy(ev);
    hipStreamDestroy(ctx->copy_stream);
    if (ctx->fifo_stream) hipStreamDestroy(ctx->fifo_stream);
    if (ctx->ev_side) hipEventDestroy(ctx->ev_side);
    if (ctx->ev_rtab) hipEventDestroy(ctx->ev_rtab);
    if (ctx->ev_small) hipEventDestroy(ctx->ev_small);
    if (ctx->d_small) hipFree(ctx->d_small);
    if (ctx->h_small) hipHostFree(ctx->h_small);
    for (int i = 0; i < 2; i++) {
        if (ctx->ev_coll[i]) hipEventDestroy(ctx->ev_coll[i]);
        if (ctx->h_coll[i]) hipHostFree(ctx->h_coll[i]);
    }
    if (ctx->d_rdesc) hipFree(ctx->d_rdesc);
    if (ctx->h_rdesc) hipHostFree(ctx->h_rdesc);
    if (ctx->d_rstatus) hipFree(ctx->d_rstatus);
    if (ctx->d_rts) hipFree(ctx->d_rts);
    for (int i = 0; i < kParamRing; i++) {
        if (ctx->h_rparam[i]) ctx->param_bar ? hipFree(ctx->h_rparam[i]) : hipHostFree(ctx->h_rparam[i]);
    }
    if (ctx->ev_main) hipEventDestroy(ctx->ev_main);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return FW_OK;
}

void *fw_ctx_stream(const fw_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

fw_status fw_ctx_synchronize(fw_ctx *ctx) {
    if (!ctx) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    return check_device_errors(ctx);
}

fw_status fw_ctx_set_colliders(fw_ctx *ctx, const fw_collider *colliders, uint32_t n) {
    if (!ctx || (n && !colliders)) return fail(ctx, FW_EINVAL, "bad collider set");
    hipSetDevice(ctx->device);
    for (uint32_t i = 0; i < n; i++)
        if (colliders[i].kind < FW_COLLIDER_PLANE || colliders[i].kind > FW_COLLIDER_CONE)
            return fail(ctx, FW_EINVAL, "unknown collider kind");
    // The reference asks the live physics world every frame (core.rs:756-765): a set that changes every frame must not
    // stall the frames in flight.  The new set is staged in pinned memory and copied by the stream itself.
    const int slot = (int)(ctx->coll_seq++ & 1u);
    if (ctx->coll_pending[slot]) {  // the copy of two calls ago: long done unless the caller replaces the set in a tight loop
        FW_HIP(ctx, hipEventSynchronize(ctx->ev_coll[slot]));
        ctx->coll_pending[slot] = false;
    }
    if (!ctx->ev_coll[slot]) FW_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_coll[slot], hipEventDisableTiming));
    if (n > ctx->h_coll_cap[slot]) {
        const size_t ncap = std::max<size_t>(64, (size_t)n * 2);
        if (ctx->h_coll[slot]) FW_HIP(ctx, hipHostFree(ctx->h_coll[slot]));
        ctx->h_coll[slot] = nullptr, ctx->h_coll_cap[slot] = 0;
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_coll[slot], ncap * sizeof(FwCollider), hipHostMallocDefault));
        ctx->h_coll_cap[slot] = ncap;
    }
    if (n > ctx->coll_cap) {  // a larger world than ever before: the one case that waits (kernels in flight read the old table)
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = std::max<size_t>(64, (size_t)n * 2);
        if (ctx->d_colliders) FW_HIP(ctx, hipFree(ctx->d_colliders));
        ctx->d_colliders = nullptr, ctx->coll_cap = 0;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_colliders, ncap * sizeof(FwCollider)));
        ctx->coll_cap = ncap;
    }
    for (uint32_t i = 0; i < n; i++) {
        const fw_collider &c = colliders[i];
        FwCollider &d = ctx->h_coll[slot][i];
        d = FwCollider{};
        d.kind = c.kind, d.layers = c.layers, d.radius = c.radius;
        // (the sphere around `position` that contains it: a wave skips a collider none of its rays can reach, fw_cast_ray)
        d.bound = c.kind == 1 ? c.radius
                  : c.kind == 2 ? std::sqrt(c.half_extents[0] * c.half_extents[0] + c.half_extents[1] * c.half_extents[1] + c.half_extents[2] * c.half_extents[2]) * 1.0001f
                  : (c.kind == 3 || c.kind == 4) ? std::sqrt(c.radius * c.radius + c.half_extents[1] * c.half_extents[1]) * 1.0001f  // (the rim of a cap / of the base)
                                : INFINITY;
        if (!(d.bound >= 0.0f)) d.bound = INFINITY;  // (NaN / negative extents: never skipped)
        memcpy(d.position, c.position, sizeof c.position);
        memcpy(d.rotation, c.rotation, sizeof c.rotation);
        memcpy(d.normal, c.normal, sizeof c.normal);
        memcpy(d.half_extents, c.half_extents, sizeof c.half_extents);
    }
    if (n) {
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_colliders, ctx->h_coll[slot], n * sizeof(FwCollider), hipMemcpyHostToDevice, ctx->stream));
        FW_HIP(ctx, hipEventRecord(ctx->ev_coll[slot], ctx->stream));
        ctx->coll_pending[slot] = true;
    }
    ctx->n_colliders = n;
    ctx->g.colliders = ctx->d_colliders, ctx->g.n_colliders = n;
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    return FW_OK;
}

fw_status fw_spawner_create(fw_ctx *ctx, const fw_spawner_desc *desc, fw_spawner *out) {
    if (!ctx || !out) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = validate_desc(ctx, desc);
    if (st) return st;
    if ((st = sync(ctx))) return st;
    int h = -1;
    for (size_t i = 0; i < ctx->spawners.size(); i++)
        if (!ctx->spawners[i].alive) {
            h = (int)i;
            break;
        }
    if (h < 0) {
        ctx->spawners.push_back(SpawnerHost{});
        h = (int)ctx->spawners.size() - 1;
    }
    ctx->spawners[h] = SpawnerHost{};
    ctx->spawners[h].alive = true;
    st = build_spawner(ctx, h, desc, nullptr);
    if (st) {
        release_spawner_segments(ctx, ctx->spawners[h]);
        ctx->spawners[h].alive = false;
        return st;
    }
    *out = h;
    return FW_OK;
}

fw_status fw_spawner_update_settings(fw_ctx *ctx, fw_spawner h, const fw_spawner_desc *desc) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = validate_desc(ctx, desc);
    if (st) return st;
    if ((st = sync(ctx))) return st;
    if (sp->poisoned) {
        // everything enqueued on top of the invalid state has finished: what those frames reported again is not news, and the
        // segment slots the report names are about to be reused by the rebuilt spawner
        // (... but the word may by now name ANOTHER spawner, healthy so far -- one word, the last writer wins: mark it before the
        // report is dropped)
        poll_device_error(ctx);
        ctx->h_err[0] = ctx->h_err[1] = 0ull;
        const uint32_t zero = 0;
        FW_HIP(ctx, hipMemcpy(ctx->g.err, &zero, sizeof zero, hipMemcpyHostToDevice));
    }
    // RNG streams never replay: carry the serials of surviving emission indices
    std::vector<uint64_t> serials;
    for (const EmissionHost &e : sp->em) {
        unsigned long long s = e.serial;
        if (e.es.mode == FW_MODE_NESTED)
            hipMemcpy(&s, ctx->d_emit_serial.d + e.emit_slot, sizeof s, hipMemcpyDeviceToHost);
        serials.push_back(s);
    }
    if ((st = release_spawner_segments(ctx, *sp))) return st;
    const bool finished_notified = sp->finished_notified;
    SpawnerHost keep = *sp;
    *sp = SpawnerHost{};
    sp->alive = true;
    sp->manual_queued_count = keep.manual_queued_count;
    memcpy(sp->origin_pos, keep.origin_pos, sizeof keep.origin_pos);
    memcpy(sp->origin_rot, keep.origin_rot, sizeof keep.origin_rot);
    memcpy(sp->parent_vel, keep.parent_vel, sizeof keep.parent_vel);
    sp->mod_scale = keep.mod_scale, sp->mod_speed = keep.mod_speed;
    sp->no_rings = keep.no_rings || keep.poisoned;  // (rebuilt after an internal error: compacting path only from now on)
    st = build_spawner(ctx, h, desc, &serials);
    if (st) {
        // the old particle types are gone and the new ones could not be built: the handle dies (every later call on
        // it returns FW_EINVAL) instead of pointing at segments that do not exist
        const std::string why = ctx->err;
        release_spawner_segments(ctx, ctx->spawners[h]);
        ctx->spawners[h] = SpawnerHost{};
        ctx->err = why + " (spawner destroyed: its settings could not be rebuilt)";
        return st;
    }
    ctx->spawners[h].finished_notified = finished_notified;
    ctx->n_poisoned = 0;  // (fw_step looks for invalid spawners only while some exist)
    for (const SpawnerHost &x : ctx->spawners) ctx->n_poisoned += (x.alive && x.poisoned) ? 1u : 0u;
    return st;
}

fw_status fw_spawner_destroy(fw_ctx *ctx, fw_spawner h) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    if (sp->poisoned) {  // (as in fw_spawner_update_settings)
        poll_device_error(ctx);
        ctx->h_err[0] = ctx->h_err[1] = 0ull;
        const uint32_t zero = 0;
        FW_HIP(ctx, hipMemcpy(ctx->g.err, &zero, sizeof zero, hipMemcpyHostToDevice));
    }
    if ((st = release_spawner_segments(ctx, *sp))) return st;
    *sp = SpawnerHost{};
    ctx->n_poisoned = 0;
    for (const SpawnerHost &x : ctx->spawners) ctx->n_poisoned += (x.alive && x.poisoned) ? 1u : 0u;
    return FW_OK;
}

fw_status fw_spawner_set_origin(fw_ctx *ctx, fw_spawner h, const float t[3], const float r[4]) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !t || !r) return FW_EINVAL;
    memcpy(sp->origin_pos, t, sizeof sp->origin_pos);
    memcpy(sp->origin_rot, r, sizeof sp->origin_rot);
    return FW_OK;
}

fw_status fw_ctx_set_origins(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *translations,
                             const float *rotations_xyzw) {
    if (!ctx || (n && (!handles || !translations || !rotations_xyzw))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_set_origins: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) {
        SpawnerHost &sp = ctx->spawners[(size_t)handles[i]];
        memcpy(sp.origin_pos, translations + (size_t)i * 3, sizeof sp.origin_pos);
        memcpy(sp.origin_rot, rotations_xyzw + (size_t)i * 4, sizeof sp.origin_rot);
    }
    return FW_OK;
}

fw_status fw_spawner_set_parent_velocity(fw_ctx *ctx, fw_spawner h, const float v[3]) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !v) return FW_EINVAL;
    memcpy(sp->parent_vel, v, sizeof sp->parent_vel);
    return FW_OK;
}

fw_status fw_spawner_set_modifier(fw_ctx *ctx, fw_spawner h, float scale, float speed) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    sp->mod_scale = scale, sp->mod_speed = speed;
    return FW_OK;
}

fw_status fw_spawner_queue(fw_ctx *ctx, fw_spawner h, uint64_t count) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    sp->manual_queued_count += count;  // core.rs:284-286
    return FW_OK;
}

fw_status fw_ctx_set_parent_velocities(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *velocities) {
    if (!ctx || (n && (!handles || !velocities))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_set_parent_velocities: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) memcpy(ctx->spawners[(size_t)handles[i]].parent_vel, velocities + (size_t)i * 3, 3 * sizeof(float));
    return FW_OK;
}

fw_status fw_ctx_set_modifiers(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *scales, const float *speeds) {
    if (!ctx || (n && (!handles || !scales || !speeds))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_set_modifiers: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) {
        SpawnerHost &sp = ctx->spawners[(size_t)handles[i]];
        sp.mod_scale = scales[i], sp.mod_speed = speeds[i];
    }
    return FW_OK;
}

fw_status fw_ctx_queue(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const uint64_t *counts) {
    if (!ctx || (n && (!handles || !counts))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_queue: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) ctx->spawners[(size_t)handles[i]].manual_queued_count += counts[i];  // core.rs:284-286
    return FW_OK;
}

// ---- outputs -----------------------------------------------------------------------------
fw_status fw_spawner_counts(fw_ctx *ctx, fw_spawner h, uint32_t *per_type, uint32_t n_types) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !per_type) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    for (uint32_t t = 0; t < n_types && t < sp->seg.size(); t++) per_type[t] = c[sp->seg[t]];
    return st;
}

fw_status fw_spawner_active(fw_ctx *ctx, fw_spawner h, int32_t *out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !out) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    *out = spawner_active(ctx, *sp, c) ? 1 : 0;
    return st;
}

fw_status fw_spawner_poll_finished(fw_ctx *ctx, fw_spawner h, int32_t *out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !out) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    bool all_empty = true;
    for (uint32_t si : sp->seg) all_empty &= c[si] == 0;
    *out = 0;
    if (all_empty && !spawner_active(ctx, *sp, c) && sp->initialized && !sp->finished_notified) {  // core.rs:679-686
        sp->finished_notified = true;
        *out = 1;
    }
    return st;
}

static fw_status stage_buffer(fw_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->stage_bytes) {
        fw_status st = sync(ctx);
        if (st) return st;
        if (ctx->d_stage) FW_HIP(ctx, hipFree(ctx->d_stage));
        ctx->d_stage = nullptr, ctx->stage_bytes = 0;
        const size_t nb = (std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20) + 65535u) & ~(size_t)65535u;
        FW_HIP(ctx, hipMalloc(&ctx->d_stage, nb));
        ctx->stage_bytes = nb;
    }
    *out = ctx->d_stage;
    return FW_OK;
}

static fw_status read_records(fw_ctx *ctx, const char *buf, uint32_t cap_seg, uint32_t n, int32_t pbr, bool aos,
                              fw_particle *out, uint64_t cap, uint32_t head = 0, const float *const_rot = nullptr,
                              uint32_t life_plane = 0xFFFFFFFFu, float life_const = 0.f, const FwType *derived = nullptr, bool cpl = false) {
    const uint64_t m = std::min<uint64_t>(n, cap);
    if (!m || !out) return FW_OK;
    if (aos) {
        FW_HIP(ctx, hipMemcpy(out, buf, m * sizeof(fw_particle), hipMemcpyDeviceToHost));
        return FW_OK;
    }
    void *tmp = nullptr;
    fw_status st = stage_buffer(ctx, m * sizeof(fw_particle), &tmp);
    if (st) return st;
    hipError_t e = fw_launch_gather(ctx->stream, buf, cap_seg, head, (uint32_t)m, pbr, tmp, const_rot, life_plane, life_const,
                                    derived, ctx->d_keys.d, cpl);
    // (the copy goes through the stream the kernel ran on, then one wait for both)
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, m * sizeof(fw_particle), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    FW_HIP(ctx, e);
    return FW_OK;
}

fw_status fw_spawner_read_particles(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle *out, uint64_t cap,
                                    uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const SegHost &S = ctx->segs[sp->seg[type]];
    const uint32_t n = c[sp->seg[type]];
    if (n_out) *n_out = n;
    fw_status st2 = read_records(ctx, S.buf[ctx->parity], S.capacity, n, sp->types[type].ps.pbr, false, out, cap,
                                 ring_head_exact(S, n), S.nospin ? S.const_rot : nullptr,
                                 (S.nospin && !S.fifo) ? S.n_lplanes : 0xFFFFFFFFu, S.fifo_life,
                                 S.derived ? ctx->d_types.d + S.type_idx : nullptr, S.ring());
    return st2 ? st2 : st;
}

fw_status fw_spawner_read_last_emitted(fw_ctx *ctx, fw_spawner h, uint32_t type, uint32_t emission_index, float *out,
                                       uint64_t cap, uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || emission_index >= sp->em.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const SegHost &S = ctx->segs[sp->seg[type]];
    const uint32_t n = c[sp->seg[type]];
    if (n_out) *n_out = n;
    const uint64_t m = std::min<uint64_t>(n, cap);
    if (!m || !out) return st;
    int plane = -1;
    for (uint32_t k = 0; k < S.n_lplanes; k++)
        if (S.lplane_emission[k] == (int32_t)emission_index) plane = (int)k;
    if (plane < 0) {
        for (uint64_t i = 0; i < m; i++) out[i] = FW_F32_MIN;  // never touched: still vec![f32::MIN; n] (core.rs:467)
        return st;
    }
    const char *pl = S.buf[ctx->parity] + FW_OFF_L((size_t)S.capacity, plane);
    const uint32_t h0 = ring_head_exact(S, n);  // a ring: from the head to the end of the buffer, then from slot 0
    const uint64_t m1 = std::min<uint64_t>(m, S.capacity - h0);
    FW_HIP(ctx, hipMemcpy(out, pl + (size_t)h0 * sizeof(float), m1 * sizeof(float), hipMemcpyDeviceToHost));
    if (m > m1) FW_HIP(ctx, hipMemcpy(out + m1, pl, (m - m1) * sizeof(float), hipMemcpyDeviceToHost));
    return st;
}

fw_status fw_spawner_write_particles(fw_ctx *ctx, fw_spawner h, uint32_t type, const fw_particle *in, uint64_t n) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || (n && !in) || n > 0xF0000000ull) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    const uint32_t si = sp->seg[type];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    if ((st = fifo_to_general(ctx, si))) return st;  // ages and lifetimes will be whatever the caller writes
    leave_small(ctx, ctx->segs[si]);                  // (any number of particles, any colours: the compacting kernels take it from here)
    if ((st = leave_nospin(ctx, si))) return st;      // ... and so will rotations and angular velocities
    // ... and scales and colours: the planes are stored and read again until every particle has been through an update
    // (an attached buffer keeps receiving records; the mode comes back after the next step)
    if ((st = set_derived(ctx, si, false, false))) return st;
    ctx->segs[si].derive_pending = wants_derived(ctx, ctx->segs[si]);
    if (n > ctx->segs[si].capacity) {
        ctx->segs[si].ub = 0;
        const uint32_t zero = 0;
        FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)ctx->parity * ctx->max_seg + si, &zero, 4, hipMemcpyHostToDevice));
        if ((st = grow_segment(ctx, si, (uint32_t)n))) return st;
    }
    SegHost &S = ctx->segs[si];
    if (n) {
        void *tmp = nullptr;
        if ((st = stage_buffer(ctx, n * sizeof(fw_particle), &tmp))) return st;
        hipError_t e = hipMemcpyAsync(tmp, in, n * sizeof(fw_particle), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = fw_launch_scatter(ctx->stream, S.buf[ctx->parity], S.capacity, (uint32_t)n, S.n_lplanes, tmp);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        FW_HIP(ctx, e);
    }
    const uint32_t n32 = (uint32_t)n;
    FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)ctx->parity * ctx->max_seg + si, &n32, 4, hipMemcpyHostToDevice));
    S.ub = n32;
    S.win_ok = false;  // ages and lifetimes are now whatever the caller wrote
    S.colors_dirty = true, ctx->colors_dirty = true;  // and so are the colours: the next update writes them all
    for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;
    return FW_OK;
}

fw_status fw_spawner_write_last_emitted(fw_ctx *ctx, fw_spawner h, uint32_t type, uint32_t emission_index,
                                        const float *in, uint64_t n) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || emission_index >= sp->em.size() || (n && !in)) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    if ((st = fifo_to_general(ctx, sp->seg[type]))) return st;  // (caller-written state: the general path takes over)
    const SegHost &S = ctx->segs[sp->seg[type]];
    int plane = -1;
    for (uint32_t k = 0; k < S.n_lplanes; k++)
        if (S.lplane_emission[k] == (int32_t)emission_index) plane = (int)k;
    if (plane < 0) return FW_OK;  // entry never reads it
    const uint64_t m = std::min<uint64_t>(n, S.capacity);
    if (m)
        FW_HIP(ctx, hipMemcpy(S.buf[ctx->parity] + FW_OFF_L((size_t)S.capacity, plane), in, m * sizeof(float),
                              hipMemcpyHostToDevice));
    return FW_OK;
}

fw_status fw_spawner_read_destroyed(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle *out, uint64_t cap,
                                    uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    const SegHost &S = ctx->segs[sp->seg[type]];
    uint32_t n = 0;
    if (S.destroyed) FW_HIP(ctx, hipMemcpy(&n, ctx->g.ndestroyed + sp->seg[type], 4, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    // (a range ring fills its records from the END of the buffer, the youngest dead first: the last n are in list order)
    const char *first = S.destroyed + (S.dead_at_end && n <= S.capacity ? (size_t)(S.capacity - n) * sizeof(fw_particle) : (size_t)0);
    return read_records(ctx, first, S.capacity, n, 0, true, out, cap);
}

fw_status fw_spawner_pack_instances_device(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap,
                                           uint64_t *n_upper_bound) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || !d_out) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    const uint32_t si = sp->seg[type];
    const SegHost &S = ctx->segs[si];
    const uint32_t ub = (uint32_t)std::min<uint64_t>(S.nested_fed ? S.capacity : std::min(S.ub, S.capacity), cap);
    if (n_upper_bound) *n_upper_bound = ub;
    if (S.fifo || S.small) {  // (whose launches may run on the ring stream)
        fw_status jst = join_side(ctx);
        if (jst) return jst;
    }
    // (a range ring: particle 0 sits `count - young_n` slots before the first young particle -- the kernel reads the count)
    FW_HIP(ctx, fw_launch_pack_instances(ctx->stream, S.buf[ctx->parity], S.capacity, S.range ? S.young_lo : (S.fifo ? S.head : 0u),
                                         ctx->g.count + (size_t)ctx->parity * ctx->max_seg + si, ub, d_out,
                                         S.nospin ? S.const_rot : nullptr, S.range ? ctx->g.rold + (size_t)ctx->parity * ctx->max_seg + si : nullptr,
                                         S.derived ? ctx->d_types.d + S.type_idx : nullptr, ctx->d_keys.d, S.life_plane(), S.fifo_life, S.ring()));
    return FW_OK;
}

static fw_status attach_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap, bool window) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || (d_out && !cap)) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);  // kernels in flight hold the old record
    // ... and whatever the caller enqueued on ITS streams to initialise the buffer has happened before a frame writes to it
    // (the context's streams are non-blocking ones: nothing else orders them against, say, a fill on the null stream)
    if (!st && d_out) FW_HIP(ctx, hipDeviceSynchronize());
    if (st) return st;
    if (d_out && !window && ctx->segs[sp->seg[type]].range) {
        // records at index `list position` counted from 0: a tile of a range ring only knows that once the whole old part
        // has been counted -- such a type continues on the compacting path (the windowed attach keeps it a ring)
        if ((st = fifo_to_general(ctx, sp->seg[type]))) return st;
    }
    SegHost &S = ctx->segs[sp->seg[type]];
    // (a small type keeps its kernel: fw_k_update_small<INST> writes the records of its survivors itself)
    ctx->n_inst += (d_out != nullptr ? 1u : 0u) - (S.inst != nullptr ? 1u : 0u);
    S.inst = (char *)d_out;
    S.inst_cap = d_out ? (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFFFull) : 0u;
    S.inst_window = d_out != nullptr && window;
    if (S.range) ctx->r_force = true;  // (the range descriptors carry FW_TYPE_IDX_NOLIFE: derived AND no buffer attached)
    // the records carry scale and colours from now on: the update stops storing the three planes that would duplicate them
    // (every reader of those planes evaluates them instead, so a buffer smaller than the live count loses nothing either);
    // colliding types stay as they are (the feature path)
    const bool derive = wants_derived(ctx, S);
    S.derive_pending = derive && S.colors_dirty;  // (particles written by the caller, not updated yet: one frame later)
    if ((st = set_derived(ctx, sp->seg[type], derive && !S.colors_dirty))) return st;
    return upload_seg(ctx, sp->seg[type]);
}

fw_status fw_spawner_attach_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap) {
    return attach_instances(ctx, h, type, d_out, cap, false);
}

fw_status fw_spawner_attach_instances_window(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap) {
    return attach_instances(ctx, h, type, d_out, cap, true);
}

fw_status fw_spawner_instance_window(fw_ctx *ctx, fw_spawner h, uint32_t type, uint64_t *first, uint64_t *count) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || !first || !count) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const uint32_t si = sp->seg[type];
    const SegHost &S = ctx->segs[si];
    uint32_t dead = 0;
    // a range ring's update numbers its records from the particles it destroys (fw_k_update_range): they sit behind them
    if (S.dead_at_end && S.inst_window) FW_HIP(ctx, hipMemcpy(&dead, ctx->g.ndestroyed + si, 4, hipMemcpyDeviceToHost));
    *first = dead, *count = c[si];
    return st;
}

fw_status fw_spawner_pack_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle_instance *out, uint64_t cap,
                                    uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const uint32_t n = c[sp->seg[type]];
    if (n_out) *n_out = n;
    const uint64_t m = std::min<uint64_t>(n, cap);
    if (!m || !out) return st;
    void *tmp = nullptr;
    fw_status sst = stage_buffer(ctx, m * sizeof(fw_particle_instance), &tmp);
    if (sst) return sst;
    uint64_t ub = 0;
    fw_status st2 = fw_spawner_pack_instances_device(ctx, h, type, tmp, m, &ub);
    hipError_t e = hipMemcpyAsync(out, tmp, m * sizeof(fw_particle_instance), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    FW_HIP(ctx, e);
    return st2 ? st2 : st;
}

fw_status fw_spawner_aabb(fw_ctx *ctx, fw_spawner h, float out_min[3], float out_max[3], int32_t *any) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !out_min || !out_max) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    if (!ctx->h_aabb) FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_aabb, 8 * sizeof(float), hipHostMallocDefault));
    {
        fw_status jst = join_side(ctx);  // (the query kernels run on the main stream and may read rings)
        if (jst) return jst;
    }
    // the query kernels take up to eight particle types at a time (their segment list rides in the kernel arguments): a
    // spawner with more is answered in chunks, folded here -- min / max are exact and order-independent
    const size_t nt = sp->seg.size();
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    bool got = false;
    fw_status keep = FW_OK;
    for (size_t t0 = 0; t0 < nt || t0 == 0; t0 += 8) {
        const uint32_t n = (uint32_t)std::min<size_t>(8, nt - std::min(nt, t0));
        if (!n) break;
        bool any_ring = false;  // rings (and types a wave / workgroup of fw_k_update_small updates) leave no per-tile boxes: the two-pass query reads them
        uint32_t heads[8] = {}, range_y[8], life_plane[8];
        float life_const[8];
        for (uint32_t t = 0; t < n; t++) {
            const SegHost &S = ctx->segs[sp->seg[t0 + t]];
            any_ring |= S.ring() || S.small;
            heads[t] = S.range ? S.young_lo : (S.fifo ? S.head : 0u);
            range_y[t] = S.range ? 1u : 0xFFFFFFFFu;  // (a range ring: the kernel takes the old part's size from FwGlobals::rold)
            life_plane[t] = S.life_plane(), life_const[t] = S.fifo_life;
        }
        if (ctx->boxes_epoch && ctx->d_tile_first && !any_ring) {
            // the last update left the box of every tile's survivors (fw_ctx_track_aabbs): fold those -- one small launch
            FW_HIP(ctx, fw_launch_aabb_from_tiles(ctx->stream, ctx->g, sp->seg.data() + t0, n, ctx->parity, ctx->boxes_epoch,
                                                  ctx->d_tile_first, ctx->h_aabb));
        } else {
            // two launches over the particles, the result lands in pinned memory: one synchronisation, no copies
            FW_HIP(ctx, fw_launch_aabb(ctx->stream, ctx->g, sp->seg.data() + t0, heads, n, ctx->parity, ctx->d_aabb, ctx->h_aabb,
                                       range_y, life_plane, life_const));
        }
        fw_status st = sync(ctx);
        if (!st) st = check_device_errors(ctx);
        if (st && st != FW_ECAPACITY) return st;
        if (st) keep = st;
        const volatile float *r = ctx->h_aabb;
        if (r[3] != 0.0f) {
            for (int c = 0; c < 3; c++) {
                mn[c] = got ? std::min(mn[c], (float)r[c]) : (float)r[c];
                mx[c] = got ? std::max(mx[c], (float)r[4 + c]) : (float)r[4 + c];
            }
            got = true;
        } else if (!got && t0 + 8 >= nt) {  // nothing anywhere: report the last chunk's (empty) box as before
            for (int c = 0; c < 3; c++) mn[c] = r[c], mx[c] = r[4 + c];
        }
    }
    if (any) *any = got ? 1 : 0;
    for (int c = 0; c < 3; c++) out_min[c] = mn[c], out_max[c] = mx[c];
    return keep;
}

fw_status fw_ctx_track_aabbs(fw_ctx *ctx, int32_t enable) {
    if (!ctx) return FW_EINVAL;
    ctx->track_aabb = enable != 0;
    if (!enable) ctx->boxes_epoch = 0;
    // (per-tile boxes are left by the compacting kernels' tiles; rings and small types are answered by the two-pass query)
    return FW_OK;
}

fw_status fw_ctx_live_count(fw_ctx *ctx, uint64_t *out) {
    if (!ctx || !out) return FW_EINVAL;
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    uint64_t t = 0;
    for (size_t i = 0; i < c.size(); i++)
        if (ctx->segs[i].in_use) t += c[i];
    *out = t;
    return st;
}

fw_status fw_ctx_live_count_device(fw_ctx *ctx, void *d_out_u64) {
    if (!ctx || !d_out_u64) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status jst = join_side(ctx);  // the counts of ring segments are written by the side stream's launches
    if (jst) return jst;
    FW_HIP(ctx, fw_launch_total(ctx->stream, ctx->g.count + (size_t)ctx->parity * ctx->max_seg,
                                (uint32_t)ctx->segs.size(), (unsigned long long *)d_out_u64));
    return FW_OK;
}

fw_status fw_ctx_live_count_ring(fw_ctx *ctx, void *d_ring_u64, uint32_t n_slots) {
    if (!ctx || (d_ring_u64 && n_slots < 2)) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (!st && d_ring_u64) FW_HIP(ctx, hipDeviceSynchronize());  // (the caller's zero-fill of the ring, on whatever stream)
    if (st) return st;
    ctx->live_ring = (unsigned long long *)d_ring_u64;
    ctx->live_ring_n = d_ring_u64 ? n_slots : 0;
    ctx->live_ring_frames = 0;
    if (d_ring_u64) FW_HIP(ctx, fw_memset_done(d_ring_u64, 0, (size_t)n_slots * sizeof(unsigned long long)));
    return FW_OK;
}

fw_status fw_ctx_last_step_updated(fw_ctx *ctx, uint64_t *out) {
    if (!ctx || !out) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    unsigned long long now = 0, slots[FW_STAT_SLOTS];
    FW_HIP(ctx, hipMemcpy(slots, ctx->g.stats, sizeof slots, hipMemcpyDeviceToHost));
    for (unsigned long long v : slots) now += v;
    *out = now;  // running total of particles that entered update_particles
    return FW_OK;
}

fw_status fw_ctx_kernel_timing(fw_ctx *ctx, int32_t enable) {
    if (!ctx) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    if (enable && ctx->tev.empty()) {
        ctx->tev.resize(kTimingEvents);
        for (auto &ev : ctx->tev) FW_HIP(ctx, hipEventCreate(&ev));
    }
    if (enable) {
        // for reference only: what an empty hipEventRecord pair on the stream costs (the timed launches do not use
        // marker packets: their events are attached to the dispatch, see fw_launch_update)
        const int n = 64;
        for (int i = 0; i < n; i++) {
            FW_HIP(ctx, hipEventRecord(ctx->tev[2 * i], ctx->stream));
            FW_HIP(ctx, hipEventRecord(ctx->tev[2 * i + 1], ctx->stream));
        }
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double tot = 0;
        for (int i = 8; i < n; i++) {
            float t = 0;
            FW_HIP(ctx, hipEventElapsedTime(&t, ctx->tev[2 * i], ctx->tev[2 * i + 1]));
            tot += t;
        }
        ctx->tev_overhead_ms = tot / (n - 8);
    }
    ctx->timing = enable != 0;
    ctx->tev_used = 0, ctx->tev_frames = 0;
    unsigned long long now = 0, slots[FW_STAT_SLOTS];
    FW_HIP(ctx, hipMemcpy(slots, ctx->g.stats, sizeof slots, hipMemcpyDeviceToHost));
    for (unsigned long long v : slots) now += v;
    ctx->timing_particles_start = now;
    return FW_OK;
}

fw_status fw_ctx_kernel_timing_read(fw_ctx *ctx, double *ms_total, uint64_t *launches, uint64_t *particles) {
    if (!ctx) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    double ms = 0;
    for (size_t i = 0; i + 1 < ctx->tev_used; i += 2) {
        float t = 0;
        FW_HIP(ctx, hipEventElapsedTime(&t, ctx->tev[i], ctx->tev[i + 1]));
        ms += t;
    }
    const uint64_t nl = ctx->tev_frames;  // frames: a frame's update may be several launches (FIFO + general), all summed
    if (ms_total) *ms_total = ms;
    if (launches) *launches = nl;
    unsigned long long now = 0, slots[FW_STAT_SLOTS];
    FW_HIP(ctx, hipMemcpy(slots, ctx->g.stats, sizeof slots, hipMemcpyDeviceToHost));
    for (unsigned long long v : slots) now += v;
    if (particles) *particles = now - ctx->timing_particles_start;
    return FW_OK;
}

// profiling hook (not in the public header): per-tile timestamps of the last update (and, with `prev`, of the one
// before it: the two are kept apart by launch parity) when FW_DEBUG & 8
fw_status fw_debug_read_timestamps2(fw_ctx *ctx, unsigned long long *out, unsigned long long *prev, uint64_t max_tiles,
                                    uint64_t *n_tiles) {
    if (!ctx || !ctx->g.dbg_ts) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    const uint64_t n = std::min<uint64_t>(max_tiles, ctx->total_tiles_dev);
    if (n_tiles) *n_tiles = n;
    const uint32_t last = (uint32_t)(ctx->frame & 1u);  // epoch of the last launch = frame (after the increment)
    const size_t stride = (size_t)ctx->total_tiles_dev * 8;
    if (n && out)
        FW_HIP(ctx, hipMemcpy(out, ctx->g.dbg_ts + 32768 + last * stride, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (n && prev)
        FW_HIP(ctx, hipMemcpy(prev, ctx->g.dbg_ts + 32768 + (last ^ 1u) * stride, n * 8 * sizeof(unsigned long long),
                              hipMemcpyDeviceToHost));
    return FW_OK;
}
// ... and of the last range-ring launch: 8 words per workgroup {start, 0, 0, end of wave 0, role_k, segment, 0, 0}
fw_status fw_debug_read_range_timestamps(fw_ctx *ctx, unsigned long long *out, uint64_t max_tiles, uint64_t *n_tiles) {
    if (!ctx || !ctx->d_rts) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    const uint64_t n = std::min<uint64_t>(max_tiles, ctx->r_total);
    if (n_tiles) *n_tiles = n;
    if (n && out) FW_HIP(ctx, hipMemcpy(out, ctx->d_rts, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FW_OK;
}
// {~earliest workgroup start [64], latest workgroup end [64]} of the last 256 update launches (slot = epoch & 255);
// *epoch = the last launch's.  out512: room for 32768 words.
fw_status fw_debug_read_launches(fw_ctx *ctx, unsigned long long *out512, uint32_t *epoch) {
    if (!ctx || !ctx->g.dbg_ts || !out512) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    FW_HIP(ctx, hipMemcpy(out512, ctx->g.dbg_ts, 32768 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (epoch) *epoch = (uint32_t)(ctx->frame & 0x3FFFFFFFu);
    return FW_OK;
}
// which update path a particle type is on: *mode = 1 FIFO ring (in place), 0 general (compacting); *moved_bytes = the
// bytes one update of a live particle reads and writes on that path; *algorithmic_bytes = the part of them that carries
// state the update needs or changes (SURVEY.md 8(d)'s convention: a float4 plane rewritten for three changed components
// moves 4 bytes -- initial_scale, lifetime -- that are not algorithmic)
fw_status fw_debug_update_path(fw_ctx *ctx, fw_spawner h, uint32_t type, int32_t *mode, uint32_t *moved_bytes,
                               uint32_t *algorithmic_bytes) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    const SegHost &S = ctx->segs[sp->seg[type]];
    const TypeHost &T = sp->types[type];
    if (mode) *mode = S.fifo ? 1 : (S.range ? 2 : (S.small ? ((S.wide && !ctx->wave_all_on) ? 4 : 3) : 0));
    const uint32_t colours = (T.base.kind != 0 ? 16u : 0u) + (T.emis.kind != 0 ? 16u : 0u);  // one-key gradients: never rewritten
    uint32_t moved, algo;
    if (S.ring()) {
        // in place: position+age and velocity always; rotation only where some emitter makes the particles spin (or the
        // type accelerates them), angular velocity only if it then changes; scale unless its curve is constant
        bool spins = false;
        for (const EmissionHost &E : sp->em)
            if ((uint32_t)E.es.particle_index == type)
                spins |= !(E.es.initial_angular_velocity.magnitude.min == 0.f && E.es.initial_angular_velocity.magnitude.max == 0.f);
        const float *aa = T.ps.angular_acceleration;
        const bool acc = aa[0] != 0.f || aa[1] != 0.f || aa[2] != 0.f;
        const uint32_t q2 = (spins || acc) ? 16u : 0u, q3 = ((spins && T.ps.angular_drag != 0.f) || acc) ? 16u : 0u;
        // (a type that cannot turn: neither the rotation nor the angular-velocity / lifetime plane is read)
        // (round 6: Q1 / Q3 are component planes -- velocity and angular velocity move as 12 bytes each way, the constants that
        // shared their float4, initial_scale and lifetime, stay where they are: every byte moved is algorithmic)
        moved = (S.nospin ? 28u : 56u) + 28u + q2 + (q3 ? 12u : 0u) + (T.scale.kind != 0 ? 4u : 0u) + colours;
        algo = moved;
        // (a range ring: the part of the list that may lose particles, a fifth of configs[2], is compacted in place and reads and
        // rewrites every plane it keeps, the 4-byte lifetime included: the figure is the young part's)
        // (initial_scale -- and in a range ring the lifetime -- are read where somebody evaluates the scale: a type whose planes are
        // stored, an attached instance buffer.  Otherwise the young tile at the boundary to the old part alone reads the lifetimes, for the
        // consistency check: FW_TYPE_IDX_NOLIFE)
        if (!S.derived || S.inst != nullptr) moved += S.range ? 8u : 4u, algo += S.range ? 8u : 4u;
    } else {
        // compacting: every state plane lands at a new slot (+ the last_emitted planes of a Nested parent, read and written);
        // a type that cannot turn keeps no rotation plane: -16 B read, -16 B written,
        // ... and its lifetimes in a 4-byte plane instead of Q3: -32 B again, +4 B read, +4 B written
        moved = 64u + 64u + 4u + colours + 8u * S.n_lplanes - (S.nospin ? 32u + 32u - 8u : 0u);
        algo = moved - 8u;
    }
    if (S.derived) {  // scale and colour planes are not stored (the instance record carries them: +64 B written per particle)
        const uint32_t skipped = colours + ((S.ring() && T.scale.kind == 0) ? 0u : 4u);
        moved -= std::min(moved, skipped), algo -= std::min(algo, skipped);
    }
    // an attached instance buffer: the update also writes the 64-byte ParticleInstance record of every survivor (render.rs:95-103)
    if (S.inst != nullptr) moved += 64u, algo += 64u;
    if (moved_bytes) *moved_bytes = moved;
    if (algorithmic_bytes) *algorithmic_bytes = algo;
    return FW_OK;
}
// frames with Nested entries so far: those whose entries ran inside the FIFO launch (FwFifoNest) / those that ran the separate
// fw_k_spawn / fw_k_nest passes
fw_status fw_debug_nest_frames(fw_ctx *ctx, uint64_t *fused, uint64_t *separate) {
    if (!ctx) return FW_EINVAL;
    if (fused) *fused = ctx->fused_nest_frames;
    if (separate) *separate = ctx->nest_pass_frames;
    return FW_OK;
}
fw_status fw_debug_tile_scratch(fw_ctx *ctx, uint64_t *table_tiles, uint64_t *scratch_tiles) {
    if (!ctx) return FW_EINVAL;
    if (table_tiles) *table_tiles = ctx->total_tiles_dev;
    if (scratch_tiles) *scratch_tiles = ctx->tiles_cap;
    return FW_OK;
}
fw_status fw_debug_recovered_rings(fw_ctx *ctx, uint64_t *n) {
    if (!ctx || !n) return FW_EINVAL;
    *n = ctx->recovered_rings;
    return FW_OK;
}
fw_status fw_debug_tf_frames(fw_ctx *ctx, uint64_t *n) {
    if (!ctx || !n) return FW_EINVAL;
    *n = ctx->tf_frames;
    return FW_OK;
}
fw_status fw_debug_param_bar(fw_ctx *ctx, int32_t *on) {  // fw_ctx::param_bar
    if (!ctx || !on) return FW_EINVAL;
    *on = ctx->param_bar ? 1 : 0;
    return FW_OK;
}
fw_status fw_debug_read_timestamps(fw_ctx *ctx, unsigned long long *out, uint64_t max_tiles, uint64_t *n_tiles) {
    return fw_debug_read_timestamps2(ctx, out, nullptr, max_tiles, n_tiles);
}

fw_status fw_ctx_kernel_timing_overhead(fw_ctx *ctx, double *ms_per_pair) {
    if (!ctx || !ms_per_pair) return FW_EINVAL;
    *ms_per_pair = ctx->tev_overhead_ms;
    return FW_OK;
}

fw_status fw_ctx_measure_copy_bandwidth(fw_ctx *ctx, uint64_t bytes, int32_t iters, double *bytes_per_s) {
    if (!ctx || !bytes_per_s || bytes < 4096 || iters < 1) return FW_EINVAL;
    hipSetDevice(ctx->device);
    bytes &= ~(uint64_t)0xFFF;
    void *a = nullptr, *b = nullptr;
    FW_HIP(ctx, hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) {
        hipFree(a);
        return fail(ctx, FW_ENOMEM, "copy probe allocation");
    }
    fw_memset_done(a, 1, bytes);
    fw_memset_done(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) fw_launch_copy_probe(ctx->stream, a, b, bytes);
    hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < iters; i++) fw_launch_copy_probe(ctx->stream, (i & 1) ? b : a, (i & 1) ? a : b, bytes);
    hipEventRecord(e1, ctx->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0), hipEventDestroy(e1);
    hipFree(a), hipFree(b);
    FW_HIP(ctx, e);
    *bytes_per_s = 2.0 * (double)bytes * iters / (ms * 1e-3);
    return FW_OK;
}

}  // extern "C"
