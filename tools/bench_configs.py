#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one GPU (not the bench line; for DESIGN.md / profiles)."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import json, os, sys, time
import numpy as np
import torch  # (before the library: two HIP runtimes in one process must be loaded in this order)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import workloads, sharding
from bevy_firework_amd.system import ParticleSystem

dt = np.float32(1 / 60)


ATTACH = os.environ.get("FW_BENCH_ATTACH", "")  # "window" / "plain": every type gets a device buffer for its ParticleInstance records


def run(name, spawners, fill, steps, uids=None, inst_cap=None, colliders=None):
    ps = ParticleSystem(seed=workloads.SEED)
    if colliders:
        ps.set_colliders(colliders)
    keep = []
    for i, (sp, tf) in enumerate(spawners):
        h = ps.spawn(sp, tf, uid=(uids[i] if uids else i))
        if ATTACH and inst_cap:
            for t, cap in enumerate(inst_cap):
                buf = torch.empty(cap * 64, dtype=torch.uint8, device="cuda")
                keep.append(buf)
                (h.attach_instances_window if ATTACH == "window" else h.attach_instances)(buf.data_ptr(), cap, particle_type=t)
    if ATTACH and inst_cap:
        name += f" + {ATTACH} instance buffers"
    ps.update(dt)
    for _ in range(fill):
        ps.step(dt)
    ps.synchronize()
    # (a free-running host is hundreds of frames ahead of the device: whatever it does when it SEES the device's state -- a
    # Nested-fed segment past half its derived capacity is reallocated once, 30 ms at 4M particles -- happens in the first frames
    # after this wait, not during the fill; keep that out of the timed region)
    for _ in range(8):
        ps.step(dt)
    ps.synchronize()
    u0 = ps.updated_total()
    t0 = time.perf_counter()
    for _ in range(steps):
        ps.step(dt)
    ps.synchronize()
    el = time.perf_counter() - t0
    upd = ps.updated_total() - u0
    live = ps.live_count()
    # bytes per particle-update from the library, per particle TYPE (fw_debug_update_path), weighted by the live counts:
    # a type that cannot turn, or whose gradients are constant, moves far fewer than SURVEY's 156 B
    algo_b = moved_b = 0.0
    for h in ps.spawners.values():
        for t, c in enumerate(h.counts()):
            _, moved, algo = h.update_path(t)
            algo_b += algo * c
            moved_b += moved * c
    algo_pp, moved_pp = algo_b / max(live, 1), moved_b / max(live, 1)
    # the update launches alone (hipEvent pairs attached to the dispatches, the duration rocprofv3 reports)
    ps.kernel_timing(True)
    for _ in range(min(steps, 200)):
        ps.step(dt)
    ev_ms, launches, kparts = ps.kernel_timing_read()
    ps.kernel_timing(False)
    k_us = ev_ms * 1e3 / max(launches, 1)
    print(json.dumps({"config": name, "live": live, "us_per_step": el / steps * 1e6, "particles_per_s": upd / el,
                      "algorithmic_bytes_per_particle": round(algo_pp, 1), "moved_bytes_per_particle": round(moved_pp, 1),
                      "algorithmic_GBps": upd / el * algo_pp / 1e9, "frac_of_8TBps": upd / el * algo_pp / 8e12,
                      "update_kernels_us_per_frame": k_us,
                      "update_kernels_algorithmic_GBps": kparts / max(launches, 1) * algo_pp / (k_us * 1e-6) / 1e9 if k_us else None,
                      "at_survey_156B_GBps": upd / el * 156 / 1e9}))
    ps.close()


which = sys.argv[1:] or ["c3", "c4", "c5", "c1"]
if "c1" in which:
    run("configs[0] stress_test rate 160000", [workloads.stress_test(160000.0)], 70, 600)
if "c3" in which:
    run("configs[2] 256 emitters x 64Ki", workloads.many_emitters(256, 65536), 80, 100, inst_cap=[110000])
if "c5" in which:
    ems = workloads.many_emitters(4096, 8192)
    mine = sharding.local_indices(4096, 0, 8)
    run("configs[4] one GPU's share: 512 of 4096 emitters x 8192", [ems[e] for e in mine], 80, 200, uids=mine, inst_cap=[20000])
if "c4" in which:
    run("configs[3] nested sparks->smoke ~4M", [workloads.nested(100000.0, 20.0)], 250, 100)
if "cc" in which:  # examples/stress_test_collision.rs: bouncing particles (at the example's rate, and at 8x it: 1.26M live)
    sp, tf, world = workloads.stress_test_collision(80000.0)
    run("stress_test_collision rate 80000 (~157k live)", [(sp, tf)], 130, 600, colliders=world)
    sp, tf, world = workloads.stress_test_collision(640000.0)
    run("stress_test_collision rate 640000 (~1.26M live)", [(sp, tf)], 130, 300, colliders=world)
if "r1" in which:  # the reference's stress_test with a lifetime RANGE (0.8-1.2 s): one mid-size range ring
    from bevy_firework_amd import settings as _S
    for rate in (160000.0, 500000.0):
        sp, tf = workloads.stress_test(rate)
        sp.particle_settings[0].lifetime = _S.RandF32(0.8, 1.2)
        run("stress_test with lifetimes 0.8-1.2 s, rate %d" % rate, [(sp, tf)], 90, 600)
