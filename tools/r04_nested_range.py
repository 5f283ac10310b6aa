#!/usr/bin/env python3
"""configs[3]-shaped spawner (sparks -> smoke) whose lifetimes are RANGES (1.6-2.4 s): range rings (round 4) against the
compacting path, fixed dt and a dt that never repeats.  Prints one JSON line per run (run on the GPU box)."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import json, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import settings as S, workloads
from bevy_firework_amd.system import ParticleSystem

dt = np.float32(1 / 60)
jit = [np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) for k in range(64)]


def spawner():
    sp, tf = workloads.nested(100000.0, 20.0)
    for ps in sp.particle_settings:
        ps.lifetime = S.RandF32(1.6, 2.4)
    return sp, tf


def run(label, env, var):
    for k, v in env.items():
        os.environ[k] = v
    with ParticleSystem(seed=workloads.SEED) as ps:
        sp, tf = spawner()
        h = ps.spawn(sp, tf, uid=0)
        paths = [h.update_path(t)[0] for t in (0, 1)]
        ps.update(dt)
        for _ in range(300):
            ps.step(dt)
        ps.synchronize()
        for k in range(16):
            ps.step(jit[k % 64] if var else dt)
        ps.synchronize()
        u0 = ps.updated_total()
        t0 = time.perf_counter()
        for k in range(100):
            ps.step(jit[k % 64] if var else dt)
        ps.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps({"run": label, "variable_dt": var, "paths": paths, "live": ps.live_count(), "counts": h.counts(),
                          "us_per_frame": el / 100 * 1e6, "particles_per_s": (ps.updated_total() - u0) / el}), flush=True)
    for k in env:
        os.environ.pop(k, None)


for var in (False, True):
    run("range rings", {}, var)
    run("compacting path", {"FW_RANGE": "0"}, var)
