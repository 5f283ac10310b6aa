#!/usr/bin/env python3
"""Range-ring path over a grid of (emitters x live per emitter): us per frame.  Used to compare builds (FW_LIB_PATH) --
e.g. young workgroups of 512 / 1024 slots (FW_RANGE_YR) -- and the compacting path (FW_RANGE=0).  Run on the GPU box."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem

dt = np.float32(1 / 60)
grid = [(1, 16 << 20), (16, 1 << 20), (64, 1 << 18), (256, 1 << 16), (128, 1 << 15), (1024, 1 << 14), (512, 1 << 13), (1, 1 << 20)]
for n_em, live in grid:
    ps = ParticleSystem(seed=workloads.SEED)
    for e, (sp, tf) in enumerate(workloads.many_emitters(n_em, live)):
        ps.spawn(sp, tf, uid=e)
    ps.update(dt)
    for _ in range(90):
        ps.step(dt)
    ps.synchronize()
    for _ in range(8):
        ps.step(dt)
    ps.synchronize()
    steps = 100
    t0 = time.perf_counter()
    for _ in range(steps):
        ps.step(dt)
    ps.synchronize()
    el = time.perf_counter() - t0
    h = next(iter(ps.spawners.values()))
    print(json.dumps({"emitters": n_em, "live_per_emitter": live, "live": ps.live_count(), "path": h.update_path(0)[0],
                      "us_per_step": round(el / steps * 1e6, 1)}), flush=True)
    ps.close()
