"""A FEW small emitters in one context -- the reference's own regime (examples/sparks.rs: one spawner, ~730 particles) -- on the three
update paths: product defaults (general below the ring thresholds), FIFO rings from any size (FW_FIFO_MIN=0), range rings from any
size (FW_FIFO=0 FW_RANGE_MIN=0).  us per frame, pipelined (best of 4 x 300 frames) / with a synchronisation every frame."""
import os; os.environ["FW_ENABLE_KNOBS"] = "1"
import json, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.settings import EmissionPacing, Transform
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
MODES = {"default": {}, "fifo_any_size": {"FW_FIFO_MIN": "0"}, "range_any_size": {"FW_FIFO": "0", "FW_RANGE_MIN": "0"}}
def run(n_em, rate, mode):
    for k in ("FW_FIFO", "FW_FIFO_MIN", "FW_RANGE", "FW_RANGE_MIN"): os.environ.pop(k, None)
    os.environ.update(MODES[mode])
    ps = ParticleSystem(seed=workloads.SEED)
    hs = []
    for e in range(n_em):
        sp, tf = workloads.example_sparks(EmissionPacing.rate(rate))
        hs.append(ps.spawn(sp, Transform((2.0 * e, 0.1, 0.0)), uid=e))
    ps.update(dt)
    for _ in range(70): ps.step(dt)
    ps.synchronize()
    best = 1e9
    for rep in range(4):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(300): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    t0 = time.perf_counter()
    for _ in range(150): ps.step(dt); ps.synchronize()
    sync = (time.perf_counter() - t0) / 150 * 1e6
    path = hs[-1].update_path(0)[0]; live = ps.live_count()
    ps.close()
    return live, path, round(best, 2), round(sync, 2)
for n_em in (1, 2, 4, 8, 9, 16, 64):
    for rate in (270.0, 1000.0, 7000.0, 30000.0):
        row = {"emitters": n_em, "rate": rate}
        for mode in MODES:
            live, path, best, sync = run(n_em, rate, mode)
            row["live"] = live
            row[mode] = {"path_of_last": path, "us": best, "us_sync": sync}
        print(json.dumps(row), flush=True)
