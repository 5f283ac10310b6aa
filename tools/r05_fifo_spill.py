"""More one-lifetime particle types in a context than ONE FIFO launch holds (round 5, fw_ctx::n_spilled): the product's default path
choice against every type forced onto range rings (FW_FIFO=0), us per frame pipelined (best of 4 x 300 frames) / with a
synchronisation every frame.  Rows: N emitters x 22 000 live (sparks.rs settings, rate 30 000/s) and configs[2] with ONE lifetime
value (256 emitters x 64 Ki).  Run it a second time with FW_LIB_PATH=variants/r04/libfirework_hip.so for the round-4 default
(eight FIFO rings + the rest on range rings, two kinds of launch per frame)."""
import os; os.environ["FW_ENABLE_KNOBS"] = "1"
import json, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.settings import EmissionPacing, RandF32, Transform
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
MODES = {"default": {}, "range_any_size": {"FW_FIFO": "0", "FW_RANGE_MIN": "0"}}
TAG = "r04 library" if os.environ.get("FW_LIB_PATH") else "this build"
def measure(make, mode, fill, reps, frames):
    for k in ("FW_FIFO", "FW_FIFO_MIN", "FW_RANGE", "FW_RANGE_MIN"): os.environ.pop(k, None)
    os.environ.update(MODES[mode])
    ps = ParticleSystem(seed=workloads.SEED)
    hs = make(ps)
    ps.update(dt)
    for _ in range(fill): ps.step(dt)
    ps.synchronize()
    best = 1e9
    for rep in range(reps):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(frames): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / frames * 1e6)
    t0 = time.perf_counter()
    for _ in range(100): ps.step(dt); ps.synchronize()
    sync = (time.perf_counter() - t0) / 100 * 1e6
    paths = [h.update_path(0)[0] for h in hs]
    live = ps.live_count()
    ps.close()
    return {"paths": {p: paths.count(p) for p in sorted(set(paths))}, "us": round(best, 2), "us_sync": round(sync, 2), "live": live}
def sparks(n_em, rate):
    def make(ps):
        hs = []
        for e in range(n_em):
            sp, tf = workloads.example_sparks(EmissionPacing.rate(rate))
            hs.append(ps.spawn(sp, Transform((2.0 * e, 0.1, 0.0)), uid=e))
        return hs
    return make
def configs2_one_lifetime(ps):
    hs = []
    for e, (sp, tf) in enumerate(workloads.many_emitters(256, 65536)):
        sp.particle_settings[0].lifetime = RandF32.constant(1.0)
        hs.append(ps.spawn(sp, tf, uid=e))
    return hs
rows = [(f"{n} emitters x 22k (one lifetime value)", sparks(n, 30000.0), 70, 4, 300) for n in (8, 9, 12, 16, 32, 64)]
rows.append(("configs[2] with ONE lifetime value: 256 emitters x 64Ki", configs2_one_lifetime, 80, 3, 60))
for name, make, fill, reps, frames in rows:
    row = {"build": TAG, "workload": name}
    for mode in MODES:
        row[mode] = measure(make, mode, fill, reps, frames)
    print(json.dumps(row), flush=True)
