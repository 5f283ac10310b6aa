"""one-round against four-round range-ring tiles by segment size at ~1M particles in all (FW_RANGE_SMALL = launch-wide threshold in
four-round tiles; 384 = shipped): us per frame, best of 2.  python tools/r04_range_small_sweep.py"""
import os, sys, time
os.environ["FW_ENABLE_KNOBS"] = "1"
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
print("emitters x live | four-round tiles | one-round tiles")
for n_em, per in ((1024, 1000), (512, 2000), (256, 4000), (128, 8000), (64, 16000), (512, 4000), (512, 8192)):
    row = []
    for small in (384, 1 << 20):
        best = 1e9
        for rep in range(2):
            os.environ["FW_RANGE_SMALL"] = str(small)
            os.environ["FW_RANGE_MIN"] = "0"
            ps = ParticleSystem(seed=workloads.SEED)
            ems = workloads.many_emitters(n_em, per)
            hs = [ps.spawn(ems[e][0], ems[e][1], uid=e) for e in range(n_em)]
            ps.update(dt)
            for _ in range(80): ps.step(dt)
            ps.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): ps.step(dt)
            ps.synchronize(); t2 = time.perf_counter()
            best = min(best, (t2 - t0) / 200 * 1e6)
            path = hs[0].update_path(0)[0]
            ps.close()
        row.append(best)
    print(f"{n_em:5d} x {per:5d} ({path}) | {row[0]:8.1f} | {row[1]:8.1f}", flush=True)
