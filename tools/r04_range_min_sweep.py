"""FW_RANGE_MIN sweep: us per frame of many equal emitters, per emitter size, for several capacity thresholds, interleaved
in one process on one box (tools/small_emitters.py's loop).  Usage: python tools/r04_range_min_sweep.py > gpurun_out/x.txt"""
import os, sys, time
os.environ["FW_ENABLE_KNOBS"] = "1"
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
CASES = ((2048, 200), (1024, 600), (1024, 1000), (512, 1500), (512, 2000), (512, 3000), (256, 4000), (256, 6000), (128, 8000))
MINS = tuple(int(x) for x in os.environ.get("FW_SWEEP_MINS", "12288,8192,6144,5120").split(","))
print("emitters x live  | " + " | ".join(f"min {m:5d}" for m in MINS) + "   (us per frame, best of 2; paths)")
for n_em, per in CASES:
    row, paths = [], []
    for m in MINS:
        best = 1e9
        for rep in range(2):
            os.environ["FW_RANGE_MIN"] = str(m)
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
            if rep == 0: paths.append(hs[0].update_path(0)[0][0])
            ps.close()
        row.append(best)
    print(f"{n_em:5d} x {per:5d}    | " + " | ".join(f"{v:9.1f}" for v in row) + "   " + "".join(paths), flush=True)
