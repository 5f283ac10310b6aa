"""1-64 small emitters (examples/sparks.rs at 1000/s) at product defaults: us per frame pipelined / synchronised -- a quick form of
tools/r04_few_small_emitters.py for same-box A/Bs of library variants (FW_LIB_PATH)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.settings import EmissionPacing, Transform
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
out = []
for n_em in (1, 8, 64):
    ps = ParticleSystem(seed=workloads.SEED)
    hs = [ps.spawn(workloads.example_sparks(EmissionPacing.rate(1000.0))[0], Transform((2.0 * e, 0.1, 0.0)), uid=e) for e in range(n_em)]
    ps.update(dt)
    for _ in range(70): ps.step(dt)
    best = 1e9
    for rep in range(5):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(400): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
    t0 = time.perf_counter()
    for _ in range(200): ps.step(dt); ps.synchronize()
    sync = (time.perf_counter() - t0) / 200 * 1e6
    out.append(f"{n_em} x sparks [{hs[-1].update_path(0)[0]}] {best:.2f} / {sync:.2f}")
    ps.close()
print("  ".join(out))
