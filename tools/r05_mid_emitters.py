"""65-1024 small emitters x 200 particles (the wave-per-type launch, or with FW_SMALL=0 the compacting launch, with an op TABLE):
us per frame pipelined / with a synchronisation every frame.  For same-box A/Bs of where the table lives (FW_PARAM_BAR)."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
out = []
for n_em in (96, 256, 512, 1024):
    ps = ParticleSystem(seed=workloads.SEED)
    ems = workloads.many_emitters(n_em, 200)
    hs = [ps.spawn(ems[e][0], ems[e][1], uid=e) for e in range(n_em)]
    ps.update(dt)
    for _ in range(80): ps.step(dt)
    best = 1e9
    for rep in range(4):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(300): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    t0 = time.perf_counter()
    for _ in range(150): ps.step(dt); ps.synchronize()
    sync = (time.perf_counter() - t0) / 150 * 1e6
    out.append(f"{n_em} x 200 [{hs[-1].update_path(0)[0]}] {best:.2f} / {sync:.2f}")
    ps.close()
print("  ".join(out))
