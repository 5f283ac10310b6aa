import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for n_em, per in ((2048, 200), (512, 2000)):
    ps = ParticleSystem(seed=workloads.SEED)
    ems = workloads.many_emitters(n_em, per)
    for e in range(n_em):
        ps.spawn(ems[e][0], ems[e][1], uid=e)
    ps.update(dt)
    for _ in range(80): ps.step(dt)
    ps.kernel_timing(True)
    for _ in range(200): ps.step(dt)
    ms, n, parts = ps.kernel_timing_read()
    ps.kernel_timing(False)
    print(n_em, "x", per, "kernel us", ms / n * 1e3, "particles/launch", parts / n)
    ps.close()
