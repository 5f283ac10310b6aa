import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
ps = ParticleSystem(seed=workloads.SEED)
sp, tf = workloads.nested(100000.0, 20.0)
ps.spawn(sp, tf, uid=0)
dt = np.float32(1 / 60)
ps.update(dt)
for _ in range(250):
    ps.step(dt)
ps.synchronize()
for _ in range(8):  # (the one reallocation of the smoke ring, once the host has seen the counts, lands here)
    ps.step(dt)
ps.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    ps.step(dt)
ps.synchronize()
print("us/step", (time.perf_counter() - t0) / 100 * 1e6, "live", ps.live_count())
