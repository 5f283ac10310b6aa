import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
ps = ParticleSystem(seed=workloads.SEED)
sp, tf = workloads.one_million()
h = ps.spawn(sp, tf, uid=0)
dt = np.float32(1 / 60)
ps.update(dt)
t0 = time.perf_counter()
for i in range(40000):
    ps.step(dt)
    if i % 10000 == 9999:
        print(i + 1, "frames, live", ps.live_count(), "%.1f us/step" % ((time.perf_counter() - t0) / (i + 1) * 1e6))
rng = np.random.default_rng(1)
for i in range(5000):
    ps.step(np.float32(1 / 60 + rng.uniform(-0.002, 0.002)))
print("variable dt: live", ps.live_count())
for i in range(2000):
    ps.step(dt)
print("back to fixed dt: live", ps.live_count(), h.counts())
