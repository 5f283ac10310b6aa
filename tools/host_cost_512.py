import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads, sharding
from bevy_firework_amd.system import ParticleSystem
ps = ParticleSystem(seed=workloads.SEED)
ems = workloads.many_emitters(4096, 8192)
mine = sharding.local_indices(4096, 0, 8)
for e in mine:
    ps.spawn(ems[e][0], ems[e][1], uid=e)
dt = np.float32(1 / 60)
ps.update(dt)
for _ in range(80): ps.step(dt)
ps.synchronize()
for batch in (2, 4, 64):
    ps.synchronize(); t0 = time.perf_counter()
    for _ in range(batch): ps.step(dt)
    t1 = time.perf_counter(); ps.synchronize(); t2 = time.perf_counter()
    print(batch, "enqueue us/step %.1f total us/step %.1f" % ((t1 - t0) / batch * 1e6, (t2 - t0) / batch * 1e6))
