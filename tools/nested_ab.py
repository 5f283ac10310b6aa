import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1/60)
ps = ParticleSystem(seed=workloads.SEED)
sp, tf = workloads.nested(100000.0, 20.0)
h = ps.spawn(sp, tf, uid=0)
print("paths", h.update_path(0), h.update_path(1))
ps.update(dt)
for _ in range(250): ps.step(dt)
ps.synchronize()
for _ in range(8): ps.step(dt)  # (what the host does on SEEING the device's counts -- one reallocation of the smoke ring -- happens here)
ps.synchronize()
u0 = ps.updated_total(); t0 = time.perf_counter()
for _ in range(100): ps.step(dt)
ps.synchronize(); el = time.perf_counter() - t0
print("counts", h.counts(), "us/step", el/100*1e6, "particles/s", (ps.updated_total()-u0)/el)
ps.close()
