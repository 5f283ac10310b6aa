import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for n_em, per in ((2048, 200), (512, 2000), (64, 16000)):
    ps = ParticleSystem(seed=workloads.SEED)
    ems = workloads.many_emitters(n_em, per)
    for e in range(n_em):
        ps.spawn(ems[e][0], ems[e][1], uid=e)
    ps.update(dt)
    for _ in range(80): ps.step(dt)
    ps.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): ps.step(dt)
    t1 = time.perf_counter(); ps.synchronize(); t2 = time.perf_counter()
    live = ps.live_count()
    print(n_em, "emitters x", per, "live", live, "us/step %.1f (host submit %.1f)" % ((t2 - t0) / 200 * 1e6, (t1 - t0) / 200 * 1e6),
          "particles/s %.2e" % (live / ((t2 - t0) / 200)))
    ps.close()
