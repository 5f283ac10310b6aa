"""Frame time against the number of 8192-particle emitters on range rings (is there a step where a residency round fills up?)."""
import os, sys, time
os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for n_em in (256, 320, 384, 416, 448, 480, 512, 544, 576, 640, 768):
    ps = ParticleSystem(seed=workloads.SEED)
    for e, (sp, tf) in enumerate(workloads.many_emitters(n_em, 8192)):
        ps.spawn(sp, tf, uid=e)
    ps.update(dt)
    for _ in range(100): ps.step(dt)
    ps.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(80): ps.step(dt)
        ps.synchronize()
        best = min(best, (time.perf_counter() - t0) / 80 * 1e6)
    print("%4d emitters x 8192: %6.1f us per frame, %5.1f ns per emitter, live %d" % (n_em, best, best * 1000 / n_em, ps.live_count()), flush=True)
    ps.close()
