"""What ONE colliding small emitter costs a context of hundreds of small emitters: us per frame with / without it."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads, settings as S
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for n_em in (512, 2048):
    for with_coll in (0, 1, 8):
        ps = ParticleSystem(seed=workloads.SEED)
        ems = workloads.many_emitters(n_em, 200)
        hs = [ps.spawn(ems[e][0], ems[e][1], uid=e) for e in range(n_em)]
        if with_coll:
            sp, tf, world = workloads.example_collision() if hasattr(workloads, "example_collision") else (None, None, None)
            ps.set_colliders(world)
            for k in range(with_coll):
                hs.append(ps.spawn(sp, S.Transform((float(k), 2.0, 0.0)), uid=100000 + k))
        ps.update(dt)
        for _ in range(90): ps.step(dt)
        best = 1e9
        for rep in range(3):
            ps.synchronize(); t0 = time.perf_counter()
            for _ in range(200): ps.step(dt)
            ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        print(f"{n_em} x 200 + {with_coll} colliding small emitter(s) [{hs[-1].update_path(0)[0]}]: {best:.1f} us per frame", flush=True)
        ps.close()
