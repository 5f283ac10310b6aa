"""Hundreds of small emitters next to ONE spawner with a Nested entry (examples/textures.rs): its frames run the separate spawn / nest
passes, so the small types' new particles are materialised by fw_k_spawn instead of being spawned by their own kernel.  us per frame."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads, settings as S
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for n_em, with_nested in ((0, 1), (512, 0), (512, 1), (2048, 0), (2048, 1)):
    ps = ParticleSystem(seed=workloads.SEED)
    ems = workloads.many_emitters(max(n_em, 1), 200)
    hs = [ps.spawn(ems[e][0], ems[e][1], uid=e) for e in range(n_em)]
    if with_nested:
        ex = workloads.example_textures()
        if len(ex) == 3: ps.set_colliders(ex[2])
        hs.append(ps.spawn(ex[0], ex[1], uid=100000))
    ps.update(dt)
    for _ in range(320): ps.step(dt)
    best = 1e9
    for rep in range(3):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(200): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    print(f"{n_em} x 200 + {with_nested} Nested spawner [{[hs[-1].update_path(t)[0] for t in range(len(hs[-1].counts()))]}]: {best:.1f} us per frame", flush=True)
    ps.close()
