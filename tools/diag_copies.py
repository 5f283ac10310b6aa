import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
ps = ParticleSystem(seed=workloads.SEED)
for e, (sp, tf) in enumerate(workloads.many_emitters(int(sys.argv[1]), int(sys.argv[2]))):
    ps.spawn(sp, tf, uid=e)
dt = np.float32(1 / 60)
ps.update(dt)
for _ in range(90):
    ps.step(dt)
ps.synchronize()
print("MARK steady", flush=True)
for _ in range(30):
    ps.step(dt)
ps.synchronize()
ps.close()
