"""host phases of fw_step on many small emitters (ab build, FW_HOST_PROF): python tools/r04_host_prof.py [n_em per]"""
import os, sys, time
os.environ["FW_ENABLE_KNOBS"] = "1"
os.environ["FW_HOST_PROF"] = "100"
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ.setdefault("FW_LIB_PATH", os.path.join(root, "bevy_firework_amd/csrc/libfirework_hip_ab.so"))
import numpy as np
sys.path.insert(0, root)
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
n_em, per = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 200)
dt = np.float32(1 / 60)
ps = ParticleSystem(seed=workloads.SEED)
ems = workloads.many_emitters(n_em, per)
for e in range(n_em): ps.spawn(ems[e][0], ems[e][1], uid=e)
ps.update(dt)
for _ in range(100): ps.step(dt)
ps.synchronize()
t0 = time.perf_counter()
for _ in range(400): ps.step(dt)
ps.synchronize(); t2 = time.perf_counter()
print(n_em, "x", per, "us/step %.1f" % ((t2 - t0) / 400 * 1e6))
ps.close()
