"""Where does a SHORT timed region (the driver's --steps 20) lose time against a long one?  Times run(K) + sync for several K,
with three ways of waiting for the end, on the headline workload.  python tools/r04_short_run.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
stream = torch.cuda.Stream()
ps = ParticleSystem(device=0, seed=workloads.SEED, stream=stream.cuda_stream)
sp, tf = workloads.one_million()
ps.spawn(sp, tf, uid=0)
ps.update(dt)
for _ in range(70): ps.step(dt)
torch.cuda.synchronize()

def timed(K, how, per_step=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if per_step is not None:
        for i in range(K):
            ps.step(dt); per_step.append(time.perf_counter() - t0)
    else:
        for _ in range(K): ps.step(dt)
    t1 = time.perf_counter()
    if how == "fw": ps.synchronize()
    elif how == "query":
        while not stream.query(): pass
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e6, (t1 - t0) * 1e6

for how in ("torch", "fw", "query"):
    for K in (1, 5, 20, 100, 600):
        rs = [timed(K, how) for _ in range(9)]
        tot = sorted(r[0] for r in rs)[4]; sub = sorted(r[1] for r in rs)[4]
        print(f"{how:6s} K={K:4d}  total {tot:9.1f} us  = {tot / K:7.2f} us/step   host submit {sub:8.1f} us  ({sub / K:6.2f}/step)")
ps_ = []
timed(20, "torch", ps_)
print("submit-return times of 20 steps (us):", " ".join(f"{x * 1e6:.0f}" for x in ps_))
