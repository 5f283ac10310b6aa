"""Hundreds of mid-size emitters: us per frame (pipelined) and the update launches alone (dispatch-attached events), for the cases of
tools/r04_range_min_sweep.py.  FW_WIDE_MAX=0: without the workgroup-per-type role of fw_k_update_small."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import os, sys, time
import numpy as np
if os.environ.get("FW_ATTACH"):
    import torch; torch.cuda.init()  # (before the library's own first HIP call)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
CASES = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("FW_CASES", "1024x600,1024x1000,512x1500,512x2000,256x3000,128x1000,256x1000").split(",")]
for n_em, per in CASES:
    ps = ParticleSystem(seed=workloads.SEED)
    ems = workloads.many_emitters(n_em, per)
    hs = [ps.spawn(ems[e][0], ems[e][1], uid=e) for e in range(n_em)]
    if os.environ.get("FW_ATTACH"):  # an instance buffer per emitter (what a renderer attaches): the update writes the render records
        keep = [torch.empty(int(per * 1.5) * 16, dtype=torch.float32, device="cuda") for _ in range(n_em)]
        for h, b in zip(hs, keep): h.attach_instances(b.data_ptr(), int(per * 1.5), particle_type=0)
    ps.update(dt)
    for _ in range(90): ps.step(dt)
    best = 1e9
    for rep in range(3):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(200): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    ps.kernel_timing(True)
    for _ in range(100): ps.step(dt)
    ev_ms, launches, _ = ps.kernel_timing_read()
    ps.kernel_timing(False)
    print(f"{n_em:5d} x {per:5d}  mode {hs[0].update_mode(0)}  {best:7.1f} us per frame   update launches {ev_ms * 1e3 / max(launches, 1):7.1f} us", flush=True)
    ps.close()
