"""Does the headline launch spill into a second round of workgroups?  fw_k_update_fifo at configs[1] is ~977 ring tiles + ~66
workgroups of new particles = ~1043 workgroups for 1024 slots (4 workgroups per CU x 256 CUs).  Sweep the emitter's rate across
the point where tiles + new-particle workgroups pass 1024 and print the frame time per particle."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for rate in (0.80e6, 0.90e6, 0.94e6, 0.96e6, 0.97e6, 0.98e6, 0.99e6, 1.0e6, 1.02e6, 1.05e6, 1.10e6):
    ps = ParticleSystem(seed=workloads.SEED)
    sp, tf = workloads.one_million(rate=rate)
    h = ps.spawn(sp, tf, uid=0)
    ps.update(dt)
    for _ in range(70): ps.step(dt)
    ps.synchronize()
    live = ps.live_count()
    best = 1e9
    for rep in range(5):
        for _ in range(20): ps.step(dt)
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(300): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    n_in = live + rate / 60.0
    wgs = int(np.ceil(n_in / 1024)) + 1 + int(np.ceil(rate / 60.0 / 256))
    print(json.dumps({"rate": rate, "live": live, "path": h.update_path(0)[0], "workgroups_about": wgs, "frame_us": round(best, 2),
                      "ps_per_particle": round(best * 1e6 / n_in, 3)}))
    ps.close()
