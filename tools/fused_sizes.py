import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
for rate in (1e6, 2e6, 4e6, 8e6):
    ps = ParticleSystem(seed=workloads.SEED)
    sp, tf = workloads.one_million(rate=rate)
    h = ps.spawn(sp, tf, uid=0)
    ps.update(dt)
    for _ in range(70): ps.step(dt)
    ps.synchronize()
    live = ps.live_count()
    out = torch.empty(int(live * 1.1) * 64, dtype=torch.uint8, device="cuda")
    def run(n=100):
        for _ in range(10): ps.step(dt)
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(n): ps.step(dt)
        ps.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    t_plain = run()
    h.attach_instances(out.data_ptr(), 1)      # INST kernels, one record only
    t_variant = run()
    h.attach_instances(out.data_ptr(), out.numel() // 64)
    t_fused = run()
    h.attach_instances(0, 0)
    print(json.dumps({"live": live, "plain_us": round(t_plain, 1), "inst_variant_cap1_us": round(t_variant, 1), "fused_us": round(t_fused, 1),
                      "plain_TBps_164": round(live * 164 / t_plain / 1e6, 2), "fused_TBps_228": round(live * 228 / t_fused / 1e6, 2)}))
    ps.close()
