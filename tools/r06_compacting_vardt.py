"""Round 6: configs[2] on the COMPACTING path (FW_RANGE=0) stepped with a dt that never repeats -- the frame an
unmodified Bevy `Update` schedule (plugin.rs:26-31) gives every large type left on that path.  us per frame (pipelined) and the update
launches alone, for the schedules the library has: decoupled look-back in one launch (the product's choice without a forecast),
FW_UPDATE_MODE=split (count -> scan -> update: three launches, no inter-workgroup traffic), and the fixed-dt forecast schedule as the
yardstick.   python tools/r06_compacting_vardt.py [emitters live]   (GPU box)"""
import os, sys, time
os.environ["FW_ENABLE_KNOBS"] = "1"
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem

n_em = int(sys.argv[1]) if len(sys.argv) > 1 else 256
live = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
jit = [np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) for k in range(64)]
fixed = [np.float32(1.0 / 60.0)] * 64
for name, env, dts in (("forecast (fixed dt)", {}, fixed),
                       ("threshold forecast: resolve + stream (variable dt; the product)", {}, jit),
                       ("decoupled look-back, FW_TF=0 (variable dt; rounds 1-5)", {"FW_TF": "0"}, jit),
                       ("split: count -> scan -> update (variable dt)", {"FW_UPDATE_MODE": "split", "FW_TF": "0"}, jit)):
    for k in ("FW_UPDATE_MODE", "FW_STATIC_NEW", "FW_TF"):
        os.environ.pop(k, None)
    os.environ["FW_RANGE"] = "0"
    os.environ.update(env)
    with ParticleSystem(seed=workloads.SEED) as ps:
        for e, (s_, tf_) in enumerate(workloads.many_emitters(n_em, live)):
            ps.spawn(s_, tf_, uid=e)
        ps.update(dts[0])
        for k in range(96):
            ps.step(dts[k % 64])
        best = 1e9
        for rep in range(3):
            ps.synchronize(); t0 = time.perf_counter()
            for k in range(60):
                ps.step(dts[k % 64])
            ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 60 * 1e6)
        ps.kernel_timing(True)
        for k in range(60):
            ps.step(dts[k % 64])
        ev_ms, launches, parts = ps.kernel_timing_read()
        ps.kernel_timing(False)
        h = next(iter(ps.spawners.values()))
        mode, moved, algo = h.update_path(0)
        k_us = ev_ms * 1e3 / max(launches, 1)
        print(f"{name:68s} tf frames {ps.tf_frames():4d}  {best:7.1f} us per frame   update launches {k_us:7.1f} us   path {mode} {moved} B moved / {algo} B algorithmic"
              f"   {parts / max(launches, 1) * algo / (k_us * 1e-6) / 8e12:.3f} of 8 TB/s", flush=True)
