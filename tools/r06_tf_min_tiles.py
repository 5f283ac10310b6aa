"""Round 6: from how many tiles on does the threshold forecast (fw_k_fc_resolve + the streaming schedule) beat the decoupled look-back
under a dt that never repeats?  fw_ctx::tf_min_tiles is the product's answer; this sweeps (emitters x live) with the scheme forced on
(FW_TF_MIN_TILES=0) and off (FW_TF=0) on the compacting path.   python tools/r06_tf_min_tiles.py   (GPU box)"""
import os, sys, time
os.environ["FW_ENABLE_KNOBS"] = "1"
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem

jit = [np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) for k in range(64)]
POINTS = ((1, 65536), (1, 262144), (1, 524288), (1, 786432), (1, 1000000), (1, 4000000), (16, 16384), (32, 16384), (48, 16384), (64, 16384), (256, 4096), (256, 16384), (256, 65536))


def run(n_em, live, env):
    for k in ("FW_TF", "FW_TF_MIN_TILES"):
        os.environ.pop(k, None)
    os.environ["FW_RANGE"] = "0"
    os.environ["FW_FIFO"] = "0"
    os.environ["FW_SMALL"] = "0"
    os.environ.update(env)
    with ParticleSystem(seed=workloads.SEED) as ps:
        for e, (s_, tf_) in enumerate(workloads.many_emitters(n_em, live)):
            ps.spawn(s_, tf_, uid=e)
        ps.update(jit[0])
        for k in range(96):
            ps.step(jit[k % 64])
        best = 1e9
        for rep in range(3):
            ps.synchronize(); t0 = time.perf_counter()
            for k in range(60):
                ps.step(jit[k % 64])
            ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 60 * 1e6)
        return best, ps.tf_frames()


print(f"{'emitters x live':>18s} {'tiles':>7s} {'look-back':>10s} {'threshold fc':>13s} {'product':>9s}   (us per frame, dt = 1/60 (1 + 0.1 sin 0.7k))")
for n_em, live in POINTS:
    lb, _ = run(n_em, live, {"FW_TF": "0"})
    tf, n = run(n_em, live, {"FW_TF_MIN_TILES": "0"})
    pr, npr = run(n_em, live, {})
    tiles = n_em * ((live + 1023) // 1024)
    print(f"{n_em:6d} x {live:9d} {tiles:7d} {lb:10.1f} {tf:13.1f} {pr:9.1f}   tf frames forced {n}, product {npr}", flush=True)
