"""The reference's own examples at their own sizes (tens to hundreds of particles -- the regime the crate is written for): what one
frame costs on the HIP backend (enqueue + kernels, steady state, one context) and on the CPU oracle (one thread).  At these sizes a
frame is launch latency on the GPU and a few microseconds of arithmetic on the CPU: the figures say where the backend starts to pay."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1 / 60)
def sparks_at(rate):
    sp, tf = workloads.example_sparks()
    sp.emission_settings[0].emission_pacing = type(sp.emission_settings[0].emission_pacing).rate(rate)
    return sp, tf, []
cases = {
    "sparks (1000/s x 0.75 s)": lambda: workloads.example_sparks() + ([],),
    "pbr (150/s x 5 s)": lambda: workloads.example_pbr() + ([],),
    "collision (100/s x 6.75 s, slab + cube)": workloads.example_collision,
    "textures (12/s x 5 s + 6 puffs each, cylinder + cone)": workloads.example_textures,
    "stress_test (160k/s x 1 s)": lambda: workloads.stress_test() + ([],),
    "stress_test_collision (80k/s x 2 s)": workloads.stress_test_collision,
    "sparks at 10k/s": lambda: sparks_at(1e4), "sparks at 100k/s": lambda: sparks_at(1e5),
}
for name, make in cases.items():
    sp, tf, world = make()
    fill = int(max(p.lifetime.max for p in sp.particle_settings) * 60) + 20
    ps = ParticleSystem(seed=workloads.SEED)
    ps.set_colliders(world)
    h = ps.spawn(sp, tf, uid=0)
    ps.update(dt)
    for _ in range(fill): ps.step(dt)
    ps.synchronize()
    best = 1e9
    for rep in range(5):
        ps.synchronize(); t0 = time.perf_counter()
        for _ in range(300): ps.step(dt)
        ps.synchronize(); best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    # ... and a frame the host waits for (step + synchronize): what a game loop that reads results every frame sees
    t0 = time.perf_counter()
    for _ in range(200): ps.step(dt); ps.synchronize()
    sync_us = (time.perf_counter() - t0) / 200 * 1e6
    live = ps.live_count(); paths = [h.update_path(t)[0] for t in range(len(sp.particle_settings))]
    ps.close()
    o = oracle.OracleSpawner(sp, seed=workloads.SEED, uid=0, transform=tf); o.set_colliders(world)
    n_fill = fill if live < 50000 else fill
    for _ in range(n_fill): o.step(dt)
    n = 200 if live < 50000 else 10
    t0 = time.perf_counter()
    for _ in range(n): o.step(dt)
    cpu_us = (time.perf_counter() - t0) / n * 1e6
    print(json.dumps({"example": name, "live": live, "paths": paths, "gpu_us_per_frame_pipelined": round(best, 2),
                      "gpu_us_per_frame_synchronised": round(sync_us, 2), "cpu_oracle_us_per_frame_1_thread": round(cpu_us, 2)}), flush=True)
