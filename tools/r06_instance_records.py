"""Round 6: the frame a RENDERER asks for -- update + the 64-byte ParticleInstance record of every survivor
(render.rs:95-115, 403) written by the update kernel into an attached device buffer -- for configs[1] (one FIFO ring, plain attach) and
configs[2] (256 range rings, windowed attach: the rings stay rings).  The same code bench.py runs under roofline.with_instance_records;
under `rocprofv3 --kernel-trace --stats` its kernel_stats.csv holds the average duration of fw_k_update_fifo<true, ..> /
fw_k_update_range<.., true, ..>.   python tools/r06_instance_records.py [c1] [c2]   (GPU box)"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem

dt = np.float32(1 / 60)
which = sys.argv[1:] or ["c1", "c2"]
cases = []
if "c1" in which:
    cases.append(("configs[1] + instance records (fw_spawner_attach_instances)", [workloads.one_million()], 82, 300, 1 << 20, False))
if "c2" in which:
    cases.append(("configs[2] + instance records (fw_spawner_attach_instances_window)", workloads.many_emitters(256, 65536), 96, 100, 110000, True))
for label, ems, fill, frames, cap, window in cases:
    with ParticleSystem(seed=workloads.SEED) as ps:
        bufs = []
        for e, (s_, tf_) in enumerate(ems):
            h = ps.spawn(s_, tf_, uid=e)
            b = torch.empty(cap * 64, dtype=torch.uint8, device="cuda")
            bufs.append(b)
            (h.attach_instances_window if window else h.attach_instances)(b.data_ptr(), cap, particle_type=0)
        ps.update(dt)
        for _ in range(fill):
            ps.step(dt)
        ps.synchronize()
        u0 = ps.updated_total(); t0 = time.perf_counter()
        for _ in range(frames):
            ps.step(dt)
        ps.synchronize()
        el = time.perf_counter() - t0
        upd = ps.updated_total() - u0
        ps.kernel_timing(True)
        for _ in range(frames):
            ps.step(dt)
        ev_ms, launches, parts = ps.kernel_timing_read()
        ps.kernel_timing(False)
        h0 = next(iter(ps.spawners.values()))
        mode, moved, algo = h0.update_path(0)
        k_us = ev_ms * 1e3 / max(launches, 1)
        print(json.dumps({"config": label, "live": ps.live_count(), "us_per_step": el / frames * 1e6, "particles_per_s": upd / el,
                          "update_path": mode, "moved_bytes_per_particle": moved, "algorithmic_bytes_per_particle": algo,
                          "update_kernel_us": k_us, "algorithmic_GBps": parts / max(launches, 1) * algo / (k_us * 1e-6) / 1e9,
                          "frac_of_8TBps": parts / max(launches, 1) * algo / (k_us * 1e-6) / 8e12}), flush=True)
        del bufs
