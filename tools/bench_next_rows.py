#!/usr/bin/env python3
"""Throughput of the rows SURVEY.md §8(f) ranks next: instance packing (render.rs:403) and the AABB reduction
(render.rs:677-703), on the 1M-particle configuration.  Run on the GPU box."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import workloads  # noqa: E402
from bevy_firework_amd.system import ParticleSystem  # noqa: E402

dt = np.float32(1 / 60)
for name, rate in (("1M live", 1.0e6), ("8M live", 8.0e6)):
    ps = ParticleSystem(seed=workloads.SEED)
    sp, tf = workloads.one_million(rate=rate)
    h = ps.spawn(sp, tf, uid=0)
    ps.update(dt)
    for _ in range(70):
        ps.step(dt)
    ps.synchronize()
    live = ps.live_count()
    out = torch.empty(int(live * 1.1) * 64, dtype=torch.uint8, device="cuda")
    L = ps._lib
    ub = C.c_uint64()
    L.fw_spawner_pack_instances_device.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    for reps in (5, 50):
        ps.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            assert L.fw_spawner_pack_instances_device(ps._ctx, h.handle, 0, C.c_void_p(out.data_ptr()), out.numel() // 64, C.byref(ub)) == 0
        ps.synchronize()
        t_pack = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(20):
        h.aabb()
    t_aabb = (time.perf_counter() - t0) / 20
    # the same query with the boxes fused into the update (fw_ctx_track_aabbs): one frame to produce them, then fold
    ps.track_aabbs(True)
    ps.step(dt)
    ps.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        h.aabb()
    t_aabb_fused = (time.perf_counter() - t0) / 20
    ps.track_aabbs(False)
    # a frame as the renderer sees it: step + pack
    ps.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        ps.step(dt)
        L.fw_spawner_pack_instances_device(ps._ctx, h.handle, 0, C.c_void_p(out.data_ptr()), out.numel() // 64, C.byref(ub))
    ps.synchronize()
    t_frame = (time.perf_counter() - t0) / 100
    # the same frame with the records written by the update kernel itself
    h.attach_instances(out.data_ptr(), out.numel() // 64)
    for _ in range(20):
        ps.step(dt)
    ps.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ps.step(dt)
    ps.synchronize()
    t_fused = (time.perf_counter() - t0) / 200
    moved_attached = h.update_path(0)[1] + 64  # bytes the update moves per particle with the records attached
    h.attach_instances(0, 0)
    print(json.dumps({"config": name, "live": live, "step_with_attached_instances_us": t_fused * 1e6,
                      "attached_bytes_per_particle": moved_attached, "fused_GBps": live * moved_attached / t_fused / 1e9,
                      "pack_us": t_pack * 1e6, "pack_GBps_132B": live * 132 / t_pack / 1e9,
                      "aabb_call_us (incl. count readback + sync)": t_aabb * 1e6,
                      "aabb_call_us with boxes fused into the update": t_aabb_fused * 1e6,
                      "step_plus_pack_us": t_frame * 1e6}))
    ps.close()
