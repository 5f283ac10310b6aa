#!/usr/bin/env python3
"""Where the non-temporal form of the in-place ring kernels starts to pay (fw_ctx::nt_bytes): us per frame with the plain and
the nt form over a grid of working-set sizes, range rings (64 emitters x n, lifetimes 0.8-1.2 s) and one FIFO ring
(configs[1] at k x the rate).  Run on the GPU box."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import json, os, sys, time
os.environ.setdefault("FW_RANGE_MIN", "0"), os.environ.setdefault("FW_FIFO_MIN", "0")  # (every size on its ring)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem

dt = np.float32(1 / 60)


WO = len(sys.argv) > 1 and sys.argv[1] == "wo"  # `nt_sweep.py wo`: plain against the form with only the write-only planes non-temporal


def measure(build, fill):
    out = {}
    for tag, mb in (("plain", "100000000"), ("nt", "0")):
        os.environ["FW_NT_MB"] = "100000000" if WO else mb
        os.environ["FW_NT_WO_MB"] = mb if WO else "100000000"
        ps = ParticleSystem(seed=workloads.SEED)
        build(ps)
        ps.update(dt)
        for _ in range(fill):
            ps.step(dt)
        ps.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(60):
                ps.step(dt)
            ps.synchronize()
            best = min(best, (time.perf_counter() - t0) / 60 * 1e6)
        h = next(iter(ps.spawners.values()))
        out["live"], out["path"], out[tag + "_us"] = ps.live_count(), h.update_path(0)[0], round(best, 1)
        out["MB_moved"] = round(out["live"] * h.update_path(0)[1] / 1e6)
        ps.close()
    out["nt_over_plain"] = round(out["nt_us"] / out["plain_us"], 3)
    return out


for live in ((1 << 13, 3 << 12, 1 << 14, 3 << 13, 1 << 15, 3 << 14, 1 << 16, 3 << 15, 1 << 17) if WO else (1 << 15, 3 << 14, 1 << 16, 3 << 15, 1 << 17, 3 << 16, 1 << 18)):
    def build(ps, live=live):
        for e, (sp, tf) in enumerate(workloads.many_emitters(64, live)):
            ps.spawn(sp, tf, uid=e)
    print(json.dumps({"workload": "64 emitters x %d" % live, **measure(build, 90)}), flush=True)
for k in ((0.5, 0.75, 1, 1.25, 1.5, 2, 3, 4, 6) if WO else (2, 3, 4, 6, 8, 12, 16)):
    def build(ps, k=k):
        sp, tf = workloads.one_million(rate=1e6 * k)
        ps.spawn(sp, tf, uid=0)
    print(json.dumps({"workload": "one ring, rate %ge6" % k, **measure(build, 70)}), flush=True)
