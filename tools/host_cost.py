#!/usr/bin/env python3
"""Host-side cost of enqueueing one fw_step (small batches so the HW queue never fills)."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
ps = ParticleSystem(seed=1)
sp, tf = workloads.one_million()
ps.spawn(sp, tf)
dt = np.float32(1 / 60)
ps.update(dt)
for i in range(100):
    ps.step(dt)
ps.synchronize()
for batch in (4, 8, 16, 256, 1024):
    ps.synchronize()
    t0 = time.perf_counter()
    for i in range(batch):
        ps.step(dt)
    t1 = time.perf_counter()
    ps.synchronize()
    t2 = time.perf_counter()
    print(batch, "enqueue us/step %.2f" % ((t1 - t0) / batch * 1e6), "total us/step %.2f" % ((t2 - t0) / batch * 1e6))
