"""configs[3] (Nested, both types on rings) in chunks: us per frame with and without a host synchronisation between the chunks
(a free-running host shows where fw_step itself has to wait for the device)."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
dt = np.float32(1/60)
for sync_chunks in (True, False):
    ps = ParticleSystem(seed=workloads.SEED)
    sp, tf = workloads.nested(100000.0, 20.0)
    h = ps.spawn(sp, tf, uid=0)
    ps.update(dt)
    out = []
    T0 = time.perf_counter()
    for c in range(16):
        if sync_chunks: ps.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): ps.step(dt)
        if sync_chunks: ps.synchronize()
        out.append((time.perf_counter() - t0) / 50 * 1e6)
    ps.synchronize()
    print("sync between chunks" if sync_chunks else "free-running host", " ".join("%.0f" % x for x in out), "| whole run %.1f us/frame" % ((time.perf_counter() - T0) / 800 * 1e6))
    ps.close()
