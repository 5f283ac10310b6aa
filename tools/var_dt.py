"""configs[1] stepped with a jittering dt (what Bevy's `Update` schedule delivers): per-step time and update-kernel time.
   python tools/var_dt.py [frames]"""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import workloads  # noqa: E402
from bevy_firework_amd.system import ParticleSystem  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ps = ParticleSystem(seed=workloads.SEED)
sp, tf = workloads.one_million()
ps.spawn(sp, tf, uid=0)
jit = [np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) for k in range(64)]
ps.update(jit[0])
for k in range(100):
    ps.step(jit[k % 64])
ps.synchronize()
t0 = time.perf_counter()
for k in range(frames):
    ps.step(jit[k % 64])
ps.synchronize()
el = time.perf_counter() - t0
ps.kernel_timing(True)
for k in range(frames):
    ps.step(jit[k % 64])
ms, n, parts = ps.kernel_timing_read()
print(f"variable dt: {el / frames * 1e6:.2f} us/step, update kernel {ms / n * 1e3:.2f} us, {parts / n:.0f} particles/launch, "
      f"live {ps.live_count()}")
