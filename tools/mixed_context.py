"""A big ring next to a few small compacting segments: what does the second launch cost?  configs[1]'s emitter plus K
small emitters with a lifetime range (1000 particles each); frame time with the big type on a ring / on the general path."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from bevy_firework_amd import settings as S, workloads
from bevy_firework_amd.system import ParticleSystem
K = int(sys.argv[1]); dt = np.float32(1/60)
with ParticleSystem(seed=1) as ps:
    h = ps.spawn(*workloads.one_million(), uid=0)
    for k in range(K):
        a = S.ParticleSettings(lifetime=S.RandF32(0.8, 1.2), linear_drag=0.1)
        ps.spawn(S.ParticleSpawner([a], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(1000.0))]), uid=1 + k)
    ps.update(dt)
    for _ in range(100): ps.step(dt)
    ps.synchronize(); t0 = time.perf_counter()
    for _ in range(400): ps.step(dt)
    ps.synchronize()
    print(h.update_path(0)[0], "%%.1f" %% ((time.perf_counter() - t0) / 400 * 1e6))
''' % ROOT
for K in (0, 1, 10, 100):
    row = []
    for env in ({}, {"FW_FIFO": "0"}):
        row.append(subprocess.run([sys.executable, "-c", CODE, str(K)], env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip())
    print(f"1M + {K:3d} small: default {row[0]} us/frame   FW_FIFO=0 {row[1]} us/frame")
