"""Where does a ring of its own pay in a MIXED context?  One general-path segment (lifetime range) of 500k particles plus
one single-lifetime type of X particles: frame time with the second type on a ring (its own launch) and on the general
path (one launch for both).  Sets the default of FW_FIFO_MIN."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from bevy_firework_amd import settings as S, workloads
from bevy_firework_amd.system import ParticleSystem
X = int(sys.argv[1]); dt = np.float32(1/60)
with ParticleSystem(seed=1) as ps:
    a = S.ParticleSettings(lifetime=S.RandF32(0.8, 1.2), linear_drag=0.1)
    ps.spawn(S.ParticleSpawner([a], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(500000.0))]), uid=0)
    b = S.ParticleSettings(lifetime=S.RandF32.constant(1.0), linear_drag=0.1)
    h = ps.spawn(S.ParticleSpawner([b], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(float(X)))]), uid=1)
    ps.update(dt)
    for _ in range(100): ps.step(dt)
    ps.synchronize(); t0 = time.perf_counter()
    for _ in range(300): ps.step(dt)
    ps.synchronize()
    print(h.update_path(0)[0], "%%.1f" %% ((time.perf_counter() - t0) / 300 * 1e6))
''' % ROOT
for X in (8000, 32000, 131072, 524288, 2000000):
    row = []
    for env in ({"FW_FIFO": "1", "FW_FIFO_MIN": "0"}, {"FW_FIFO": "0"}, {}):
        out = subprocess.run([sys.executable, "-c", CODE, str(X)], env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip()
        row.append(out)
    print(f"X = {X:8d}: ring {row[0]} us/frame   general {row[1]} us/frame   default settings: {row[2]}")
