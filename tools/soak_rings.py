"""Long run of the ring paths against closed-form expectations (no oracle: too slow for 10^5 frames): configs[1] and the
Nested configs[3] shape, fixed and jittering dt.  Any disagreement between the host's cohort bookkeeping and the particles
raises FW_ERR_FORECAST inside the update kernel and surfaces as an exception at the next count read."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem

dt = np.float32(1 / 60)
rng = np.random.default_rng(1)
with ParticleSystem(seed=workloads.SEED) as ps:
    h = ps.spawn(*workloads.one_million(), uid=0)
    print("configs[1] path", h.update_path(0))
    ps.update(dt)
    t0 = time.perf_counter()
    for i in range(60000):
        ps.step(dt if (i // 5000) % 2 == 0 else np.float32(1 / 60 + rng.uniform(-0.004, 0.004)))
        if i % 10000 == 9999:
            n = ps.live_count()
            print(i + 1, "frames, live", n, "%.1f us/step" % ((time.perf_counter() - t0) / (i + 1) * 1e6))
            assert 900000 < n < 1100000
    p = h.particles(0)
    assert np.all(np.diff(p["age"]) <= 0) and np.all(p["age"] < p["lifetime"]) and np.isfinite(p["position"]).all()
with ParticleSystem(seed=workloads.SEED) as ps:
    h = ps.spawn(*workloads.nested(20000.0, 20.0), uid=0)
    print("nested paths", h.update_path(0), h.update_path(1))
    ps.update(dt)
    t0 = time.perf_counter()
    for i in range(int(os.environ.get("FW_SOAK_NESTED_FRAMES", "20000"))):  # (the children-count report ring wraps at 32768 frames)
        ps.step(dt if (i // 2500) % 2 == 0 else np.float32(1 / 60 + rng.uniform(-0.004, 0.004)))
        if i % 5000 == 4999:
            c = h.counts()
            print(i + 1, "frames, counts", c, "%.1f us/step" % ((time.perf_counter() - t0) / (i + 1) * 1e6))
            assert 35000 < c[0] < 45000 and 600000 < c[1] < 900000
    for t in (0, 1):
        p = h.particles(t)
        assert np.all(np.diff(p["age"]) <= 0) and np.all(p["age"] < p["lifetime"]) and np.isfinite(p["position"]).all()
print("ok")
