"""Long runs of the round-4 ring paths against the ORACLE: (1) a Nested chain sparks -> smoke -> embers whose lifetimes are
RANGES (range rings: parents addressed through FwGlobals::rold, children counted by the device, cohort sizes through the pinned
report ring -- which wraps at 32768 frames), (2) the same chain with one-lifetime types (FIFO rings, parents spawned inside the
update kernel), (3) bouncing particles in a FIFO ring and in a range ring (COLL instantiations), all with a dt that jitters
and rings that wrap hundreds of times.  Any disagreement between the host's bookkeeping and the particles raises an internal
error inside the update kernel and surfaces as FW_EHIP at the next call."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("FW_RANGE_MIN", "0")
os.environ.setdefault("FW_FIFO_MIN", "0")
import oracle
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
from parity import assert_particles_match

rng = np.random.default_rng(11)
frames = int(os.environ.get("FW_SOAK_FRAMES", "40000"))


def chain(ranges):
    life = (lambda lo, hi: S.RandF32(lo, hi)) if ranges else (lambda lo, hi: S.RandF32.constant(lo))
    sparks = S.ParticleSettings(lifetime=life(0.5, 0.9), linear_drag=0.3, capacity=8192 if ranges else 0)
    smoke = S.ParticleSettings(lifetime=life(0.4, 0.7), acceleration=(0.0, 0.5, 0.0), linear_drag=0.7,
                               scale_curve=S.FireworkCurve.even_samples([1.0, 3.0]))
    embers = S.ParticleSettings(lifetime=life(0.2, 0.35), linear_drag=0.1)
    es = [S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(1500.0),
                             initial_velocity=S.RandVec3(S.RandF32(2.0, 5.0), (0.0, 1.0, 0.0), 0.0)),
          S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                             emission_pacing=S.EmissionPacing.CountOverDuration(6.0, 0.0, 0.0, 0.6), inherit_parent_velocity=False),
          S.EmissionSettings(particle_index=2, emission_mode=S.EmissionMode.Nested(1),
                             emission_pacing=S.EmissionPacing.CountOverDuration(3.0, 0.0, 0.1, 0.9),
                             initial_velocity=S.RandVec3(S.RandF32(0.0, 1.0), (0.0, -1.0, 0.0), 0.0))]
    return S.ParticleSpawner([sparks, smoke, embers], es)


def bouncer(ranges):
    ps = S.ParticleSettings(lifetime=S.RandF32(0.8, 1.4) if ranges else S.RandF32.constant(1.2), linear_drag=0.15,
                            collision_settings=S.ParticleCollisionSettings(0.6, 0.2, False))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(4000.0),
                            initial_velocity=S.RandVec3(S.RandF32(3.0, 8.0), (0.0, 1.0, 0.0), 0.0), inherit_parent_velocity=True)
    return S.ParticleSpawner([ps], [es])


_, _, world = workloads.stress_test_collision()
with ParticleSystem(seed=workloads.SEED) as ps:
    ps.set_colliders(world)
    specs = [(chain(True), S.Transform((0.0, 2.0, 0.0))), (chain(False), S.Transform((3.0, 2.0, 0.0))),
             (bouncer(True), S.Transform((5.0, 0.5, 0.0), (0.0, 0.0, 0.3826834, 0.9238795))),
             (bouncer(False), S.Transform((-3.0, 0.5, 1.0), (0.0, 0.0, -0.3826834, 0.9238795)))]
    pairs = []
    for k, (sp, tf) in enumerate(specs):
        g = ps.spawn(sp, tf, uid=70 + k)
        c = oracle.OracleSpawner(sp, seed=workloads.SEED, uid=70 + k, transform=tf)
        c.set_colliders(world)
        pairs.append((g, c, len(sp.particle_settings)))
    print("paths", [[g.update_path(t)[0] for t in range(n)] for g, _, n in pairs], flush=True)
    t0 = time.perf_counter()
    for i in range(frames):
        dt = np.float32(1 / 60 if (i // 3000) % 2 == 0 else rng.uniform(0.004, 0.03))
        ps.update(dt)
        for _, c, _ in pairs:
            c.step(dt)
        if i % 2000 == 1999 or i == frames - 1:
            for k, (g, c, n) in enumerate(pairs):
                assert g.counts() == c.counts(), (i, k, g.counts(), c.counts())
                for t in range(n):
                    assert_particles_match(g.particles(t), c.particles(t), True, f"frame {i} spawner {k} type {t}")
            print(i + 1, "frames ok, live", [g.counts() for g, _, _ in pairs],
                  "%.1f us/frame (incl. the oracle)" % ((time.perf_counter() - t0) / (i + 1) * 1e6), flush=True)
    print("paths", [[g.update_path(t)[0] for t in range(n)] for g, _, n in pairs])
print("SOAK-R04-OK")
