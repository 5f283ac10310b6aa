"""Containers the reference does not bound (Vec<ParticleSettings> / Vec<EmissionSettings>, core.rs:178-185; curve sample
vectors, curve.rs:40-75; the collider world, core.rs:756-765) are not bounded here either: more than eight particle types
and emission entries per spawner, more ring types than one FIFO launch has records for, gradients far longer than the LDS
staging area, a hundred colliders that move every frame.  Everything against the oracle.  Needs an MI355X."""
import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair

pytestmark = pytest.mark.gpu
DT = np.float32(1.0 / 60.0)
SEED = workloads.SEED


@pytest.fixture()
def system(fw_path):
    from bevy_firework_amd.system import ParticleSystem

    with ParticleSystem(device=0, seed=SEED) as ps:
        ps.path = fw_path
        yield ps


def test_eleven_particle_types_and_fourteen_emission_entries(system):
    """one spawner: 11 types (lifetime ranges and single values, spinning and not), 13 Global entries and a Nested one on the
    tenth type; the AABB query folds them in chunks of eight"""
    rng = np.random.default_rng(8)
    types, emissions = [], []
    for t in range(11):
        lo = float(rng.uniform(0.2, 0.6))
        types.append(S.ParticleSettings(
            lifetime=S.RandF32.constant(lo) if t % 3 == 0 else S.RandF32(lo, lo + float(rng.uniform(0.1, 0.5))),
            initial_scale=S.RandF32(0.02, 0.2), linear_drag=float(rng.uniform(0.0, 0.5)),
            scale_curve=S.FireworkCurve.even_samples([1.0, float(rng.uniform(0.5, 2.0))]),
            angular_acceleration=(0.0, 0.3, 0.0) if t % 4 == 1 else (0.0, 0.0, 0.0)))
    for i in range(13):
        d = rng.normal(size=3)
        emissions.append(S.EmissionSettings(
            particle_index=i % 10, emission_pacing=S.EmissionPacing.rate(float(rng.uniform(300.0, 2500.0))),
            initial_velocity=S.RandVec3(S.RandF32(0.5, 4.0), tuple(float(c) for c in d / np.linalg.norm(d)), 0.0)))
    emissions.append(S.EmissionSettings(particle_index=10, emission_mode=S.EmissionMode.Nested(9),
                                        emission_pacing=S.EmissionPacing.CountOverDuration(6.0, 1.0, 0.0, 0.8),
                                        initial_velocity=S.RandVec3(S.RandF32(0.0, 1.0), (0.0, -1.0, 0.0), 0.0)))
    pair = Pair(system, S.ParticleSpawner(types, emissions), S.Transform((0.0, 1.0, 0.0)), seed=SEED, uid=81)
    for fr in range(80):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 10 == 9:
            pair.check(what=f"frame {fr}")  # (some types spin: the rotation goes through sin / cos, tests/parity.py)
            any_, mn, mx = pair.gpu.aabb()
            parts = [pair.gpu.particles(t) for t in range(11)]  # (the box of the state the query saw)
            lo = np.min([(p["position"] - p["scale"][:, None]).min(axis=0) for p in parts if len(p)], axis=0)
            hi = np.max([(p["position"] + p["scale"][:, None]).max(axis=0) for p in parts if len(p)], axis=0)
            assert any_ and np.array_equal(mn, lo) and np.array_equal(mx, hi), fr
    c = pair.gpu.counts()
    assert len(c) == 11 and all(n > 100 for n in c), c
    assert np.array_equal(pair.gpu.last_emitted(9, 13), pair.cpu.last_emitted(9, 13))


def test_more_ring_types_than_one_fifo_launch_holds(monkeypatch):
    """twelve one-lifetime types in a context.  Eight are eight FIFO rings (their records ride in one launch's arguments); the
    ninth takes a range ring and -- round 5, fw_ctx::n_spilled -- the eight FIFO rings, full of particles by then, become range
    rings where they stand (fifo_to_range: nothing copied), so that a frame stays ONE kind of launch; none falls back to the
    compacting path.  (The same at product thresholds: tests/test_gpu_range.py, tests/test_gpu_lifecycle.py.)"""
    from bevy_firework_amd.system import ParticleSystem

    monkeypatch.setenv("FW_FIFO", "1"), monkeypatch.setenv("FW_FIFO_MIN", "0")
    monkeypatch.setenv("FW_RANGE", "1"), monkeypatch.setenv("FW_RANGE_MIN", "0")
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = []

        def add(k):
            ps = S.ParticleSettings(lifetime=S.RandF32.constant(0.3 + 0.02 * k), linear_drag=0.2,
                                    base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
            es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(4000.0 + 500.0 * k),
                                    initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0))
            pairs.append(Pair(system, S.ParticleSpawner([ps], [es]), S.Transform((float(k), 0.0, 0.0)), seed=SEED, uid=300 + k))

        def run(n, what):
            for fr in range(n):
                system.update(DT)
                for p in pairs:
                    p.step_cpu(DT)
                if fr % 10 == 9 or fr == n - 1:
                    for k, p in enumerate(pairs):
                        p.check(exact_all=True, what=f"{what}, frame {fr} spawner {k}")

        for k in range(8):
            add(k)
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["fifo"] * 8
        run(27, "eight FIFO rings")  # (lifetimes 0.30-0.44 s: the first rings lose particles by now, the last are a few frames from it)
        add(8)
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["range"] * 9
        for k, p in enumerate(pairs):
            p.check(exact_all=True, what=f"right after the ninth type, spawner {k}")
        run(6, "nine range rings")
        for k in range(9, 12):
            add(k)
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["range"] * 12
        run(50, "twelve range rings")
        assert all(p.gpu.count(0) > 1000 for p in pairs)


@pytest.mark.parametrize("defaults", [False, True])
def test_ten_global_emitters_feed_one_constant_lifetime_type(system, monkeypatch, defaults):
    """more Global entries on ONE one-lifetime type than a FIFO launch carries spawn ops for (FW_INLINE_OPS = 8): the type
    must not become a FIFO ring (its ops would overflow the launch's argument block) -- it takes the range ring or the
    compacting path; with the product's default thresholds too (capacity >= 32768 is what made it a FIFO ring before)"""
    if defaults:
        if system.path != "fifo":
            pytest.skip("one run with the product's defaults is enough")
        for k in ("FW_FIFO", "FW_FIFO_MIN", "FW_RANGE", "FW_RANGE_MIN"):
            monkeypatch.delenv(k, raising=False)
    from bevy_firework_amd.system import ParticleSystem

    ps = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), linear_drag=0.1,
                            base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
    es = [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(9000.0 + 700.0 * k),
                             initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0)) for k in range(10)]
    with ParticleSystem(device=0, seed=SEED) as fresh:  # (the knobs are read when a context is created)
        pair = Pair(fresh, S.ParticleSpawner([ps], es), S.Transform((0.0, 0.5, 0.0)), seed=SEED, uid=77)
        path = pair.gpu.update_path(0)[0]
        assert path != "fifo", path
        if defaults:
            assert path == "range", path  # ~58k live particles: well above the range threshold
        for fr in range(50):
            fresh.update(DT)
            pair.step_cpu(DT)
            if fr % 10 == 9:
                pair.check(exact_all=True, what=f"frame {fr}")
        assert pair.gpu.counts()[0] > 40000


def test_gradients_longer_than_the_staging_area(system):
    """a 33-key uneven gradient (still staged in LDS), and a type with a 200-key base colour, a 150-key emissive colour and a
    300-key scale curve (2350 floats: read from device memory by the feature kernels) next to an ordinary type"""
    rng = np.random.default_rng(21)

    def grad(n, uneven):
        cols = [tuple(float(c) for c in rng.uniform(0.0, 4.0, size=4)) for _ in range(n)]
        if not uneven:
            return S.FireworkGradient.even_samples(cols)
        ts = np.sort(rng.uniform(0.0, 1.0, size=n)).astype(np.float32)
        ts[0], ts[-1] = 0.0, 1.0
        return S.FireworkGradient.uneven_samples([(float(t), c) for t, c in zip(ts, cols)])

    medium = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.7), base_color=grad(33, True), particles_destroyed=lambda dead: None)
    long_ = S.ParticleSettings(lifetime=S.RandF32(0.2, 0.6), base_color=grad(200, True), emissive_color=grad(150, False),
                               scale_curve=S.FireworkCurve.even_samples([float(x) for x in rng.uniform(0.5, 2.0, size=300)]),
                               particles_destroyed=lambda dead: None)
    plain = S.ParticleSettings(lifetime=S.RandF32.constant(0.4))
    es = [S.EmissionSettings(particle_index=t, emission_pacing=S.EmissionPacing.rate(6000.0),
                             initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0)) for t in range(3)]
    pair = Pair(system, S.ParticleSpawner([medium, long_, plain], es), seed=SEED, uid=91)
    from parity import assert_particles_match
    for fr in range(70):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 7 == 6:
            pair.check(exact_all=True, what=f"frame {fr}")
            for t in (0, 1):
                assert_particles_match(pair.gpu.destroyed(t), pair.cpu.destroyed(t), True, f"destroyed type {t} frame {fr}")
    assert all(n > 1500 for n in pair.gpu.counts())
    assert len(np.unique(pair.gpu.particles(1)["base_color"], axis=0)) > 500  # the long gradient really was sampled


def test_a_hundred_colliders_that_move_every_frame(system):
    """the collider world is replaced before every frame (the reference queries the live physics world each frame,
    core.rs:756-765) without ever waiting for the frames in flight; 100 colliders; bit for bit against the oracle"""
    rng = np.random.default_rng(33)
    cs = S.ParticleCollisionSettings(0.6, 0.2, False, 0xFFFFFFFF)
    ps = S.ParticleSettings(lifetime=S.RandF32(0.8, 1.6), linear_drag=0.1, collision_settings=cs)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(5000.0),
                            initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (0.3, -0.9, 0.1), 0.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), S.Transform((0.0, 3.0, 0.0)), seed=SEED, uid=95)
    spheres = [S.Collider.Sphere(tuple(float(c) for c in rng.uniform(-4.0, 4.0, size=3)), float(rng.uniform(0.2, 0.6))) for _ in range(99)]
    for fr in range(120):
        floor = S.Collider.Plane((0.0, float(np.float32(0.3 * np.sin(0.2 * fr))), 0.0), (0.0, 1.0, 0.0))  # a floor that moves
        world = [floor] + spheres[: 99 if fr % 2 else 60]  # ... and a world whose size changes
        system.set_colliders(world)
        pair.cpu.set_colliders(world)
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 12 == 11:
            pair.check(exact_all=True, what=f"frame {fr}")
    assert pair.gpu.count(0) > 4000


def test_batched_parent_velocities_modifiers_and_queues(system):
    """fw_ctx_set_parent_velocities / fw_ctx_set_modifiers / fw_ctx_queue (ABI 5: what sync_parent_velocity, core.rs:706-736,
    propagate_particle_spawner_modifier, core.rs:690-703, and a gameplay system write for whole sets of spawners every frame):
    the same effect as one call per spawner; one invalid handle -> FW_EINVAL and nothing changes"""
    import ctypes as C

    from bevy_firework_amd import _ffi

    rng = np.random.default_rng(12)
    pairs = []
    for k in range(6):
        sp, _ = workloads.stress_test(rate=3000.0)
        if k % 2:
            sp.emission_settings[0].emission_pacing = S.EmissionPacing.OnDemand()
        pairs.append(Pair(system, sp, S.Transform((float(k), 0.1, 0.0)), seed=SEED, uid=520 + k))
    gpus = [p.gpu for p in pairs]
    for fr in range(30):
        vs = [tuple(float(np.float32(c)) for c in rng.uniform(-2.0, 2.0, size=3)) for _ in pairs]
        ms = [S.EffectModifier(float(np.float32(rng.uniform(0.5, 2.0))), float(np.float32(rng.uniform(0.5, 2.0)))) for _ in pairs]
        qs = [int(rng.integers(0, 300)) if k % 2 else 0 for k in range(len(pairs))]
        system.set_parent_velocities(gpus, vs)
        system.set_modifiers(gpus, ms)
        system.queue_particles(gpus, qs)
        for p, v, m, q in zip(pairs, vs, ms, qs):
            p.cpu.set_parent_velocity(v), p.cpu.set_modifier(m), p.cpu.queue_particles(q)
        system.update(DT)
        for p in pairs:
            p.step_cpu(DT)
    lib, ctx = system._lib, system._ctx
    n = len(pairs)
    handles = (C.c_int32 * n)(*[p.gpu.handle for p in pairs])
    handles[4] = 9999  # not a spawner
    big3, big1, cnt = (C.c_float * (3 * n))(*([50.0] * (3 * n))), (C.c_float * n)(*([9.0] * n)), (C.c_uint64 * n)(*([100000] * n))
    assert lib.fw_ctx_set_parent_velocities(ctx, n, handles, big3) == _ffi.FW_EINVAL
    assert lib.fw_ctx_set_modifiers(ctx, n, handles, big1, big1) == _ffi.FW_EINVAL
    assert lib.fw_ctx_queue(ctx, n, handles, cnt) == _ffi.FW_EINVAL
    for fr in range(30, 50):  # nothing of the refused calls took effect
        system.update(DT)
        for p in pairs:
            p.step_cpu(DT)
    for k, p in enumerate(pairs):
        p.check(what=f"spawner {k}")
    assert all(p.gpu.counts()[0] > 1000 for p in pairs[::2]) and sum(p.gpu.counts()[0] for p in pairs[1::2]) > 1000


def test_batched_origins_are_all_or_nothing(system):
    """fw_ctx_set_origins (one FFI call for the transforms of every spawner, core.rs:377): the same effect as one
    fw_spawner_set_origin per spawner; one invalid handle -> FW_EINVAL and no origin changes"""
    import ctypes as C

    from bevy_firework_amd import _ffi
    from bevy_firework_amd.system import FwError

    pairs = []
    for k in range(5):
        sp, _ = workloads.stress_test(rate=3000.0)
        pairs.append(Pair(system, sp, S.Transform((float(k), 0.1, 0.0)), seed=SEED, uid=500 + k))
    for fr in range(20):
        for k, p in enumerate(pairs):  # every spawner moves every frame: update() pushes all of them in one call
            tf = S.Transform((float(k) + 0.01 * fr, 0.1, 0.02 * fr))
            p.gpu.set_transform(tf)
            p.cpu.set_origin(tf.translation, tf.rotation)
        system.update(DT)
        for p in pairs:
            p.step_cpu(DT)
    lib, ctx = system._lib, system._ctx
    n = len(pairs)
    handles = (C.c_int32 * n)(*[p.gpu.handle for p in pairs])
    handles[3] = 9999  # not a spawner
    tr = (C.c_float * (3 * n))(*([100.0] * (3 * n)))
    ro = (C.c_float * (4 * n))(*([0.0, 0.0, 0.0, 1.0] * n))
    assert lib.fw_ctx_set_origins(ctx, n, handles, tr, ro) == _ffi.FW_EINVAL
    for fr in range(20, 40):  # nobody moved to x = 100: step() keeps the origins of the last update()
        system.step(DT)
        for p in pairs:
            p.step_cpu(DT)
    for k, p in enumerate(pairs):
        p.check(what=f"spawner {k}")
        assert p.gpu.counts()[0] > 1000
