"""Randomised parity: random spawner settings (curve kinds and key counts, shapes, pacings with offsets, several
emission entries per type, Nested entries, drag / acceleration, modifiers, transforms) and irregular step sequences,
HIP path against the CPU oracle through the C ABI.  Sizes are kept where the oracle finishes in about a second per case.
Needs an MI355X."""
import math
import os

import numpy as np
import pytest

from bevy_firework_amd import settings as S
from parity import Pair

pytestmark = pytest.mark.gpu
SEED = 0x5EED
EXTRA = int(os.environ.get("FW_FUZZ_EXTRA", "0"))  # more random cases than the committed suite runs (spare GPU time)
OFF = int(os.environ.get("FW_FUZZ_OFFSET", "0"))  # ... and other ones: the case numbers (= the seeds) start here


def _curve(rng):
    k = rng.integers(0, 3)
    if k == 0:
        return S.FireworkCurve.constant(float(rng.uniform(0.2, 2.0)))
    n = int(rng.integers(2, 6))
    vals = [float(v) for v in rng.uniform(0.0, 3.0, size=n)]
    if k == 1:
        return S.FireworkCurve.even_samples(vals)
    ts = np.sort(rng.uniform(0.0, 1.0, size=n))
    ts[0], ts[-1] = (0.0, 1.0) if rng.random() < 0.7 else (ts[0], ts[-1])
    ts = np.unique(ts.astype(np.float32))
    return S.FireworkCurve.uneven_samples([(float(t), vals[i]) for i, t in enumerate(ts)]) if len(ts) >= 2 \
        else S.FireworkCurve.constant(vals[0])


def _gradient(rng):
    k = rng.integers(0, 3)
    col = lambda: tuple(float(c) for c in rng.uniform(0.0, 4.0, size=4))
    if k == 0:
        return S.FireworkGradient.constant(col())
    n = int(rng.integers(2, 6))
    if k == 1:
        return S.FireworkGradient.even_samples([col() for _ in range(n)])
    ts = np.unique(np.sort(rng.uniform(0.0, 1.0, size=n)).astype(np.float32))
    if len(ts) < 2:
        return S.FireworkGradient.constant(col())
    return S.FireworkGradient.uneven_samples([(float(t), col()) for t in ts])


def _unit(rng):
    v = rng.normal(size=3)
    v /= np.linalg.norm(v)
    return tuple(float(c) for c in v)


def _randvec(rng, mag_hi):
    if rng.random() < 0.25:
        return S.RandVec3.constant(tuple(float(c) for c in rng.uniform(-1.0, 1.0, size=3)))
    lo = float(rng.uniform(0.0, mag_hi * 0.5))
    return S.RandVec3(S.RandF32(lo, float(lo + rng.uniform(0.0, mag_hi))), _unit(rng), float(rng.uniform(0.0, math.pi)))


def _pacing(rng, scale):
    k = rng.integers(0, 4)
    if k == 0:
        return S.EmissionPacing.rate(float(rng.uniform(2000.0, 40000.0) * scale))
    if k == 1:
        dur = float(rng.uniform(0.2, 1.5))
        a, b = sorted(rng.uniform(0.0, 1.0, size=2))
        if b - a < 0.05:
            a, b = 0.0, 1.0
        return S.EmissionPacing.CountOverDuration(float(rng.uniform(500.0, 20000.0) * scale), dur, float(a), float(b))
    if k == 2:
        return S.EmissionPacing.OneShot(int(rng.integers(1, 30000) * scale) + 1)
    return S.EmissionPacing.OnDemand()


def _spawner(rng, scale=1.0, const_p=0.2):
    """const_p: probability that a particle type has ONE lifetime value (the in-place ring path when it is switched on)"""
    n_types = int(rng.integers(1, 3))
    types = []
    for _ in range(n_types):
        lo = float(rng.uniform(0.01, 0.6))
        types.append(S.ParticleSettings(
            lifetime=S.RandF32(lo, float(lo + rng.uniform(0.0, 1.2))) if rng.random() < 1.0 - const_p else S.RandF32.constant(lo + 0.3),
            scale_curve=_curve(rng), initial_scale=S.RandF32(0.01, float(rng.uniform(0.02, 0.2))),
            acceleration=tuple(float(c) for c in rng.uniform(-10.0, 10.0, size=3)),
            angular_acceleration=tuple(float(c) for c in rng.uniform(-2.0, 2.0, size=3)),
            linear_drag=float(rng.uniform(0.0, 1.0)), angular_drag=float(rng.uniform(0.0, 1.0)),
            base_color=_gradient(rng), emissive_color=_gradient(rng), pbr=bool(rng.random() < 0.5)))
    emissions = []
    for _ in range(int(rng.integers(1, 4))):
        shape = [S.EmissionShape.Point(), S.EmissionShape.Sphere(float(rng.uniform(0.1, 2.0))),
                 S.EmissionShape.Circle(_unit(rng), float(rng.uniform(0.1, 2.0)))][int(rng.integers(0, 3))]
        emissions.append(S.EmissionSettings(
            particle_index=int(rng.integers(0, n_types)), emission_pacing=_pacing(rng, scale), emission_shape=shape,
            initial_velocity=_randvec(rng, 6.0), initial_velocity_radial=S.RandF32(0.0, float(rng.uniform(0.0, 3.0))),
            inherit_parent_velocity=bool(rng.random() < 0.5),
            initial_rotation=tuple(float(c) for c in (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))),
            initial_angular_velocity=_randvec(rng, 8.0)))
    if n_types == 2 and rng.random() < 0.6:  # one Nested entry: type 1 particles born on type 0 particles
        emissions.append(S.EmissionSettings(
            particle_index=1, emission_mode=S.EmissionMode.Nested(0),
            emission_pacing=S.EmissionPacing.CountOverDuration(float(rng.uniform(1.0, 8.0)), 1.0, 0.0, float(rng.uniform(0.3, 1.0))),
            inherit_parent_velocity=bool(rng.random() < 0.5), initial_velocity=_randvec(rng, 2.0)))
    return S.ParticleSpawner(types, emissions)


def _steps(rng, n):
    base = float(rng.choice([1 / 60, 1 / 30, 1 / 144, 0.011]))
    out = []
    for i in range(n):
        r = rng.random()
        out.append(base if r < 0.8 else (0.0 if r < 0.83 else float(rng.uniform(0.002, 0.05))))
    return out


def _random_spawner_case(case, seed_base, const_p, sizes):
    from bevy_firework_amd.system import ParticleSystem

    rng = np.random.default_rng(seed_base + case)
    spawner = _spawner(rng, scale=1.0 if case % 4 else 6.0, const_p=const_p)  # every fourth case is several tiles per type
    tf = S.Transform(tuple(float(c) for c in rng.uniform(-2.0, 2.0, size=3)),
                     tuple(float(c) for c in (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))))
    mod = S.EffectModifier(float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0))) if rng.random() < 0.5 else None
    with ParticleSystem(device=0, seed=SEED) as system:
        pair = Pair(system, spawner, tf, seed=SEED, uid=(seed_base // 10) + case, modifier=mod)
        on_demand = any(e.emission_pacing.kind == S.PACING_ONDEMAND for e in spawner.emission_settings)
        for i, dt in enumerate(_steps(rng, 36)):
            dt = np.float32(dt)
            if on_demand and i % 5 == 0:
                pair.queue(int(rng.integers(0, 4000)))
            system.update(dt)
            pair.step_cpu(dt)
            if i % 6 == 5 or i == 35:
                pair.check(what=f"case {case} frame {i} dt={dt}")
        sizes[case] = pair.gpu.counts()


@pytest.mark.parametrize("case", list(range(OFF, OFF + 40 + EXTRA)) + [339])  # 339: OnDemand parents outgrow the derived capacity of their Nested children
def test_random_spawner_matches_the_oracle(case):
    _random_spawner_case(case, 1000, 0.2, test_random_spawner_matches_the_oracle.sizes)


test_random_spawner_matches_the_oracle.sizes = {}


@pytest.mark.parametrize("case", list(range(OFF, OFF + 30 + EXTRA)) + ([1100] if OFF or EXTRA <= 1070 else []))  # 1100: a 17k-per-frame Global burst into a type that also receives Nested children
def test_random_spawner_with_single_lifetimes_matches_the_oracle(case):
    """the same generator with nine types in ten on ONE lifetime value: rings that wrap and grow, rings of spawners with
    Nested entries (materialised spawns, children counted on the device), types that receive both kinds of particles
    (not eligible) -- whatever the draw gives, on both update paths"""
    _random_spawner_case(case, 21000, 0.9, test_random_spawner_with_single_lifetimes_matches_the_oracle.sizes)


test_random_spawner_with_single_lifetimes_matches_the_oracle.sizes = {}


def test_random_cases_were_not_trivial():
    """bookkeeping for the cases above: enough of them must have held a meaningful number of particles at the end"""
    sizes = test_random_spawner_matches_the_oracle.sizes
    if not sizes:
        pytest.skip("the random cases did not run in this session")
    totals = [sum(c) for c in sizes.values()]
    assert sum(t > 2000 for t in totals) >= len(totals) // 2, totals
    assert max(totals) > 50000, totals


@pytest.mark.parametrize("case", range(OFF, OFF + 4 + EXTRA // 16))
def test_random_multi_spawner_system(case):
    """a dozen random spawners in one context: many segments in one launch, more spawn ops than fit the kernel
    arguments (op table read from the pinned ring), per-segment forecast entries; every spawner against its own oracle"""
    from bevy_firework_amd.system import ParticleSystem

    rng = np.random.default_rng(5000 + case)
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = []
        for k in range(12):
            spawner = _spawner(rng, scale=0.5)
            tf = S.Transform(tuple(float(c) for c in rng.uniform(-3.0, 3.0, size=3)))
            pairs.append(Pair(system, spawner, tf, seed=SEED, uid=200 + 16 * case + k))
        next_uid = 200 + 16 * case + 12
        for i, dt in enumerate(_steps(rng, 30)):
            dt = np.float32(dt)
            if i in (7, 13, 21):  # entities come and go: a spawner is despawned, another takes its slots
                k = int(rng.integers(0, len(pairs)))
                system.despawn(pairs[k].gpu)
                pairs[k].cpu.close()
                pairs[k] = Pair(system, _spawner(rng, scale=0.5), S.Transform(tuple(float(c) for c in rng.uniform(-3.0, 3.0, size=3))),
                                seed=SEED, uid=1000 * (case + 1) + next_uid)
                next_uid += 1
            system.update(dt)
            for p in pairs:
                p.step_cpu(dt)
            if i % 10 == 9 or i in (8, 14, 22):
                for k, p in enumerate(pairs):
                    p.check(what=f"case {case} spawner {k} frame {i}")
        assert sum(sum(p.gpu.counts()) for p in pairs) > 5000


@pytest.mark.parametrize("case", range(OFF, OFF + 12 + EXTRA // 4))
def test_random_scenario_with_api_calls_between_frames(case):
    """the calls a host makes between frames -- moving the origin, parent velocity, modifier, queueing, rewriting the
    particles, the destroyed-particle stream, AABB and instance reads, attaching an instance buffer -- in random order
    on a random spawner; state, destroyed records, bounds and instance records against the oracle after every call"""
    _api_scenario(case, 9000, 0.2)


@pytest.mark.parametrize("case", range(OFF, OFF + 12 + EXTRA // 4))
def test_random_scenario_with_api_calls_on_single_lifetime_types(case):
    """... with nine types in ten on one lifetime value (rings; a rewritten type continues on the general path)"""
    _api_scenario(case, 23000, 0.9)


def _api_scenario(case, seed_base, const_p):
    import torch
    from bevy_firework_amd.system import ParticleSystem

    rng = np.random.default_rng(seed_base + case)
    spawner = _spawner(rng, scale=1.0 if case % 3 else 4.0, const_p=const_p)
    for p in spawner.particle_settings:  # the destroyed stream only exists for types that register a handler
        p.particles_destroyed = (lambda dead: None) if rng.random() < 0.6 else None
    with ParticleSystem(device=0, seed=SEED) as system:
        pair = Pair(system, spawner, S.Transform(), seed=SEED, uid=(seed_base // 30) + case)
        buf = None
        # every other case attaches the WINDOWED buffer (records at [first, first + count): a range ring keeps its path)
        attach = pair.gpu.attach_instances_window if case % 2 else pair.gpu.attach_instances
        for i, dt in enumerate(_steps(rng, 30)):
            dt = np.float32(dt)
            a = int(rng.integers(0, 8))
            if a == 0:
                tf = S.Transform(tuple(float(c) for c in rng.uniform(-3.0, 3.0, size=3)),
                                 tuple(float(c) for c in (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))))
                pair.gpu.set_transform(tf)
                pair.cpu.set_origin(tf.translation, tf.rotation)
            elif a == 1:
                v = tuple(float(c) for c in rng.uniform(-2.0, 2.0, size=3))
                pair.gpu.set_parent_velocity(v)
                pair.cpu.set_parent_velocity(v)
            elif a == 2:
                m = S.EffectModifier(float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))
                pair.gpu.set_modifier(m)
                pair.cpu.set_modifier(m)
            elif a == 3:
                pair.queue(int(rng.integers(0, 3000)))
            elif a == 4 and i > 4:
                t = int(rng.integers(0, pair.n_types))
                parts = pair.cpu.particles(t)[:: int(rng.integers(1, 4))].copy()
                pair.gpu.write_particles(t, parts)
                pair.cpu.write_particles(t, parts)
            elif a == 5 and buf is None:
                buf = torch.full((60000 * 16,), float("nan"), dtype=torch.float32, device="cuda")
                attach(buf.data_ptr(), 60000, particle_type=0)
            elif a == 6 and i > 8 and rng.random() < 0.4:  # Changed<ParticleSpawner>: state reset (core.rs:343-365)
                pair.gpu.update_settings(spawner)
                pair.cpu.reset()
                if buf is not None:  # the rebuilt types start detached
                    attach(buf.data_ptr(), 60000, particle_type=0)
            system.update(dt)
            pair.step_cpu(dt)
            if i % 3 == 2:
                pair.check(what=f"case {case} frame {i}")
                for t in range(pair.n_types):
                    if spawner.particle_settings[t].particles_destroyed is not None:
                        from parity import assert_particles_match
                        assert_particles_match(pair.gpu.destroyed(t), pair.cpu.destroyed(t), False, f"destroyed type {t} frame {i}")
                any_g, mn_g, mx_g = pair.gpu.aabb()
                any_c, mn_c, mx_c = pair.cpu.aabb()
                assert any_g == any_c
                if any_g:
                    assert np.allclose(mn_g, mn_c, rtol=1e-5, atol=1e-4) and np.allclose(mx_g, mx_c, rtol=1e-5, atol=1e-4)
                if buf is not None:
                    first = pair.gpu.instance_window(0)[0] if case % 2 else 0
                    n = max(0, min(pair.gpu.count(0), 60000 - first))
                    ref = pair.gpu.instances(0)[:n]
                    got = buf[first * 16: (first + n) * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                    want = ref.view(np.uint32).reshape(n, 16)
                    if not np.array_equal(got, want):
                        rows = np.flatnonzero((got != want).any(axis=1))
                        raise AssertionError(f"case {case} frame {i}: instance records: {len(rows)} of {n} differ, rows "
                                             f"{rows[0]}..{rows[-1]}, first got {got[rows[0]].view(np.float32)} want "
                                             f"{want[rows[0]].view(np.float32)}")


@pytest.mark.parametrize("case", range(OFF, OFF + 16 + EXTRA // 32))
def test_every_shortcut_gives_the_state_of_the_plain_path(case, monkeypatch):
    """rings (types with one lifetime value), no rotation / angular-velocity planes (types that cannot turn), the side
    stream: each is a shortcut around work whose result is known in advance.  A random spawner at a size the oracle would
    take minutes for (hundreds of thousands of particles), stepped irregularly, must end in EXACTLY the state the plain
    path -- everything compacted, every plane kept, one stream -- produces: every field of every particle, numerically
    equal (a shortcut may turn a negative zero into a positive one, nothing else), destroyed streams and bounds too."""
    from bevy_firework_amd.system import ParticleSystem

    rng0 = np.random.default_rng(31000 + case)
    spawner = _spawner(rng0, scale=12.0, const_p=0.6)
    for e in spawner.emission_settings:  # half of the cases: nothing spins
        if case % 2 == 0:
            e.initial_angular_velocity = S.RandVec3.constant((0.0, 0.0, 0.0))
    if case % 2 == 0:
        for p in spawner.particle_settings:
            p.angular_acceleration = (0.0, 0.0, 0.0)
    for p in spawner.particle_settings:
        p.particles_destroyed = lambda dead: None
    dts = [np.float32(x) for x in _steps(np.random.default_rng(32000 + case), 40)]
    on_demand = any(e.emission_pacing.kind == S.PACING_ONDEMAND for e in spawner.emission_settings)
    results = {}
    for name, env in (("shortcuts", {"FW_FIFO": "1", "FW_FIFO_MIN": "0", "FW_NOSPIN": "1", "FW_RANGE": "1", "FW_RANGE_MIN": "0"}),
                      # (the non-temporal forms of the ring kernels: what launches of more than fw_ctx::nt_bytes / nt_wo_bytes run)
                      ("shortcuts, nt", {"FW_FIFO": "1", "FW_FIFO_MIN": "0", "FW_NOSPIN": "1", "FW_RANGE": "1", "FW_RANGE_MIN": "0", "FW_NT_MB": "0"}),
                      ("rings, nt", {"FW_FIFO": "1", "FW_FIFO_MIN": "0", "FW_NOSPIN": "0", "FW_RANGE": "1", "FW_RANGE_MIN": "0", "FW_NT_WO_MB": "0"}),
                      ("rings only", {"FW_FIFO": "1", "FW_FIFO_MIN": "0", "FW_NOSPIN": "0", "FW_RANGE": "0", "FW_FIFO_SMALL": "0"}),
                      ("range rings only", {"FW_FIFO": "0", "FW_NOSPIN": "0", "FW_RANGE": "1", "FW_RANGE_MIN": "0", "FW_RANGE_SMALL": "0"}),
                      ("range rings, no planes", {"FW_FIFO": "0", "FW_NOSPIN": "1", "FW_RANGE": "1", "FW_RANGE_MIN": "0"}),
                      ("no planes only", {"FW_FIFO": "0", "FW_NOSPIN": "1", "FW_RANGE": "0"}),
                      ("plain", {"FW_FIFO": "0", "FW_NOSPIN": "0", "FW_FIFO_STREAM": "0", "FW_RANGE": "0"})):
        for k in ("FW_FIFO", "FW_FIFO_MIN", "FW_NOSPIN", "FW_FIFO_STREAM", "FW_RANGE", "FW_RANGE_MIN", "FW_NT_MB", "FW_NT_WO_MB", "FW_FIFO_SMALL", "FW_RANGE_SMALL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with ParticleSystem(device=0, seed=SEED) as system:
            h = system.spawn(spawner, S.Transform((0.5, 1.0, -0.5)), uid=900 + case)
            dead_total = 0
            for i, dt in enumerate(dts):
                if on_demand and i % 5 == 0:
                    h.queue_particles(20000)
                system.update(dt)
                dead_total += sum(len(h.destroyed(t)) for t in range(len(spawner.particle_settings)))
            results[name] = ([h.particles(t) for t in range(len(spawner.particle_settings))], h.aabb(), dead_total)
    ref_parts, ref_box, ref_dead = results["plain"]
    for name in ("shortcuts", "shortcuts, nt", "rings, nt", "rings only", "range rings only", "range rings, no planes", "no planes only"):
        parts, box, dead = results[name]
        assert dead == ref_dead, (name, dead, ref_dead)
        for t, (a, b) in enumerate(zip(parts, ref_parts)):
            assert len(a) == len(b), (name, t, len(a), len(b))
            for f in a.dtype.names:
                assert np.array_equal(a[f], b[f], equal_nan=True) if a[f].dtype.kind == "f" else np.array_equal(a[f], b[f]), (name, t, f)
        assert box[0] == ref_box[0] and np.array_equal(box[1], ref_box[1]) and np.array_equal(box[2], ref_box[2]), name
    test_every_shortcut_gives_the_state_of_the_plain_path.sizes[case] = sum(len(p) for p in ref_parts)


test_every_shortcut_gives_the_state_of_the_plain_path.sizes = {}


def _collider(rng):
    k = int(rng.choice([0, 0, 1, 2, 3, 4]))  # (planes twice as often: the collider most particles meet)
    layers = int(rng.choice([1, 1, 2, 3]))
    # (solids sit where the particles pass: the emitters are around (0, 2, 0) and mostly point down)
    centre = lambda: (float(rng.uniform(-1.3, 1.3)), float(rng.uniform(-1.5, 1.8)), float(rng.uniform(-1.3, 1.3)))
    if k >= 3:  # cylinder / cone (examples/textures.rs:195, 211), tilted
        q = rng.normal(size=4)
        make = S.Collider.Cylinder if k == 3 else S.Collider.Cone
        return make(centre(), float(rng.uniform(0.4, 1.4)), float(rng.uniform(0.2, 1.6)), tuple(float(c) for c in q / np.linalg.norm(q)), layers)
    if k == 0:
        return S.Collider.Plane(tuple(float(c) for c in rng.uniform(-1.0, 0.5, size=3)),
                                tuple(float(c) for c in (rng.normal(size=3) * 0.3 + np.array([0.0, 1.0, 0.0]))), layers)
    if k == 1:
        return S.Collider.Sphere(centre(), float(rng.uniform(0.2, 1.2)), layers)
    q = rng.normal(size=4)
    return S.Collider.Box(centre(), tuple(float(c) for c in rng.uniform(0.2, 1.0, size=3)), tuple(float(c) for c in q / np.linalg.norm(q)), layers)


@pytest.mark.parametrize("case", range(OFF, OFF + 16 + EXTRA // 4))
def test_random_colliding_spawner_matches_the_oracle_bit_for_bit(case):
    """particle_collision (core.rs:744-800) under random settings: one to four random colliders (planes, spheres, rotated
    boxes, tilted cylinders and cones, on different layers), random restitution / friction / destroy_on_collision / filter mask, one or two colliding
    types next to a plain one, a Nested entry now and then, colliders replaced half way.  A bounce amplifies any
    difference, so the scene is built without a single libm call (Point emission, zero spread: directions vary through the
    entries and a parent velocity that changes every frame) and EVERY field is compared bit for bit."""
    from bevy_firework_amd.system import ParticleSystem

    rng = np.random.default_rng(41000 + case)
    n_types = int(rng.integers(1, 4))
    types = []
    for t in range(n_types):
        lo = float(rng.uniform(0.2, 0.9))
        colliding = t == 0 or rng.random() < 0.5
        cs = S.ParticleCollisionSettings(float(rng.uniform(0.0, 1.0)), float(rng.uniform(0.0, 1.0)), bool(rng.random() < 0.3),
                                         int(rng.choice([0xFFFFFFFF, 1, 2, 3]))) if colliding else None
        p = S.ParticleSettings(lifetime=S.RandF32(lo, float(lo + rng.uniform(0.0, 0.8))) if rng.random() < 0.7 else S.RandF32.constant(lo),
                               scale_curve=_curve(rng), initial_scale=S.RandF32(0.01, 0.05),
                               acceleration=tuple(float(c) for c in rng.uniform(-10.0, 3.0, size=3)),
                               linear_drag=float(rng.uniform(0.0, 0.5)), base_color=_gradient(rng), collision_settings=cs)
        p.particles_destroyed = lambda dead: None
        types.append(p)
    emissions = []
    for t in range(n_types):
        for _ in range(int(rng.integers(1, 3))):
            d = rng.normal(size=3) + np.array([0.0, -1.0, 0.0])
            emissions.append(S.EmissionSettings(
                particle_index=t, emission_pacing=S.EmissionPacing.rate(float(rng.uniform(500.0, 6000.0))),
                initial_velocity=S.RandVec3(S.RandF32(0.5, float(rng.uniform(1.0, 9.0))), tuple(float(c) for c in d / np.linalg.norm(d)), 0.0),
                inherit_parent_velocity=bool(rng.random() < 0.7)))
    if n_types >= 2 and rng.random() < 0.5:
        emissions.append(S.EmissionSettings(
            particle_index=1, emission_mode=S.EmissionMode.Nested(0),
            emission_pacing=S.EmissionPacing.CountOverDuration(float(rng.uniform(2.0, 10.0)), 1.0, 0.0, float(rng.uniform(0.3, 1.0))),
            initial_velocity=S.RandVec3(S.RandF32(0.0, 2.0), (0.0, -1.0, 0.0), 0.0), inherit_parent_velocity=bool(rng.random() < 0.5)))
    worlds = [[_collider(rng) for _ in range(int(rng.integers(1, 5)))] for _ in range(2)]
    with ParticleSystem(device=0, seed=SEED) as system:
        pair = Pair(system, S.ParticleSpawner(types, emissions), S.Transform(tuple(float(c) for c in rng.uniform(-0.5, 0.5, size=3) + np.array([0.0, 2.0, 0.0]))),
                    seed=SEED, uid=700 + case)
        system.set_colliders(worlds[0])
        pair.cpu.set_colliders(worlds[0])
        import oracle
        free = oracle.OracleSpawner(pair.spawner, seed=SEED, uid=700 + case, transform=pair.cpu_transform) if case < 8 else None
        hits = 0
        for i, dt in enumerate(_steps(rng, 40)):
            dt = np.float32(dt)
            if i == 20:
                system.set_colliders(worlds[1])
                pair.cpu.set_colliders(worlds[1])
            pv = tuple(float(np.float32(c)) for c in rng.uniform(-1.0, 1.0, size=3))
            pair.gpu.set_parent_velocity(pv)
            pair.cpu.set_parent_velocity(pv)
            system.update(dt)
            pair.step_cpu(dt)
            if free is not None:  # the same spawner in an empty world: how much do the colliders matter?
                free.set_parent_velocity(pv)
                free.step(dt)
            if i % 8 == 7:
                pair.check(exact_all=True, what=f"case {case} frame {i}")
                for t in range(n_types):
                    gd, cd = pair.gpu.destroyed(t), pair.cpu.destroyed(t)
                    assert len(gd) == len(cd), f"case {case} frame {i} type {t}: destroyed {len(gd)} != {len(cd)}"
                    for f in ("age", "position", "velocity", "scale", "rotation", "angular_velocity", "lifetime",
                              "initial_scale", "base_color", "emissive_color"):
                        assert np.array_equal(gd[f], cd[f]), f"case {case} frame {i} type {t}: destroyed.{f}"
                    hits += int(np.count_nonzero(cd["age"] < cd["lifetime"]))  # destroyed by a collision, not by age
        moved = None
        if free is not None:
            a, b = pair.cpu.particles(0), free.particles(0)
            moved = len(a) != len(b) or int(np.count_nonzero((a["position"] != b["position"]).any(axis=1)))
            free.close()
        test_random_colliding_spawner_matches_the_oracle_bit_for_bit.sizes[case] = (sum(pair.gpu.counts()), hits, moved)


test_random_colliding_spawner_matches_the_oracle_bit_for_bit.sizes = {}


def test_colliding_cases_were_not_trivial():
    """bookkeeping for the cases above: the colliders must have changed the outcome in most of the cases that were also run in
    an empty world, and some particles must have been destroyed by a collision"""
    sizes = test_random_colliding_spawner_matches_the_oracle_bit_for_bit.sizes
    checked = [v for v in sizes.values() if v[2] is not None]
    if len(checked) < 8:
        pytest.skip("the colliding cases did not run in this session")
    assert sum(bool(v[2]) for v in checked) >= len(checked) // 2, sizes
    assert sum(v[0] for v in sizes.values()) > 20000 and sum(v[1] for v in sizes.values()) > 0, sizes


@pytest.mark.parametrize("case", range(OFF, OFF + 24 + EXTRA // 2))
def test_random_nested_topologies(case):
    """Nested entries in every arrangement the settings allow: chains (smoke on sparks on seeds), several Nested entries on one
    parent type (one last_emitted_age plane each), particles that emit onto their own type, a type that receives children from
    two parent types and Global particles too -- and the entries in RANDOM order, so that a Nested entry may come before the
    Global entry that feeds its parents (it then sees only the parents of earlier frames, core.rs:377-546 walks the entries in
    index order).  Counts, order, ages and last_emitted_age bit for bit; small per-parent counts keep the population bounded."""
    from bevy_firework_amd.system import ParticleSystem

    rng, types, entries = _nested_topology(case)
    with ParticleSystem(device=0, seed=SEED) as system:
        pair = Pair(system, S.ParticleSpawner(types, entries), S.Transform(tuple(float(c) for c in rng.uniform(-1.0, 1.0, size=3))),
                    seed=SEED, uid=800 + case)
        for i, dt in enumerate(_steps(rng, 48)):
            dt = np.float32(dt)
            system.update(dt)
            pair.step_cpu(dt)
            if i % 8 == 7:
                pair.check(what=f"case {case} frame {i}")
                for k, e in enumerate(entries):
                    if e.emission_mode.kind == S.MODE_NESTED:
                        t = e.emission_mode.target_particle_type
                        assert np.array_equal(pair.gpu.last_emitted(t, k), pair.cpu.last_emitted(t, k)), f"case {case} frame {i}: last_emitted_age[{k}] of type {t}"
        test_random_nested_topologies.sizes[case] = pair.gpu.counts()


def _nested_topology(case):
    rng = np.random.default_rng(51000 + case)
    n_types = int(rng.integers(2, 4))
    proto = _spawner(np.random.default_rng(52000 + case), scale=0.3, const_p=0.5 if case % 2 else 0.1)
    types = []
    for t in range(n_types):
        p = proto.particle_settings[t % len(proto.particle_settings)]
        lo = float(rng.uniform(0.15, 0.5))
        types.append(S.ParticleSettings(
            lifetime=S.RandF32.constant(lo) if rng.random() < (0.5 if case % 2 else 0.1) else S.RandF32(lo, float(lo + rng.uniform(0.0, 0.4))),
            scale_curve=p.scale_curve, initial_scale=p.initial_scale, acceleration=p.acceleration,
            angular_acceleration=p.angular_acceleration if rng.random() < 0.5 else (0.0, 0.0, 0.0), linear_drag=p.linear_drag,
            angular_drag=p.angular_drag, base_color=p.base_color, emissive_color=p.emissive_color))
    entries = [S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(float(rng.uniform(800.0, 4000.0))),
                                  emission_shape=S.EmissionShape.Sphere(0.5), initial_velocity=_randvec(rng, 4.0),
                                  initial_angular_velocity=_randvec(rng, 4.0) if rng.random() < 0.5 else S.RandVec3.constant((0.0, 0.0, 0.0)))]
    if rng.random() < 0.4:  # a second Global entry somewhere (a type that gets both kinds of particles)
        entries.append(S.EmissionSettings(particle_index=int(rng.integers(0, n_types)),
                                          emission_pacing=_pacing(rng, 0.1), initial_velocity=_randvec(rng, 3.0)))
    for _ in range(int(rng.integers(1, 5))):
        child, parent = int(rng.integers(0, n_types)), int(rng.integers(0, n_types))
        loop = child <= parent  # own type or back up the chain: keep the offspring below one per parent life
        a, b = sorted(rng.uniform(0.0, 1.0, size=2))
        if b - a < 0.1:
            a, b = 0.0, float(rng.uniform(0.3, 1.0))
        entries.append(S.EmissionSettings(
            particle_index=child, emission_mode=S.EmissionMode.Nested(parent),
            emission_pacing=S.EmissionPacing.CountOverDuration(float(rng.uniform(0.3, 0.9) if loop else rng.uniform(1.0, 5.0)), 1.0, float(a), float(b)),
            inherit_parent_velocity=bool(rng.random() < 0.5), initial_velocity=_randvec(rng, 2.0),
            emission_shape=S.EmissionShape.Point() if rng.random() < 0.5 else S.EmissionShape.Sphere(0.2)))
    order = rng.permutation(len(entries))
    return rng, types, [entries[k] for k in order]


test_random_nested_topologies.sizes = {}


def test_nested_topology_cases_were_not_trivial():
    sizes = test_random_nested_topologies.sizes
    if len(sizes) < 24:
        pytest.skip("the topology cases did not run in this session")
    totals = [sum(c) for c in sizes.values()]
    children = [sum(c[1:]) for c in sizes.values()]
    assert sum(t > 1000 for t in totals) >= len(totals) // 2 and sum(c > 300 for c in children) >= len(children) // 3, sizes
