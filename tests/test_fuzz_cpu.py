"""The random spawners of tests/test_gpu_fuzz.py (every curve kind and key count, shapes, pacings with offsets, several entries per
type, a Nested entry, drag / acceleration, modifiers, rotated emitters, irregular steps with zero-length frames, OnDemand bursts) on
the two CPU restatements against each other: the C oracle (what the GPU suite checks the HIP path against) and the independent numpy
restatement (tests/golden/np_sim.py, a different libm).  Counts and order, bit-identical age / lifetime / scale / colours /
last_emitted_age, vector fields inside the tolerance of parity.py.  The oracle is thereby cross-checked on the same distribution of
settings the HIP path is fuzzed on -- at sizes the numpy restatement steps in a second."""
import os
import sys

import numpy as np
import pytest

import oracle
import parity
from bevy_firework_amd import settings as S

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 0x5EED


def _mods():
    for p in (HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import np_sim
    import test_gpu_fuzz as fz  # (the generators only: nothing in them touches a GPU)

    return np_sim, fz


@pytest.mark.parametrize("case", range(64))
def test_oracle_and_numpy_restatement_agree_on_random_spawners(case):
    np_sim, fz = _mods()
    rng = np.random.default_rng(77000 + case)
    spawner = fz._spawner(rng, scale=0.12 if case % 3 else 0.4, const_p=0.2 if case % 2 else 0.8)
    tf = S.Transform(tuple(float(c) for c in rng.uniform(-2.0, 2.0, size=3)),
                     tuple(float(c) for c in (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))))
    mod = S.EffectModifier(float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0))) if rng.random() < 0.5 else None
    pv = tuple(float(c) for c in rng.uniform(-1.0, 1.0, size=3))
    o = oracle.OracleSpawner(spawner, seed=SEED, uid=900 + case, transform=tf)
    if mod is not None:
        o.set_modifier(mod)
    o.set_parent_velocity(pv)
    n = np_sim.Spawner(spawner, SEED, 900 + case, tf, mod)
    n.parent_velocity = np.asarray(pv, dtype=np.float32)
    on_demand = any(e.emission_pacing.kind == S.PACING_ONDEMAND for e in spawner.emission_settings)
    n_em = len(spawner.emission_settings)
    peak = 0
    for i, dt in enumerate(fz._steps(rng, 40)):
        dt = np.float32(dt)
        if on_demand and i % 5 == 0:
            q = int(rng.integers(0, 500))
            o.queue_particles(q)
            n.queued += q
        o.step(dt), n.step(dt)
        assert o.counts() == [n.count(t) for t in range(len(n.particles))], (case, i)
        peak = max(peak, sum(o.counts()))
        if i % 8 == 7 or i == 39:
            for t in range(len(n.particles)):
                got = o.particles(t)
                want = np.zeros(n.count(t), dtype=[(k, got.dtype[k]) for k in got.dtype.names if k in n.particles[t]])
                for k in want.dtype.names:
                    want[k] = n.particles[t][k]
                parity.assert_particles_match(got, want, what=f"case {case} frame {i} type {t}")
                for e in range(n_em):
                    assert np.array_equal(o.last_emitted(t, e), n.particles[t]["last_emitted_age"][:, e]), (case, i, t, e)
            assert o.active() == n.active(), (case, i)
    test_oracle_and_numpy_restatement_agree_on_random_spawners.peaks[case] = peak


test_oracle_and_numpy_restatement_agree_on_random_spawners.peaks = {}


def test_the_random_cases_held_particles():
    peaks = test_oracle_and_numpy_restatement_agree_on_random_spawners.peaks
    if not peaks:
        pytest.skip("the random cases did not run in this session")
    assert sum(p > 300 for p in peaks.values()) >= len(peaks) // 2, peaks


@pytest.mark.parametrize("case", range(24))
def test_oracle_and_numpy_restatement_agree_on_random_nested_topologies(case):
    """the Nested arrangements of tests/test_gpu_fuzz.py::test_random_nested_topologies -- chains, several Nested entries on one parent
    type, particles that emit onto their own type, a type fed by two parent types and Global particles, entries in random order"""
    np_sim, fz = _mods()
    rng, types, entries = fz._nested_topology(case)
    spawner = S.ParticleSpawner(types, entries)
    tf = S.Transform(tuple(float(c) for c in rng.uniform(-1.0, 1.0, size=3)))
    o = oracle.OracleSpawner(spawner, seed=SEED, uid=800 + case, transform=tf)
    n = np_sim.Spawner(spawner, SEED, 800 + case, tf)
    for i, dt in enumerate(fz._steps(rng, 48)):
        dt = np.float32(dt)
        o.step(dt), n.step(dt)
        assert o.counts() == [n.count(t) for t in range(len(n.particles))], (case, i)
        if i % 8 == 7:
            for t in range(len(n.particles)):
                got = o.particles(t)
                for f in ("age", "lifetime", "initial_scale", "scale", "base_color", "emissive_color"):
                    assert np.array_equal(got[f], n.particles[t][f]), (case, i, t, f)
                for k, e in enumerate(entries):
                    assert np.array_equal(o.last_emitted(t, k), n.particles[t]["last_emitted_age"][:, k]), (case, i, t, k)
    assert sum(o.counts()) >= 0


@pytest.mark.parametrize("case", range(16))
def test_oracle_and_numpy_restatement_agree_on_random_colliding_spawners(case):
    """particle_collision (core.rs:744-800) + the analytic ray casts inside whole simulations: one to three types (colliding or
    not, destroy_on_collision now and then, layer masks), one to four random planes / spheres / rotated boxes replaced half way, a
    parent velocity that changes every frame.  Built like tests/test_gpu_fuzz.py's colliding scenes without a single libm call, so
    the two restatements must agree on EVERY field bit for bit -- live particles and the records of the destroyed ones"""
    np_sim, fz = _mods()
    rng = np.random.default_rng(43000 + case)
    n_types = int(rng.integers(1, 4))
    types, emissions = [], []
    for t in range(n_types):
        lo = float(rng.uniform(0.2, 0.9))
        cs = S.ParticleCollisionSettings(float(rng.uniform(0.0, 1.0)), float(rng.uniform(0.0, 1.0)), bool(rng.random() < 0.3),
                                         int(rng.choice([0xFFFFFFFF, 1, 2, 3]))) if (t == 0 or rng.random() < 0.5) else None
        types.append(S.ParticleSettings(
            lifetime=S.RandF32(lo, float(lo + rng.uniform(0.0, 0.8))) if rng.random() < 0.7 else S.RandF32.constant(lo),
            scale_curve=fz._curve(rng), initial_scale=S.RandF32(0.01, 0.05), acceleration=tuple(float(c) for c in rng.uniform(-10.0, 3.0, size=3)),
            linear_drag=float(rng.uniform(0.0, 0.5)), base_color=fz._gradient(rng), collision_settings=cs, particles_destroyed=lambda dead: None))
        for _ in range(int(rng.integers(1, 3))):
            d = rng.normal(size=3) + np.array([0.0, -1.0, 0.0])
            emissions.append(S.EmissionSettings(
                particle_index=t, emission_pacing=S.EmissionPacing.rate(float(rng.uniform(200.0, 1500.0))),
                initial_velocity=S.RandVec3(S.RandF32(0.5, float(rng.uniform(1.0, 9.0))), tuple(float(c) for c in d / np.linalg.norm(d)), 0.0),
                inherit_parent_velocity=bool(rng.random() < 0.7)))
    if n_types >= 2 and rng.random() < 0.5:
        emissions.append(S.EmissionSettings(
            particle_index=1, emission_mode=S.EmissionMode.Nested(0),
            emission_pacing=S.EmissionPacing.CountOverDuration(float(rng.uniform(2.0, 6.0)), 1.0, 0.0, float(rng.uniform(0.3, 1.0))),
            initial_velocity=S.RandVec3(S.RandF32(0.0, 2.0), (0.0, -1.0, 0.0), 0.0), inherit_parent_velocity=bool(rng.random() < 0.5)))
    worlds = [[fz._collider(rng) for _ in range(int(rng.integers(1, 5)))] for _ in range(2)]
    spawner = S.ParticleSpawner(types, emissions)
    tf = S.Transform(tuple(float(c) for c in rng.uniform(-0.5, 0.5, size=3) + np.array([0.0, 2.0, 0.0])))
    o = oracle.OracleSpawner(spawner, seed=SEED, uid=700 + case, transform=tf)
    n = np_sim.Spawner(spawner, SEED, 700 + case, tf)
    o.set_colliders(worlds[0])
    n.colliders = list(worlds[0])
    hits = 0
    for i, dt in enumerate(fz._steps(rng, 40)):
        dt = np.float32(dt)
        if i == 20:
            o.set_colliders(worlds[1])
            n.colliders = list(worlds[1])
        pv = tuple(float(np.float32(c)) for c in rng.uniform(-1.0, 1.0, size=3))
        o.set_parent_velocity(pv)
        n.parent_velocity = np.asarray(pv, dtype=np.float32)
        o.step(dt), n.step(dt)
        assert o.counts() == [n.count(t) for t in range(n_types)], (case, i)
        for t in range(n_types):
            got, dead = o.particles(t), o.destroyed(t)
            for f in ("position", "velocity", "rotation", "angular_velocity", "age", "lifetime", "initial_scale", "scale", "base_color", "emissive_color"):
                assert np.array_equal(got[f], n.particles[t][f]), (case, i, t, f)
            assert len(dead) == len(n.destroyed[t]["age"]), (case, i, t)
            for f in ("position", "velocity", "age", "lifetime", "scale"):
                assert np.array_equal(dead[f], n.destroyed[t][f]), (case, i, t, "destroyed", f)
            hits += int(np.count_nonzero(dead["age"] < dead["lifetime"]))  # destroyed by a collision, not by age
    test_oracle_and_numpy_restatement_agree_on_random_colliding_spawners.hits[case] = hits


test_oracle_and_numpy_restatement_agree_on_random_colliding_spawners.hits = {}


def test_the_colliding_cases_destroyed_particles_on_contact():
    hits = test_oracle_and_numpy_restatement_agree_on_random_colliding_spawners.hits
    if not hits:
        pytest.skip("the colliding cases did not run in this session")
    assert sum(hits.values()) > 50, hits
