"""The spawners of the reference's own examples (/root/reference/examples/*.rs, restated as settings in
bevy_firework_amd/workloads.py) at the examples' own sizes -- tens to hundreds of particles, the regime the crate is written
for -- against the oracle, frame by frame through fill, steady state and deaths, on every update path (tests/conftest.py).
stress_test.rs and stress_test_collision.rs are tests/test_gpu_parity.py::test_stress_test_example and
tests/test_gpu_configs.py::test_stress_test_collision_example."""
import numpy as np
import pytest

from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair, assert_particles_match

pytestmark = pytest.mark.gpu
DT = np.float32(1.0 / 60.0)
SEED = workloads.SEED
BIT_EXACT = ("age", "lifetime", "initial_scale", "scale", "base_color", "emissive_color")


@pytest.fixture()
def system():
    from bevy_firework_amd.system import ParticleSystem

    with ParticleSystem(device=0, seed=SEED) as ps:
        yield ps


def _run(system, pair, frames, every, what, dts=None):
    for fr in range(frames):
        dt = DT if dts is None else np.float32(dts[fr % len(dts)])
        system.update(dt)
        pair.step_cpu(dt)
        if fr % every == every - 1 or fr == frames - 1:
            pair.check(what=f"{what} f{fr}")
            assert pair.gpu.active() == pair.cpu.active()


def test_example_sparks(system):
    """examples/sparks.rs: rate 1000/s, lifetime 0.75 s (~730 live), Circle + cone, 5-key uneven gradient; also under the
    example's slow motion (Space: relative speed 0.05, sparks.rs:110-125) -- frames in which nothing is due"""
    spawner, tf = workloads.example_sparks()
    pair = Pair(system, spawner, tf, seed=SEED, uid=21)
    _run(system, pair, 120, 10, "sparks")
    assert 700 < pair.gpu.count(0) < 760
    _run(system, pair, 200, 20, "sparks, slow motion", dts=[float(DT) * 0.05])
    _run(system, pair, 60, 10, "sparks, back to speed")


def test_example_on_demand(system):
    """examples/on_demand.rs: the sparks spawner with EmissionPacing::OnDemand, one particle per click
    (data.queue_particles(1), on_demand.rs:130-141), a few clicks in a frame now and then"""
    spawner, tf = workloads.example_on_demand()
    pair = Pair(system, spawner, tf, seed=SEED, uid=22)
    rng = np.random.default_rng(5)
    clicks = rng.integers(0, 4, size=240) * (rng.random(240) < 0.3)
    total = 0
    for fr in range(240):
        for _ in range(int(clicks[fr])):
            pair.queue(1)
        total += int(clicks[fr])
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(what=f"on_demand f{fr}")
    assert total > 60 and 0 < pair.gpu.count(0) < total  # (lifetime 0.75 s: most have gone again)
    assert pair.gpu.active() and pair.cpu.active()        # an OnDemand entry never finishes


def test_example_pbr(system):
    """examples/pbr.rs: 150/s for 5 s (~750 live), no initial velocity at all, upward acceleration against drag 0.7, 2-key
    scale curve, 3-key alpha gradient, constant emissive colour"""
    spawner, tf = workloads.example_pbr()
    pair = Pair(system, spawner, tf, seed=SEED, uid=23)
    _run(system, pair, 340, 20, "pbr")
    assert 730 < pair.gpu.count(0) < 760
    g = pair.gpu.particles(0)
    assert (g["pbr"] != 0).all() and (g["velocity"][:, 1] > 0).all()  # nothing but the acceleration moves them


@pytest.mark.parametrize("normal", [(0.0, 1.0, 0.0), (0.3, 0.9, 0.1), (-1.0, 0.2, 0.0)])
def test_example_one_shot(system, normal):
    """examples/one_shot.rs: OneShot(20) in SpawnTransformMode::Local on an emitter rotated onto a contact normal, radial
    velocity; ParticleSpawnerFinished fires once, 2.5 s later, when the last puff has gone (the example despawns the entity
    then: one_shot.rs:138-142)"""
    spawner, tf = workloads.example_one_shot(impulse=4.0, normal=normal, translation=(0.4, -2.0, -0.7))
    pair = Pair(system, spawner, tf, seed=SEED, uid=24)
    fired = []
    pair.gpu.on_finished.append(lambda d: fired.append(d.handle))
    cpu_fired = 0
    for fr in range(170):
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(what=f"one_shot f{fr}")
        assert pair.gpu.active() == pair.cpu.active()
        cpu_fired += bool(pair.cpu.poll_finished())
        if fr == 0:
            assert pair.gpu.count(0) == 20
            s = pair.gpu.particles(0)["initial_scale"]
            assert (s >= np.float32(0.3)).all() and (s <= np.float32(0.5)).all()
    assert fired == [pair.gpu.handle] and cpu_fired == 1 and pair.gpu.count(0) == 0
    system.despawn(pair.gpu)  # what the example's observer does


def _collision_pair(system, spawner, tf, world, uid):
    system.set_colliders(world)
    pair = Pair(system, spawner, tf, seed=SEED, uid=uid)
    pair.cpu.set_colliders(world)
    return pair


def _check_bounced(pair, t, fr, young_age):
    """a colliding type against the oracle: everything without trigonometry in its history bit for bit whatever the particle
    hit; position / velocity / rotation inside the usual allowance for the particles too young to have reached a collider (a
    bounce amplifies the last-bit difference of two libm's cone: tests/test_gpu_configs.py::test_stress_test_collision_example)"""
    assert pair.gpu.counts() == pair.cpu.counts(), fr
    g, c = pair.gpu.particles(t), pair.cpu.particles(t)
    for f in BIT_EXACT:
        assert np.array_equal(g[f], c[f]), (f, fr)
    young = c["age"] < young_age
    assert_particles_match(g[young], c[young], what=f"young particles f{fr}")
    assert np.isfinite(g["position"]).all() and np.isfinite(g["velocity"]).all()
    return g, c


def test_example_collision(system):
    """examples/collision.rs: 100/s for 6.75 s (~670 live) bouncing off the slab and the angled cube; uneven 3-key scale
    curve, 4-key emissive gradient whose last key sits at 0.8 (clamped beyond)"""
    spawner, tf, world = workloads.example_collision()
    pair = _collision_pair(system, spawner, tf, world, 25)
    for fr in range(450):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 30 == 29 or fr == 449:
            g, c = _check_bounced(pair, 0, fr, 0.3)
    assert 650 < len(g) < 690
    # most of the old particles lie on the slab or have bounced off it; nobody on it fell through
    on = (np.abs(g["position"][:, 0]) < 3.9) & (np.abs(g["position"][:, 2]) < 3.9)
    assert np.count_nonzero(on) > 200 and (g["position"][on, 1] > -1e-3).all()
    # and, particle by particle, nearly everybody is where the oracle has it (the rest: grazing rays, see above)
    err = np.abs(g["position"].astype(np.float64) - c["position"].astype(np.float64)).max(axis=1)
    assert np.count_nonzero(err > 1e-3) < 0.05 * len(c)


def test_example_textures_without_a_world(system):
    """examples/textures.rs's spawner with nothing to hit: two particle types, a Global rate entry + a Nested
    CountOverDuration entry (six puffs in the first tenth of a case's life), SpawnTransformMode::Local, an initial rotation and
    a spin slowed by angular_drag -- the full state of both types inside the allowance, the last_emitted_age plane bit for bit"""
    spawner, tf, _ = workloads.example_textures(with_world=False)
    pair = Pair(system, spawner, tf, seed=SEED, uid=26)
    for fr in range(360):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 20 == 19:
            pair.check(what=f"textures f{fr}")
            assert np.array_equal(pair.gpu.last_emitted(0, 1), pair.cpu.last_emitted(0, 1))
    n = pair.gpu.counts()
    assert 55 <= n[0] <= 61 and 100 < n[1] < 160
    g = pair.gpu.particles(0)
    assert (np.abs(np.linalg.norm(g["rotation"], axis=1) - 1.0) < 1e-5).all()
    assert (np.linalg.norm(g["angular_velocity"], axis=1)[g["age"] > 2.0] < 3.0).all()  # 5..15 rad/s, slowed by the drag


def test_example_textures(system):
    """... and in a world (stand-ins for the example's cylinder and cone: workloads.example_textures): the cases bounce"""
    spawner, tf, world = workloads.example_textures()
    pair = _collision_pair(system, spawner, tf, world, 27)
    for fr in range(360):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 20 == 19:
            g, _ = _check_bounced(pair, 0, fr, 0.25)
            assert pair.gpu.counts() == pair.cpu.counts()
            g1, c1 = pair.gpu.particles(1), pair.cpu.particles(1)
            for f in BIT_EXACT:
                assert np.array_equal(g1[f], c1[f]), (f, fr)
            assert np.array_equal(pair.gpu.last_emitted(0, 1), pair.cpu.last_emitted(0, 1))
    inside = np.hypot(g["position"][:, 0], g["position"][:, 2]) < 3.9
    assert (g["position"][inside, 1] > 0.1 - 1e-3).all()  # the slab's top face
    assert np.count_nonzero(g["age"] > 1.0) > 30
