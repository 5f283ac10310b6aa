"""Parity of the HIP path against the CPU oracle, through the C ABI.  Needs an MI355X."""
import math

import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair, assert_particles_match
import parity
from parity import trig_field_errors as parity_trig

pytestmark = pytest.mark.gpu
DT = np.float32(1.0 / 60.0)
SEED = workloads.SEED


@pytest.fixture()
def system():
    from bevy_firework_amd.system import ParticleSystem

    with ParticleSystem(device=0, seed=SEED) as ps:
        yield ps


def run(system, pair, frames, dt=DT, check_every=1, exact_all=False):
    for fr in range(frames):
        system.update(dt)
        pair.step_cpu(dt)
        if (fr + 1) % check_every == 0 or fr == frames - 1:
            pair.check(exact_all, f"frame {fr}")


def test_update_kats_golden(system):
    """the hand-derived single-particle vectors of tests/golden/update_kat.json, bit-exact"""
    import json
    import os

    from test_oracle_golden import _spawner_for, b, f, particle_from_kat

    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "update_kat.json")))
    for case in d["cases"]:
        sp = _spawner_for(case)
        sp.particle_settings[0].particles_destroyed = lambda dead: None
        h = system.spawn(sp)
        h.write_particles(0, particle_from_kat(case))
        system.step(f(case["dt_bits"]))
        if not case["alive"]:
            assert h.count(0) == 0
            dead = h.destroyed(0)
            assert len(dead) == 1 and b(dead["age"][0]) == case["out"]["age"]
            assert [b(x) for x in dead["position"][0]] == case["in"]["position"]
        else:
            got = h.particles(0)
            assert len(got) == 1
            for k, want in case["out"].items():
                gb = [b(x) for x in np.atleast_1d(got[0][k])]
                assert gb == (want if isinstance(want, list) else [want]), (case["name"], k)
        system.despawn(h)


def test_point_emitter_bit_exact(system):
    """no trig anywhere (Point, spread 0, zero angular velocity) -> the whole state is bit-exact"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.9), initial_scale=S.RandF32(0.5, 2.0),
                            scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]),
                            base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(30000.0),
                            initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (0.0, 1.0, 0.0), 0.0),
                            initial_velocity_radial=S.RandF32(0.0, 1.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), S.Transform((1.0, 2.0, 3.0)), seed=SEED)
    run(system, pair, 100, check_every=10, exact_all=True)
    assert pair.gpu.count(0) > 10000


def test_stress_test_example(system):
    """configs[0]: examples/stress_test.rs parameters at rate 50 000 (~49k live)"""
    spawner, tf = workloads.stress_test(rate=50000.0)
    pair = Pair(system, spawner, tf, seed=SEED)
    run(system, pair, 150, check_every=15)
    assert 48000 < pair.gpu.count(0) < 50001


def test_rotation_path(system):
    """angular velocity != 0: from_scaled_axis + quaternion product, tolerance 1e-5"""
    ps = S.ParticleSettings(lifetime=S.RandF32.constant(2.0), angular_acceleration=(0.1, 0.0, -0.2), angular_drag=0.3)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(5000.0),
                            emission_shape=S.EmissionShape.Sphere(0.5),
                            initial_rotation=(0.0, math.sin(0.4), 0.0, math.cos(0.4)),
                            initial_velocity=S.RandVec3(S.RandF32(0.5, 2.0), (0.6, 0.8, 0.0), 0.7),
                            initial_angular_velocity=S.RandVec3(S.RandF32(1.0, 9.0), (0.0, 0.6, 0.8), 0.5))
    tf = S.Transform((0.0, 1.0, 0.0), (math.sin(0.3), 0.0, 0.0, math.cos(0.3)))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), tf, seed=SEED, uid=5)
    pair.gpu.set_parent_velocity((0.5, 0.0, -0.25))
    pair.cpu.set_parent_velocity((0.5, 0.0, -0.25))
    run(system, pair, 150, check_every=25)


def test_oneshot_ondemand_and_finished(system):
    """OneShot disables itself (core.rs:397-400); OnDemand drains the queue (401-405); finished fires once"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.2, 0.4))
    one = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OneShot(3000))]),
               seed=SEED, uid=1)
    dem = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]),
               seed=SEED, uid=2)
    fired = []
    one.gpu.on_finished.append(lambda d: fired.append(d.handle))
    for fr in range(40):
        if fr in (0, 3, 4, 20):
            dem.queue(777 + fr)
        system.update(DT)
        one.step_cpu(DT)
        dem.step_cpu(DT)
        one.check(what=f"oneshot f{fr}")
        dem.check(what=f"ondemand f{fr}")
        assert one.gpu.active() == one.cpu.active() and dem.gpu.active() == dem.cpu.active()
        cpu_fin = one.cpu.poll_finished()
        assert (len(fired) == 1) == (cpu_fin or len(fired) == 1 and not cpu_fin)
    assert fired == [one.gpu.handle] and one.gpu.count(0) == 0 and not one.gpu.active()
    assert dem.gpu.active()


def test_compaction_order_ragged_lifetimes(system):
    """random lifetimes -> deaths scattered through the array; order must stay the reference's"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.05, 1.5))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(200000.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=9)
    run(system, pair, 120, check_every=20, exact_all=True)
    assert pair.gpu.count(0) > 100000


def test_variable_dt_switches_between_forecast_and_lookback(system):
    """constant dt -> survivor forecast; a changed dt or a touched state -> decoupled look-back; the
    particle order and state must be identical either way (several tiles, deaths everywhere)"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.05, 0.8), linear_drag=0.3)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(120000.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 3.0), (0.0, 1.0, 0.0), 0.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=21)
    dts = [1 / 60] * 25 + [1 / 30, 1 / 60, 1 / 60, 1 / 144, 1 / 144, 1 / 144, 0.0, 1 / 60] + [1 / 60] * 20 + [0.011, 0.012, 0.013]
    for i, dt in enumerate(dts):
        dt = np.float32(dt)
        system.update(dt)
        pair.step_cpu(dt)
        if i % 4 == 3 or i > 24:
            pair.check(exact_all=True, what=f"frame {i} dt={dt}")
    # touching the state between two equal-dt frames must not reuse the stale forecast
    parts = pair.cpu.particles(0)[::3].copy()
    pair.gpu.write_particles(0, parts)
    pair.cpu.write_particles(0, parts)
    for i in range(6):
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(exact_all=True, what=f"after write {i}")
    assert pair.gpu.count(0) > 20000


def test_constant_colour_planes_survive_caller_written_colours(system):
    """a one-key gradient's colour plane is filled once and not written by the update (FwOutWin::wr5 / wr6); particles
    the caller writes may carry ANY colour, in slots later reused by new particles: every colour must still be the
    reference's (the gradient sampled each update, core.rs:640-646), on both buffers, with and without the forecast"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.05, 0.5), base_color=S.FireworkGradient.constant((0.25, 0.5, 0.75, 1.0)))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(60000.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 3.0), (0.0, 1.0, 0.0), 0.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=77)
    run(system, pair, 20, check_every=20, exact_all=True)
    rng = np.random.default_rng(5)
    for rep in range(3):
        parts = pair.cpu.particles(0).copy()
        parts = np.concatenate([parts, parts, parts])[: 40000 + 1000 * rep]   # more than are live: slots past the survivors
        parts["base_color"] = rng.uniform(0.0, 9.0, size=(len(parts), 4)).astype(np.float32)
        parts["emissive_color"] = rng.uniform(0.0, 9.0, size=(len(parts), 4)).astype(np.float32)
        parts["age"][::2] = parts["lifetime"][::2]                             # half of them die in the next update
        pair.gpu.write_particles(0, parts)
        pair.cpu.write_particles(0, parts)
        got = pair.gpu.particles(0)                                            # until then they read back as written
        assert np.array_equal(got["base_color"], parts["base_color"]) and np.array_equal(got["emissive_color"], parts["emissive_color"])
        for i in range(5 + rep):
            system.update(DT)
            pair.step_cpu(DT)
            pair.check(exact_all=True, what=f"rep {rep} after write {i}")
    assert pair.gpu.count(0) > 10000


def test_two_types_two_emitters_and_modifier(system):
    p0 = S.ParticleSettings(lifetime=S.RandF32(0.5, 0.7), linear_drag=0.5)
    p1 = S.ParticleSettings(lifetime=S.RandF32.constant(0.25), acceleration=(0.0, 1.0, 0.0),
                            scale_curve=S.FireworkCurve.uneven_samples([(0.0, 1.0), (0.8, 1.2), (1.0, 0.0)]))
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(9000.0),
                            emission_shape=S.EmissionShape.Circle((0.0, 0.0, 1.0), 2.0))
    e1 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.CountOverDuration(500.0, 0.5, 0.2, 0.9))
    e2 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(1234.0),
                            initial_velocity_radial=S.RandF32(1.0, 2.0), emission_shape=S.EmissionShape.Sphere(1.0))
    pair = Pair(system, S.ParticleSpawner([p0, p1], [e0, e1, e2]), S.Transform((0, 0, 0)), seed=SEED, uid=3,
                modifier=S.EffectModifier(scale=2.0, speed=0.5))
    run(system, pair, 90, check_every=10)


def test_nested_emission(system):
    """configs[3] shape at test size: sparks -> smoke, parent-major child order, last_emitted_age plane"""
    spawner, tf = workloads.nested(spark_rate=3000.0, smoke_per_spark=20.0)
    pair = Pair(system, spawner, tf, seed=SEED, uid=11)
    for fr in range(150):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 15 == 14:
            pair.check(what=f"nested f{fr}")
            assert np.array_equal(pair.gpu.last_emitted(0, 1), pair.cpu.last_emitted(0, 1))
    c = pair.gpu.counts()
    assert c[0] > 5000 and c[1] > 50000


def test_settings_change_resets(system):
    """Changed<ParticleSpawner> -> sync_spawner_data drops particles and restarts clocks (core.rs:343-365)"""
    spawner, tf = workloads.stress_test(rate=20000.0)
    pair = Pair(system, spawner, tf, seed=SEED, uid=4)
    run(system, pair, 30, check_every=30)
    pair.gpu.update_settings(spawner)
    pair.cpu.reset()
    assert pair.gpu.counts() == [0]
    run(system, pair, 30, check_every=30)


def test_empty_and_tiny(system):
    """empty spawner, single particle, dt = 0"""
    ps = S.ParticleSettings(lifetime=S.RandF32.constant(1.0))
    pair = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]),
                seed=SEED, uid=6)
    run(system, pair, 3)
    pair.queue(1)
    run(system, pair, 5, exact_all=True)
    run(system, pair, 2, dt=np.float32(0.0), exact_all=True)
    assert pair.gpu.count(0) == 1


def test_instances_aabb_and_invalid_settings(system):
    spawner, tf = workloads.stress_test(rate=20000.0)
    pair = Pair(system, spawner, tf, seed=SEED, uid=7)
    run(system, pair, 40, check_every=40)
    inst, parts = pair.gpu.instances(0), pair.gpu.particles(0)
    assert np.array_equal(inst["position"], parts["position"]) and np.array_equal(inst["scale"], parts["scale"])
    assert np.array_equal(inst["rotation"], parts["rotation"])
    assert np.array_equal(inst["base_color"], parts["base_color"])
    assert np.array_equal(inst["emissive_color"], parts["emissive_color"])
    cpu_parts = pair.cpu.particles(0)  # and against the oracle's own state: the record is {pos, scale, rot, base, emissive}
    for k in ("scale", "base_color", "emissive_color"):
        assert np.array_equal(inst[k], cpu_parts[k]), k
    any_g, mn_g, mx_g = pair.gpu.aabb()
    cp = pair.cpu.particles(0)
    assert any_g
    assert np.array_equal(mn_g, (parts["position"] - parts["scale"][:, None]).min(axis=0))
    assert np.array_equal(mx_g, (parts["position"] + parts["scale"][:, None]).max(axis=0))
    any_c, mn_c, mx_c = pair.cpu.aabb()
    assert np.allclose(mn_g, mn_c, rtol=1e-5, atol=1e-5) and np.allclose(mx_g, mx_c, rtol=1e-5, atol=1e-5)
    from bevy_firework_amd.system import FwError

    bad = S.ParticleSpawner([S.ParticleSettings()], [S.EmissionSettings(particle_index=3)])
    with pytest.raises(FwError):  # index panic core.rs:392 -> FW_EINVAL
        system.spawn(bad)


def test_capacity_growth_on_demand(system):
    """Vec growth: bursts far beyond the derived capacity (4096 for an OnDemand entry) must grow the device
    buffers without losing or reordering particles"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.2, 0.6))
    pair = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]),
                seed=SEED, uid=31)
    for fr in range(30):
        if fr in (0, 1, 5, 6, 7, 20):
            pair.queue(30000 + 1000 * fr)
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 5 == 4:
            pair.check(exact_all=True, what=f"growth f{fr}")
    assert pair.gpu.count(0) > 25000


def test_nested_overflow_reports_capacity(system):
    """children beyond the child type's device capacity are dropped and reported, never written out of bounds"""
    from bevy_firework_amd.system import FwError

    spawner, tf = workloads.nested(spark_rate=3000.0, smoke_per_spark=20.0)
    spawner.particle_settings[1].capacity = 2048
    h = system.spawn(spawner, tf, uid=12)
    for _ in range(80):
        system.update(DT)
    with pytest.raises(FwError) as e:
        h.counts()
    assert e.value.status == -4  # FW_ECAPACITY
    c = h.counts()  # flag is cleared once reported; state stays consistent
    assert c[1] <= 2048 and c[0] > 3000


@pytest.mark.parametrize("env", [{"FW_FORECAST": "0", "FW_SPIN_LIMIT": "0"}, {"FW_UPDATE_MODE": "split"},
                                 {"FW_STREAM": "0"}, {"FW_STATIC_NEW": "0"}])
def test_alternative_prefix_paths(monkeypatch, env):
    """the other ways a tile can obtain its output offset must give the same particles: the look-back's
    recount fallback (spin limit 0 forces it wherever a predecessor has not published yet), the three-launch
    split mode, forecast frames on the count-park-store kernel, and new particles counted + looked up even when
    the host could prove that all of them survive the step"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from bevy_firework_amd.system import ParticleSystem

    with ParticleSystem(device=0, seed=SEED) as ps_:
        ps = S.ParticleSettings(lifetime=S.RandF32(0.05, 1.2), linear_drag=0.3)
        es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(150000.0),
                                initial_velocity=S.RandVec3(S.RandF32(0.0, 3.0), (0.0, 1.0, 0.0), 0.0))
        pair = Pair(ps_, S.ParticleSpawner([ps], [es]), seed=SEED, uid=41)
        for fr in range(80):
            ps_.update(DT)
            pair.step_cpu(DT)
            if fr % 20 == 19:
                pair.check(exact_all=True, what=f"{env} f{fr}")
        assert pair.gpu.count(0) > 60000


def test_new_particles_that_die_in_their_first_step(system):
    """lifetimes below dt: a particle can be spawned and destroyed by the same frame's update (core.rs:437-469 then
    594-599), so the slots of the new particles are not static; frames with a tiny dt (all survive) in between
    switch to the static-slot path and back"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.001, 0.04), linear_drag=0.2)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(400000.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 2.0), (0.0, 1.0, 0.0), 0.3))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=57)
    dts = [1 / 60] * 12 + [1 / 2000] * 3 + [1 / 60] * 6 + [1 / 2000, 1 / 60, 1 / 2000, 1 / 60] + [1 / 30] * 4
    for i, dt in enumerate(dts):
        dt = np.float32(dt)
        system.update(dt)
        pair.step_cpu(dt)
        pair.check(what=f"frame {i} dt={dt}")  # cone spread -> sin/cos: trig tolerance on position / velocity
    assert pair.gpu.count(0) > 2000


def test_lifetime_window_bound_long_run(system):
    """the host bounds a segment's live count by the spawns of the last `max lifetime` of simulated time and sizes the
    update grid from it; irregular steps (zero, long, short), a burst of long-lived particles from a second entry and
    a rewrite of the particle state must never leave a live particle outside the grid"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.2, 2.0), linear_drag=0.1)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(40000.0))
    burst = S.EmissionSettings(emission_pacing=S.EmissionPacing.OneShot(30000))
    pair = Pair(system, S.ParticleSpawner([ps], [es, burst]), seed=SEED, uid=63)
    rng = np.random.default_rng(5)
    pattern = [1 / 60] * 40 + [0.0, 0.0, 0.25, 1 / 240, 1 / 240, 0.1, 1 / 60, 0.5, 1 / 60, 1 / 60]
    frames = 0
    for rep in range(4):
        for dt in pattern + list(rng.uniform(0.001, 0.05, size=30)):
            dt = np.float32(dt)
            system.update(dt)
            pair.step_cpu(dt)
            frames += 1
            if frames % 40 == 0:
                pair.check(exact_all=True, what=f"frame {frames}")
        if rep == 1:  # ages and lifetimes rewritten by the caller: the window no longer applies
            parts = pair.cpu.particles(0)[::2].copy()
            parts["lifetime"] = parts["lifetime"] + np.float32(3.0)
            pair.gpu.write_particles(0, parts)
            pair.cpu.write_particles(0, parts)
    pair.check(exact_all=True, what="end")
    assert pair.gpu.count(0) > 20000


def test_attached_instances_are_the_packed_records(system):
    """render hand-off fused into the update (render.rs:95-115, :403): the records the update kernel writes into an
    attached device buffer must be byte-identical to what the packing pass produces from the stored planes, on
    forecast frames (streaming schedule), on changed-dt frames (look-back schedule) and with new particles in the
    frame; records beyond the buffer's capacity are dropped, nothing is written past it"""
    import torch

    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 1.1), initial_scale=S.RandF32(0.02, 0.08), linear_drag=0.2,
                            scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]),
                            base_color=S.FireworkGradient.even_samples([(1.0, 1.0, 1.0, 1.0), (0.0, 0.0, 0.0, 0.0)]),
                            emissive_color=S.FireworkGradient.even_samples([(4.0, 2.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)]))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(90000.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 4.0), (0.0, 1.0, 0.0), 0.4),
                            initial_angular_velocity=S.RandVec3(S.RandF32(0.0, 5.0), (0.0, 1.0, 0.0), 0.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=71)
    cap = 120000
    guard = 64
    buf = torch.full(((cap + guard) * 16,), float("nan"), dtype=torch.float32, device="cuda")
    pair.gpu.attach_instances(buf.data_ptr(), cap)
    dts = [1 / 60] * 30 + [1 / 45, 1 / 60, 1 / 60, 1 / 90] + [1 / 60] * 30
    for i, dt in enumerate(dts):
        dt = np.float32(dt)
        system.update(dt)
        pair.step_cpu(dt)
        if i in (0, 5, 29, 30, 31, 33, 34, 40, len(dts) - 1):
            n = pair.gpu.count(0)
            ref = pair.gpu.instances(0)  # packing pass over the planes the same update stored
            got = buf[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
            assert np.array_equal(got, ref.view(np.uint32).reshape(n, 16)), f"frame {i}: attached records differ from packed ones"
            assert bool(torch.isnan(buf[cap * 16:]).all()), "wrote past the attached buffer"
    pair.check(what="state after attached frames")
    assert pair.gpu.count(0) > 40000
    # ... and the records against the ORACLE's state (not only against the packing pass of the same library): scale and
    # both colours involve no libm call, so they are bit-identical; this gradient pair has a non-constant emissive
    n = pair.gpu.count(0)
    rec = buf[: n * 16].cpu().numpy().view(S.INSTANCE_DTYPE).reshape(n)
    cp = pair.cpu.particles(0)
    for k in ("scale", "base_color", "emissive_color"):
        assert np.array_equal(rec[k], cp[k]), k
    ok, _ = parity_trig(rec["position"], cp["position"])
    assert ok.all()
    # a buffer smaller than the live count: the first `small` records, nothing beyond
    small = 10000
    buf2 = torch.full(((small + guard) * 16,), float("nan"), dtype=torch.float32, device="cuda")
    pair.gpu.attach_instances(buf2.data_ptr(), small)
    for _ in range(3):
        system.update(DT)
        pair.step_cpu(DT)
    ref = pair.gpu.instances(0)
    assert np.array_equal(buf2[: small * 16].cpu().numpy().view(np.uint32), ref.view(np.uint32).reshape(-1, 16)[:small].ravel())
    assert bool(torch.isnan(buf2[small * 16:]).all())
    pair.gpu.attach_instances(0, 0)  # detach
    buf2.fill_(float("nan"))
    system.update(DT)
    pair.step_cpu(DT)
    assert bool(torch.isnan(buf2).all())
    pair.check(what="after detach")


def test_attached_instances_replace_the_scale_and_colour_planes(system):
    """with an instance buffer attached, the records carry scale and both colours:
    the update stops storing the three planes that would duplicate them (FW_TYPE_DERIVED) and every reader evaluates them
    from age / lifetime / initial_scale instead -- particles, the packing pass, destroyed records, the AABB must be what
    they were; detaching fills the planes again.  A one-lifetime type (a FIFO ring where rings are on) and a lifetime range."""
    import torch

    grads = dict(scale_curve=S.FireworkCurve.even_samples([1.5, 2.0, 0.5]),  # (1.5 at age 0: scale != initial_scale there)
                 base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT),
                 emissive_color=S.FireworkGradient.even_samples([(4.0, 2.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)]))
    cap = 65536
    t0 = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), initial_scale=S.RandF32(0.02, 0.08), capacity=cap,
                            particles_destroyed=lambda dead: None, **grads)
    t1 = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.9), initial_scale=S.RandF32(0.02, 0.08), capacity=cap, linear_drag=0.4,
                            particles_destroyed=lambda dead: None, **grads)
    es = [S.EmissionSettings(particle_index=t, emission_pacing=S.EmissionPacing.rate(50000.0),
                             initial_velocity=S.RandVec3(S.RandF32(0.0, 4.0), (0.0, 1.0, 0.0), 0.0)) for t in (0, 1)]
    pair = Pair(system, S.ParticleSpawner([t0, t1], es), S.Transform((0.0, 1.0, 0.0)), seed=SEED, uid=73)
    bufs = [torch.full((cap * 16,), float("nan"), dtype=torch.float32, device="cuda") for _ in (0, 1)]
    before = [pair.gpu.update_path(t)[1] for t in (0, 1)]
    pair_path0 = pair.gpu.update_path(0)[0]
    D = parity.planes_left_to_readers()  # (round 6: the planes are not stored before the attach either)

    def check(what):
        pair.check(exact_all=True, what=what)
        for t in (0, 1):
            assert_particles_match(pair.gpu.destroyed(t), pair.cpu.destroyed(t), True, f"{what}: destroyed records, type {t}")
        any_, mn, mx = pair.gpu.aabb()
        parts = [p for p in (pair.cpu.particles(t) for t in (0, 1)) if len(p)]
        lo = np.min([(p["position"] - p["scale"][:, None]).min(axis=0) for p in parts], axis=0)
        hi = np.max([(p["position"] + p["scale"][:, None]).max(axis=0) for p in parts], axis=0)
        assert any_ and np.array_equal(mn, lo) and np.array_equal(mx, hi), what

    for fr in range(130):
        if fr == 40:
            for t in (0, 1):
                pair.gpu.attach_instances(bufs[t].data_ptr(), cap, particle_type=t)
            attached = [pair.gpu.update_path(t) for t in (0, 1)]
            # 4 B of scale + 16 B per non-constant gradient no longer stored (the path of type 1 may have changed: a range ring
            # continues on the compacting path once records are wanted)
            # (a range ring continues on the compacting path once records are wanted: 4 B more for the lifetime plane it rewrites)
            # (+ the 64-byte record itself, which the update now writes per survivor)
            # (a small type -- fw_k_small.hip -- keeps its kernel: the same layout, the same bytes)
            # (round 6, component planes: a ring moves velocity as 12 bytes each way and reads initial_scale -- 4 B -- only where somebody
            # evaluates the scale: with stored planes, or with records; a range ring that continues on the compacting path moves the
            # 16 bytes of Q1 and the lifetime plane both ways again: 60 -> 72)
            if attached[0][0] == pair_path0:
                adj = (8 if pair_path0 == "range" else 4) * D if pair_path0 in ("fifo", "range") else 0
            else:
                adj = 16 - 8 * (1 - D)  # (range ring, planes left to the readers: 56; with stored planes + lifetime + initial_scale)
            assert attached[0][1] == before[0] - 36 * (1 - D) + 64 + adj, (before, attached)
        if fr == 100:
            for t in (0, 1):
                pair.gpu.attach_instances(0, 0, particle_type=t)
            assert pair.gpu.update_path(0)[1] == attached[0][1] + 36 * (1 - D) - 64 - ((8 if attached[0][0] == "range" else 4) * D if attached[0][0] in ("fifo", "range") else 0)
            check("right after detaching")
        dt = np.float32(0.55 if fr == 70 else DT)  # frame 70: longer than type 0 lives -- born and destroyed in one frame
        system.update(dt)
        pair.step_cpu(dt)
        if fr % 10 == 9 or fr in (40, 41, 70, 71, 100, 101):
            check(f"frame {fr}")
            if 40 <= fr < 100:
                for t in (0, 1):
                    n = pair.gpu.count(t)
                    got = bufs[t][: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                    assert np.array_equal(got, pair.gpu.instances(t).view(np.uint32).reshape(n, 16)), f"frame {fr} type {t}"
                    rec = bufs[t][: n * 16].cpu().numpy().view(S.INSTANCE_DTYPE).reshape(n)
                    for k in ("scale", "base_color", "emissive_color"):
                        assert np.array_equal(rec[k], pair.cpu.particles(t)[k]), (fr, t, k)
    assert all(n > 15000 for n in pair.gpu.counts())


def test_windowed_attach_is_the_plain_attach_where_the_list_starts_at_record_zero(system):
    """fw_spawner_attach_instances_window on whatever path a type is on: the live records are buffer[first : first + count];
    first is 0 except on a range ring, where it is the number of particles the step destroyed"""
    import torch

    t0 = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), initial_scale=S.RandF32(0.02, 0.08), capacity=32768,
                            particles_destroyed=lambda dead: None,
                            base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
    t1 = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.9), initial_scale=S.RandF32(0.02, 0.08), capacity=32768, linear_drag=0.4,
                            particles_destroyed=lambda dead: None,
                            scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]))
    es = [S.EmissionSettings(particle_index=t, emission_pacing=S.EmissionPacing.rate(30000.0),
                             initial_velocity=S.RandVec3(S.RandF32(0.0, 4.0), (0.0, 1.0, 0.0), 0.0)) for t in (0, 1)]
    pair = Pair(system, S.ParticleSpawner([t0, t1], es), seed=SEED, uid=74)
    bufs = [torch.full((32768 * 16,), float("nan"), dtype=torch.float32, device="cuda") for _ in (0, 1)]
    paths = [pair.gpu.update_path(t)[0] for t in (0, 1)]
    for t in (0, 1):
        pair.gpu.attach_instances_window(bufs[t].data_ptr(), 32768, particle_type=t)
    # nobody changes path for a windowed buffer (a type updated by a wave or a workgroup of fw_k_update_small keeps it: its INST
    # instantiation writes the records)
    assert [pair.gpu.update_path(t)[0] for t in (0, 1)] == paths
    for fr in range(90):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 10 == 9:
            pair.check(exact_all=True, what=f"frame {fr}")
            for t in (0, 1):
                first, n = pair.gpu.instance_window(t)
                assert n == pair.cpu.count(t)
                assert first == (len(pair.cpu.destroyed(t)) if paths[t] == "range" else 0), (fr, t, first, paths)
                got = bufs[t][first * 16: (first + n) * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                assert np.array_equal(got, pair.gpu.instances(t).view(np.uint32).reshape(n, 16)), (fr, t)
    assert all(c > 10000 for c in pair.gpu.counts())


def test_attached_instances_of_a_nested_child_type(system):
    """frames with Nested entries spawn the children before the update of the same frame (plugin.rs:46-60), so the
    records the update writes for the child type cover the new children too; sparks -> smoke, both types attached"""
    import torch

    spawner, tf = workloads.nested(spark_rate=4000.0, smoke_per_spark=20.0)
    pair = Pair(system, spawner, tf, seed=SEED, uid=13)
    cap = [20000, 200000]
    bufs = [torch.full((c * 16,), float("nan"), dtype=torch.float32, device="cuda") for c in cap]
    for t in (0, 1):
        pair.gpu.attach_instances(bufs[t].data_ptr(), cap[t], particle_type=t)
    for fr in range(90):
        system.update(DT)
        pair.step_cpu(DT)
        if fr in (0, 3, 30, 60, 89):
            for t in (0, 1):
                n = pair.gpu.count(t)
                ref = pair.gpu.instances(t)
                got = bufs[t][: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                assert np.array_equal(got, ref.view(np.uint32).reshape(n, 16)), f"frame {fr} type {t}"
    pair.check(what="nested with attached buffers")
    assert pair.gpu.count(1) > 50000


def test_live_count_ring(system):
    """per-frame live totals written by the update kernel into a caller-owned device ring (RCCL feed)"""
    import torch

    spawner, tf = workloads.stress_test(rate=30000.0)
    pairs = [Pair(system, spawner, tf, seed=SEED, uid=50 + i) for i in range(3)]  # three segments
    ring = torch.zeros(8, dtype=torch.int64, device="cuda")
    system.update(DT)
    for p in pairs:
        p.step_cpu(DT)
    system.live_count_ring(ring.data_ptr(), 8)
    want = []
    for fr in range(21):
        system.step(DT)
        for p in pairs:
            p.step_cpu(DT)
        want.append(sum(p.cpu.count(0) for p in pairs))
    system.synchronize()
    got = ring.cpu().tolist()
    for k in range(21 - 7, 21):  # the slot of frame 21 (k % 8 == 5) was zeroed for the next frame
        assert got[k % 8] == want[k], (k, got, want[-8:])
    assert system.live_count() == want[-1]
    system.live_count_ring(0, 0)


def test_settings_churn_does_not_leak_slots(system):
    """Changed<ParticleSpawner> every few frames and spawn/despawn cycles reuse their table slots"""
    spawner, tf = workloads.stress_test(rate=8000.0)
    pair = Pair(system, spawner, tf, seed=SEED, uid=60)
    for cycle in range(40):
        for _ in range(3):
            system.update(DT)
            pair.step_cpu(DT)
        pair.gpu.update_settings(spawner)
        pair.cpu.reset()
        h = system.spawn(spawner, tf, uid=100 + cycle)
        system.update(DT)
        pair.step_cpu(DT)
        system.despawn(h)
    for _ in range(20):
        system.update(DT)
        pair.step_cpu(DT)
    pair.check(what="after churn")


def test_types_that_cannot_turn_keep_no_rotation_plane(system):
    """FW_TYPE_NOSPIN: every entry feeding the type spawns with zero angular velocity and the same rotation, the type has no
    angular acceleration -> rotation is that rotation for every particle for ever and the update moves no rotation
    bytes; what the ABI shows (particles, instance records, destroyed records, children of such parents) must still be
    the reference's.  A negative-zero component in the rotation, two feeding entries with the same rotation, and the
    ways out of the mode: particles written by the caller (they may spin) and a non-finite step."""
    rot = (0.0, math.sin(0.4), -0.0, math.cos(0.4))
    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.9), linear_drag=0.3, particles_destroyed=lambda dead: None,
                            base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
    e0 = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(30000.0), initial_rotation=rot,
                            initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (0.0, 1.0, 0.0), 0.0))
    e1 = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(7000.0), initial_rotation=rot,
                            emission_shape=S.EmissionShape.Sphere(0.5))
    pair = Pair(system, S.ParticleSpawner([ps], [e0, e1]), S.Transform((1.0, 2.0, 3.0)), seed=SEED, uid=61)
    mode, moved, _ = pair.gpu.update_path(0)
    # constant emissive; no rotation plane, and the lifetimes in a 4-byte plane instead of Q3: 164 - 16 - 32 - 32 + 8
    # (round 6, parity.planes_left_to_readers: nor the scale and the base colour -- 20 B more)
    assert moved == (92 - 20 * parity.planes_left_to_readers() if mode in ("general", "small") else None) or mode in ("fifo", "range")
    for fr in range(60):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 10 == 9:
            pair.check(what=f"frame {fr}")
            assert_particles_match(pair.gpu.destroyed(0), pair.cpu.destroyed(0), False, f"destroyed frame {fr}")
            inst, parts = pair.gpu.instances(0), pair.gpu.particles(0)
            assert np.array_equal(inst["rotation"], parts["rotation"]) and np.array_equal(parts["rotation"], pair.cpu.particles(0)["rotation"])
    # the caller writes particles that DO spin: the plane is back, rotations evolve as the reference's
    rng = np.random.default_rng(4)
    parts = pair.cpu.particles(0).copy()
    parts["angular_velocity"] = rng.uniform(-6.0, 6.0, size=(len(parts), 3)).astype(np.float32)
    q = rng.normal(size=(len(parts), 4))
    parts["rotation"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    pair.gpu.write_particles(0, parts)
    pair.cpu.write_particles(0, parts)
    assert pair.gpu.update_path(0)[1] == 164 - 16
    for fr in range(40):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 8 == 7:
            pair.check(what=f"after write {fr}")
    assert pair.gpu.count(0) > 10000


def test_two_entries_with_different_rotations_keep_the_plane(system):
    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.9))
    e0 = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(9000.0), initial_rotation=(0.0, math.sin(0.4), 0.0, math.cos(0.4)))
    e1 = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(9000.0), initial_rotation=(math.sin(0.2), 0.0, 0.0, math.cos(0.2)))
    pair = Pair(system, S.ParticleSpawner([ps], [e0, e1]), seed=SEED, uid=62)
    # both colours constant; all four state planes kept (ring: read only); the scale plane left to the readers from round 6 on
    # (ring, round 6: Q0 + rotation as float4, velocity + angular velocity as 12 bytes in; Q0 + velocity out; + the lifetime of a range ring,
    # + initial_scale where the scale plane is stored)
    D = parity.planes_left_to_readers()
    assert pair.gpu.update_path(0)[1] in (164 - 32 - 4 * D, 56 + 28 + 4 * (1 - D), 56 + 28 + 8 * (1 - D))
    run(system, pair, 60, check_every=12, exact_all=True)
    assert len(np.unique(pair.gpu.particles(0)["rotation"], axis=0)) == 2


def test_a_non_finite_step_brings_the_rotation_plane_back(system):
    """0 * inf = NaN: an angular velocity of zero does not stay zero through a non-finite dt (core.rs:648-650), so the type
    stops being FW_TYPE_NOSPIN first; nothing may crash or hang, and the state is the reference's NaNs"""
    ps = S.ParticleSettings(lifetime=S.RandF32(5.0, 9.0))
    pair = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OneShot(5000))]), seed=SEED, uid=63)
    run(system, pair, 5, exact_all=True)
    before = pair.gpu.update_path(0)[1]
    D = parity.planes_left_to_readers()  # (the 4-byte scale plane is left to the readers)
    if pair.gpu.update_path(0)[0] == "range":  # in place: position+age and velocity (12 B) in and out
        before += 164 - 32 - 56 - 4 * D - (28 + 28 + 8 * (1 - D))  # (+ lifetime and initial_scale where the scale plane is stored)
    system.update(np.float32("nan"))
    pair.step_cpu(np.float32("nan"))
    assert before == 164 - 32 - 56 - 4 * D and pair.gpu.update_path(0)[1] == 164 - 32 - 4 * D and pair.gpu.update_path(0)[0] in ("general", "small")
    g, c = pair.gpu.particles(0), pair.cpu.particles(0)
    assert len(g) == len(c) == 5000
    for f in ("age", "position", "angular_velocity", "rotation"):
        assert np.array_equal(np.isnan(g[f]), np.isnan(c[f])), f


def test_a_frame_that_cannot_be_enqueued_changes_nothing(system):
    """spawn_particles is all-or-nothing per frame: when one entry asks for more particles than a frame can hold
    (2^30), fw_step fails BEFORE anything is enqueued and every clock, queue, RNG serial, spawn total and lifetime
    window the earlier entries of the frame had already advanced is put back -- also those of the other spawners.
    The neighbour keeps matching the oracle afterwards, the failing spawner's state is what it was."""
    from bevy_firework_amd.system import FwError

    nb = Pair(system, S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32(0.2, 0.5))],
                                        [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(9000.0))]), seed=SEED, uid=71)
    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.6))
    bad = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(5000.0)),
                                                S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]), seed=SEED, uid=72)
    for fr in range(12):
        system.update(DT)
        nb.step_cpu(DT)
        bad.step_cpu(DT)
    nb.check(exact_all=True, what="before")
    bad.check(exact_all=True, what="before")
    before = bad.gpu.particles(0).copy()
    bad.gpu.queue_particles((1 << 30) + 5)
    for _ in range(3):  # the queue is part of what is put back: the frame keeps failing, and keeps changing nothing
        with pytest.raises(FwError) as e:
            system.update(DT)
        assert e.value.status == -4  # FW_ECAPACITY
    after = bad.gpu.particles(0)
    assert len(after) == len(before) and all(np.array_equal(after[f], before[f]) for f in before.dtype.names)
    nb.check(exact_all=True, what="after the failed frames")
    system.despawn(bad.gpu)  # the entity goes away; the rest of the world runs on as if those frames had never been asked for
    for fr in range(40):
        system.update(DT)
        nb.step_cpu(DT)
        if fr % 10 == 9:
            nb.check(exact_all=True, what=f"neighbour, frame {fr} after")
