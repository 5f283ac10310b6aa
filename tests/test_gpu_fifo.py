"""The in-place FIFO ring path (fw_k_update_fifo: particle types with one lifetime value) against the CPU oracle: the ring
wraps, grows while wrapped, falls back to the general path when its premise breaks, and everything the ABI can observe
(order, state, destroyed records, instance records, AABB) stays the reference's.  Needs an MI355X."""
import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair, assert_particles_match
import parity

pytestmark = pytest.mark.gpu
DT = np.float32(1.0 / 60.0)
SEED = workloads.SEED


# every test of the file with four-round ring tiles at any size (FW_FIFO_SMALL=0) and with one-round tiles for small launches
# (the product's default: below 384 four-round tiles in all) -- fw_k_update_fifo<.., TR = 4 / 1>
@pytest.fixture(params=["four-round tiles", "one-round tiles when small"])
def tile_rounds(request, monkeypatch, fw_path):
    if request.param == "four-round tiles":
        monkeypatch.setenv("FW_FIFO_SMALL", "0")
    else:
        if fw_path != "fifo":
            pytest.skip("no FIFO rings on this path")
        monkeypatch.delenv("FW_FIFO_SMALL", raising=False)
    return request.param


@pytest.fixture()
def system(fw_path, tile_rounds):
    from bevy_firework_amd.system import ParticleSystem

    with ParticleSystem(device=0, seed=SEED) as ps:
        ps.path = fw_path
        yield ps


def _expect_path(system, pair, t=0, fifo=True):
    want = ("fifo",) if (fifo and system.path == "fifo") else (("range", "general") if system.path == "range" else ("general", "small"))
    assert pair.gpu.update_path(t)[0] in want, (pair.gpu.update_path(t), want)


def _ring_settings(**kw):
    base = dict(lifetime=S.RandF32.constant(0.25), initial_scale=S.RandF32(0.5, 2.0), linear_drag=0.2,
                scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]),
                base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
    base.update(kw)
    return S.ParticleSettings(**base)


def test_ring_wraps_many_times_bit_exact(system):
    """a small ring (capacity 4096) turns over every ~0.3 s: the head crosses the end of the buffer again and again,
    tiles and the new-particle groups straddle it; no trig anywhere -> the whole state bit-exact every frame"""
    ps = _ring_settings(capacity=4096)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(11000.0),
                            initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (0.0, 1.0, 0.0), 0.0),
                            initial_velocity_radial=S.RandF32(0.0, 1.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), S.Transform((1.0, 2.0, 3.0)), seed=SEED, uid=3)
    _expect_path(system, pair)
    for fr in range(200):
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(exact_all=True, what=f"frame {fr}")
    assert 2500 < pair.gpu.count(0) < 3000
    _expect_path(system, pair)


def test_irregular_dt_needs_no_forecast(system):
    """any sequence of dt >= 0 (the plugin's wall-clock delta, plugin.rs:26-31): the host replays the cohort ages, so the
    destroyed count is exact every frame; zero steps and a step longer than the lifetime (everything dies, the particles
    spawned in that very frame included) are part of it"""
    ps = _ring_settings(lifetime=S.RandF32.constant(0.2), particles_destroyed=lambda dead: None)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(40000.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=4)
    _expect_path(system, pair)
    rng = np.random.default_rng(11)
    dts = list(rng.uniform(0.0, 0.03, size=60)) + [0.0, 0.0, 0.016, 0.25, 0.016, 0.016, 0.5, 0.001] + list(rng.uniform(0.001, 0.02, size=30))
    for fr, dt in enumerate(dts):
        dt = np.float32(dt)
        system.update(dt)
        pair.step_cpu(dt)
        pair.check(exact_all=True, what=f"frame {fr} dt={dt}")
        assert_particles_match(pair.gpu.destroyed(0), pair.cpu.destroyed(0), True, f"destroyed frame {fr}")
    _expect_path(system, pair)
    assert pair.gpu.count(0) > 1000


def test_growth_while_wrapped(system):
    """bursts far beyond the capacity while the head sits in the middle of the buffer: the ring is unwrapped into the
    larger allocation (Vec growth, core.rs:523) without losing or reordering a particle"""
    ps = _ring_settings(lifetime=S.RandF32.constant(0.3), capacity=4096)
    pair = Pair(system, S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]),
                seed=SEED, uid=31)
    _expect_path(system, pair)
    for fr in range(70):
        pair.queue(150 if fr % 3 else 900)          # keeps the ring turning
        if fr in (25, 26, 40, 58):
            pair.queue(9000 + 1500 * (fr % 7))      # ... and bursts through the capacity
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(exact_all=True, what=f"growth f{fr}")
    _expect_path(system, pair)
    assert pair.gpu.count(0) > 3000


def test_caller_written_particles_end_the_mode(system):
    """fw_spawner_write_particles may hand over any ages and lifetimes: the segment continues on the general path, from
    exactly the state the caller wrote (ring unwrapped first)"""
    ps = _ring_settings(capacity=8192)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(20000.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=8)
    for fr in range(40):
        system.update(DT)
        pair.step_cpu(DT)
    pair.check(exact_all=True, what="before write")
    _expect_path(system, pair)
    parts = pair.cpu.particles(0)[::2].copy()
    parts["lifetime"] = np.linspace(0.05, 0.6, len(parts)).astype(np.float32)   # no longer one lifetime
    pair.gpu.write_particles(0, parts)
    pair.cpu.write_particles(0, parts)
    assert pair.gpu.update_path(0)[0] in ("general", "small")
    for fr in range(40):
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(exact_all=True, what=f"after write {fr}")
    assert pair.gpu.count(0) > 3000


def test_negative_dt_ends_the_mode(system):
    """ages must never decrease for the oldest-first order to hold: a negative step moves the type to the general path,
    which then does whatever the reference's arithmetic does with it"""
    ps = _ring_settings(lifetime=S.RandF32.constant(0.3))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(15000.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=9)
    dts = [1 / 60] * 30 + [-1 / 120, 1 / 60, -1 / 60] + [1 / 60] * 30
    for fr, dt in enumerate(dts):
        if fr == 30:
            _expect_path(system, pair)
        dt = np.float32(dt)
        system.update(dt)
        pair.step_cpu(dt)
        pair.check(exact_all=True, what=f"frame {fr} dt={dt}")
    assert pair.gpu.update_path(0)[0] in ("general", "small")
    assert pair.gpu.count(0) > 3000


def test_more_ring_types_than_one_launch_holds(system):
    """FW_FIFO_PER_LAUNCH (8) ring segments per context; further constant-lifetime types simply take the general path.
    Two-type spawners, one of the two with a lifetime range, so both launches run in every frame."""
    pairs = []
    for k in range(6):
        p0 = _ring_settings(lifetime=S.RandF32.constant(0.2 + 0.02 * k), capacity=4096)
        p1 = S.ParticleSettings(lifetime=S.RandF32(0.1, 0.4) if k % 2 else S.RandF32.constant(0.15), linear_drag=0.1 * k)
        e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(3000.0 + 500 * k),
                                emission_shape=S.EmissionShape.Sphere(0.5 + 0.1 * k))
        e1 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.rate(2000.0))
        e2 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.CountOverDuration(300.0, 0.5, 0.1, 0.9))
        pairs.append(Pair(system, S.ParticleSpawner([p0, p1], [e0, e1, e2]), S.Transform((float(k), 0.0, 0.0)), seed=SEED, uid=k))
    modes = [p.gpu.update_path(t)[0] for p in pairs for t in (0, 1)]
    if system.path == "fifo":
        assert modes.count("fifo") == 8 and modes[:2] == ["fifo", "fifo"], modes
    else:
        assert modes.count("fifo") == 0
    for fr in range(60):
        system.update(DT)
        for p in pairs:
            p.step_cpu(DT)
        if fr % 6 == 5:
            for k, p in enumerate(pairs):
                p.check(what=f"frame {fr} spawner {k}")


def test_instances_destroyed_and_aabb_on_a_wrapped_ring(system):
    """the render hand-off of a ring: records in particle order whatever the head is (a wave that straddles the head
    writes its records one by one), destroyed records of the update that wrapped, AABB of a wrapped live range"""
    import torch

    ps = _ring_settings(lifetime=S.RandF32.constant(0.2), capacity=4096, particles_destroyed=lambda dead: None,
                        emissive_color=S.FireworkGradient.even_samples([(4.0, 2.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)]))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(17000.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 4.0), (0.0, 1.0, 0.0), 0.0),
                            initial_angular_velocity=S.RandVec3(S.RandF32(0.0, 5.0), (0.0, 1.0, 0.0), 0.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=71)
    _expect_path(system, pair)
    cap, guard = 4000, 64
    buf = torch.full(((cap + guard) * 16,), float("nan"), dtype=torch.float32, device="cuda")
    pair.gpu.attach_instances(buf.data_ptr(), cap)
    for fr in range(120):
        system.update(DT)
        pair.step_cpu(DT)
        n = pair.gpu.count(0)
        ref = pair.gpu.instances(0)  # packing pass over the ring
        got = buf[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
        assert np.array_equal(got, ref.view(np.uint32).reshape(n, 16)), f"frame {fr}: attached records differ from packed ones"
        assert bool(torch.isnan(buf[cap * 16:]).all()), "wrote past the attached buffer"
        cp = pair.cpu.particles(0)
        rec = got.view(np.float32).reshape(n, 16)
        assert np.array_equal(rec[:, 3], cp["scale"]) and np.array_equal(rec[:, 8:12], cp["base_color"])
        assert np.array_equal(rec[:, 12:16], cp["emissive_color"])
        assert_particles_match(pair.gpu.destroyed(0), pair.cpu.destroyed(0), False, f"destroyed frame {fr}")
        if fr % 10 == 9:
            pair.check(what=f"frame {fr}")
            any_g, mn_g, mx_g = pair.gpu.aabb()
            parts = pair.gpu.particles(0)
            assert any_g and np.array_equal(mn_g, (parts["position"] - parts["scale"][:, None]).min(axis=0))
            assert np.array_equal(mx_g, (parts["position"] + parts["scale"][:, None]).max(axis=0))
    assert 3000 < pair.gpu.count(0) < 4000


def test_unchanged_planes_are_not_written_but_changed_ones_are(system):
    """the in-place update skips rotation / angular velocity / scale / constant colours where their bits do not change;
    a type whose particles DO spin and scale must still get every plane (trig -> tolerance of tests/parity.py)"""
    still = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), scale_curve=S.FireworkCurve.constant(1.5), capacity=16384)
    spin = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), angular_acceleration=(0.1, 0.0, -0.2), angular_drag=0.3,
                              scale_curve=S.FireworkCurve.uneven_samples([(0.0, 1.0), (0.8, 1.2), (1.0, 0.0)]), capacity=16384)
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(9000.0))
    e1 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.rate(9000.0),
                            initial_angular_velocity=S.RandVec3(S.RandF32(1.0, 9.0), (0.0, 0.6, 0.8), 0.5))
    pair = Pair(system, S.ParticleSpawner([still, spin], [e0, e1]), seed=SEED, uid=5)
    _expect_path(system, pair, 0), _expect_path(system, pair, 1)
    if system.path == "fifo":
        # (default colours are one-key gradients: never rewritten; so is a constant scale curve; a type that cannot turn
        # reads neither its rotation nor its angular-velocity / lifetime plane: position+age and velocity in, the same out;
        # the second type spins and its angular velocity decays: all four state planes in, all of them + the scale out)
        # (round 6: the scale is left to the readers as well -- FW_TYPE_DERIVED for every type, parity.planes_left_to_readers)
        # (round 6, component planes: velocity and angular velocity move as 12 bytes, their constants stay; initial_scale is read -- 4 B --
        # only where the scale plane is stored)
        Dm = 1 - parity.planes_left_to_readers()
        assert pair.gpu.update_path(0)[1] == 28 + 28 + 4 * Dm and pair.gpu.update_path(1)[1] == 56 + 28 + 16 + 12 + 8 * Dm
    for fr in range(90):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 9 == 8:
            pair.check(what=f"frame {fr}")
    assert pair.gpu.count(0) > 4000 and pair.gpu.count(1) > 4000


def test_settings_rebuild_and_despawn_release_the_ring_slots(system):
    """update_settings / despawn give the (at most 8) ring slots of a context back"""
    sp = S.ParticleSpawner([_ring_settings(capacity=4096)], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(5000.0))])
    for rep in range(20):
        hs = [system.spawn(sp, uid=rep * 10 + k) for k in range(4)]
        if system.path == "fifo":
            assert all(h.update_path(0)[0] == "fifo" for h in hs), rep
        for _ in range(5):
            system.update(DT)
        hs[0].update_settings(sp)
        system.update(DT)
        if system.path == "fifo":
            assert hs[0].update_path(0)[0] == "fifo"
        for h in hs:
            system.despawn(h)


def _nested_rings(spark_rate=3000.0, per_spark=20.0, spark_life=0.5, smoke_life=0.4, **smoke_kw):
    sparks = S.ParticleSettings(lifetime=S.RandF32.constant(spark_life), initial_scale=S.RandF32(0.01, 0.03), linear_drag=0.3,
                                base_color=S.FireworkGradient.even_samples([(8.0, 4.0, 1.0, 1.0), (1.0, 0.2, 0.0, 0.0)]))
    kw = dict(lifetime=S.RandF32.constant(smoke_life), initial_scale=S.RandF32(0.05, 0.1), acceleration=(0.0, 0.5, 0.0),
              base_color=S.FireworkGradient.uneven_samples([(0.0, (0.1, 0.1, 0.1, 0.0)), (0.1, (0.1, 0.1, 0.1, 0.15)),
                                                            (1.0, (0.1, 0.1, 0.1, 0.0))]))
    kw.update(smoke_kw)
    smoke = S.ParticleSettings(**kw)
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(spark_rate),
                            initial_velocity=S.RandVec3(S.RandF32(2.0, 6.0), (0.0, 1.0, 0.0), 0.0))
    e1 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.rate(per_spark),
                            emission_mode=S.EmissionMode.Nested(0), inherit_parent_velocity=False)
    return S.ParticleSpawner([sparks, smoke], [e0, e1])


def test_nested_spawner_on_rings_bit_exact(system):
    """sparks -> smoke with one lifetime value each: both types are rings.  The sparks are materialised in their ring by
    fw_k_spawn, the per-parent pass reads them through the head and appends the children into the smoke ring, whose live
    count only the device knows (the host learns each frame's cohort size through the pinned report ring, in time for the
    frame in which that cohort dies); small capacities so that both rings wrap and the smoke ring grows"""
    sp = _nested_rings(particles_destroyed=lambda dead: None)
    pair = Pair(system, sp, S.Transform((0.0, 1.0, 0.0)), seed=SEED, uid=11)
    _expect_path(system, pair, 0), _expect_path(system, pair, 1)
    rng = np.random.default_rng(3)
    for fr in range(150):
        dt = np.float32(DT if fr < 60 else rng.uniform(0.004, 0.03))
        system.update(dt)
        pair.step_cpu(dt)
        if fr % 3 == 2 or fr < 5:
            pair.check(exact_all=True, what=f"frame {fr}")
            assert_particles_match(pair.gpu.destroyed(1), pair.cpu.destroyed(1), True, f"destroyed smoke frame {fr}")
            assert np.array_equal(pair.gpu.last_emitted(0, 1), pair.cpu.last_emitted(0, 1)), f"last_emitted frame {fr}"
    _expect_path(system, pair, 0), _expect_path(system, pair, 1)
    assert pair.gpu.count(0) > 1000 and pair.gpu.count(1) > 10000


@pytest.mark.parametrize("fuse", ["inside the FIFO launch", "separate passes"])
def test_nested_entry_inside_the_fifo_launch(fw_path, monkeypatch, fuse):
    """round 5: a Nested entry whose parents and children both live in FIFO rings runs INSIDE the ring launch (fw_kernels.h:
    FwFifoNest) -- the parents' tiles count, look back and spawn the children with their first update, the child ring's
    bookkeeping workgroup books the total -- instead of fw_k_nest + a second launch.  Sparks in a ring with a caller-given
    capacity that their live count nearly fills (a frame in which the ring wraps into its head tile -- the tiles' ranks are then
    not the list order -- falls back to the separate pass), and an instance buffer attached to the child type for two stretches
    of frames (it keeps the entry out of the launch): fused and separate frames alternate and hand each other the device
    counters; irregular steps, zero steps.  The whole state, the destroyed stream and
    last_emitted_age against the oracle bit for bit, both ways (FW_NEST_FUSE=0: the separate passes throughout)."""
    from bevy_firework_amd.system import ParticleSystem

    if fw_path != "fifo":
        pytest.skip("no FIFO rings on this path")
    monkeypatch.setenv("FW_NEST_FUSE", "1" if fuse == "inside the FIFO launch" else "0")
    sp = _nested_rings(spark_rate=31400.0, per_spark=12.0, particles_destroyed=lambda dead: None)  # (~15 700 sparks in 16 384 slots)
    sp.particle_settings[0].capacity = 16384
    sp.particle_settings[0].particles_destroyed = lambda dead: None
    rng = np.random.default_rng(5)
    with ParticleSystem(device=0, seed=SEED) as system:
        pair = Pair(system, sp, S.Transform((0.0, 1.0, 0.0)), seed=SEED, uid=17)
        assert [pair.gpu.update_path(t)[0] for t in (0, 1)] == ["fifo", "fifo"]
        import torch

        buf = torch.full((200000 * 16,), float("nan"), dtype=torch.float32, device="cuda")
        for fr in range(140):
            dt = np.float32(DT if fr < 70 else (0.0 if fr % 17 == 0 else rng.uniform(0.004, 0.03)))
            # frames 40-59 and 100-109: an instance buffer on the child type keeps the entry out of the FIFO launch (the separate
            # passes run); detached, it goes back in -- the two forms hand each other the device counters both ways
            if fr in (40, 100):
                pair.gpu.attach_instances(buf.data_ptr(), 200000, particle_type=1)
            if fr in (60, 110):
                pair.gpu.attach_instances(0, 0, particle_type=1)
            system.update(dt)
            pair.step_cpu(dt)
            if fr % 4 == 3 or fr < 4:
                pair.check(exact_all=True, what=f"frame {fr}")
                for t in (0, 1):
                    assert_particles_match(pair.gpu.destroyed(t), pair.cpu.destroyed(t), True, f"destroyed type {t} frame {fr}")
                assert np.array_equal(pair.gpu.last_emitted(0, 1), pair.cpu.last_emitted(0, 1)), f"last_emitted frame {fr}"
        fused, separate = system.nest_frames()
        assert [pair.gpu.update_path(t)[0] for t in (0, 1)] == ["fifo", "fifo"]
        assert pair.gpu.count(0) > 12000 and pair.gpu.count(1) > 100000, pair.gpu.counts()
        if fuse == "inside the FIFO launch":
            assert fused >= 100 and separate >= 30, (fused, separate)  # (the frames with the instance buffer attached, and any in which the ring reached into its head tile)
        else:
            assert fused == 0 and separate == 140, (fused, separate)


def test_nested_rings_with_attached_instances_and_idle_frames(system):
    """the child ring's render hand-off (records in particle order from a count the kernel reads on the device) and a step
    longer than both lifetimes: every spark dies, this frame's included (the host knows their number); the smoke type,
    whose newest cohort only this frame's Nested pass can size, leaves the ring mode for it"""
    import torch

    sp = _nested_rings(spark_rate=4000.0)
    pair = Pair(system, sp, S.Transform((0.0, 1.0, 0.0)), seed=SEED, uid=13)
    cap = [8000, 60000]
    bufs = [torch.full((c * 16,), float("nan"), dtype=torch.float32, device="cuda") for c in cap]
    for t in (0, 1):
        pair.gpu.attach_instances(bufs[t].data_ptr(), cap[t], particle_type=t)
    for fr in range(100):
        dt = np.float32(0.5 if fr == 80 else DT)
        system.update(dt)
        pair.step_cpu(dt)
        if fr % 7 == 6 or fr in (80, 81):
            pair.check(exact_all=True, what=f"frame {fr}")
            for t in (0, 1):
                n = pair.gpu.count(t)
                ref = pair.gpu.instances(t)
                got = bufs[t][: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                assert np.array_equal(got, ref.view(np.uint32).reshape(n, 16)), f"frame {fr} type {t}"
        if fr == 79:
            _expect_path(system, pair, 0), _expect_path(system, pair, 1)
    assert pair.gpu.update_path(1)[0] == "general"  # the 0.5 s step was longer than the smoke lives (a type that receives children: never the wave kernel)
    assert pair.gpu.count(1) > 5000


def test_ring_launches_switch_between_the_side_stream_and_the_main_stream(system):
    """a ring next to a compacting segment runs on a stream of its own; an attached instance buffer, a registered live-count
    ring or a reader enqueued on the caller's stream move it back (or join the streams) -- every transition, with work of
    both kinds in flight, must leave the reference's state, the records a reader sees and the per-frame live totals"""
    import torch

    ring = _ring_settings(lifetime=S.RandF32.constant(0.3), capacity=16384)
    other = S.ParticleSettings(lifetime=S.RandF32(0.1, 0.5), linear_drag=0.4)
    pa = Pair(system, S.ParticleSpawner([ring], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(30000.0))]), seed=SEED, uid=1)
    pb = Pair(system, S.ParticleSpawner([other], [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(20000.0),
                                                                      emission_shape=S.EmissionShape.Sphere(1.0))]), seed=SEED, uid=2)
    _expect_path(system, pa)
    assert pb.gpu.update_path(0)[0] in (("range",) if system.path == "range" else ("general", "small"))
    cap = 12000
    buf = torch.full((cap * 16,), float("nan"), dtype=torch.float32, device="cuda")
    live = torch.zeros(8, dtype=torch.int64, device="cuda")
    totals = []
    steps_with_ring = 0  # updates since the live-count ring was registered (slot = that - 1, mod 8)

    def step():
        nonlocal steps_with_ring
        system.update(DT)
        pa.step_cpu(DT), pb.step_cpu(DT)
        if 70 <= fr < 95:
            steps_with_ring += 1

    for fr in range(120):
        if fr == 20:
            pa.gpu.attach_instances(buf.data_ptr(), cap)        # -> ring launches on the main stream
        if fr == 45:
            pa.gpu.attach_instances(0, 0)                        # -> back to the side stream
        if fr == 70:
            system.live_count_ring(live.data_ptr(), 8)           # -> main stream again (the ring is fed by both launches)
        if fr == 95:
            system.live_count_ring(0, 0)
        step()
        if fr % 5 == 4:
            # a reader enqueued on the main stream (the packing pass) ...
            inst = pa.gpu.instances(0)
            step()
            # ... and the frame after it
            pa.check(exact_all=True, what=f"ring frame {fr}"), pb.check(what=f"compacting frame {fr}")
            assert len(inst) > 2000
        if 20 <= fr < 45 and fr % 6 == 0:
            n = pa.gpu.count(0)
            got = buf[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
            assert np.array_equal(got, pa.gpu.instances(0).view(np.uint32).reshape(n, 16)), f"frame {fr}"
        if 70 <= fr < 95:
            torch.cuda.synchronize()
            totals.append((int(live[(steps_with_ring - 1) % 8].item()), pa.cpu.counts()[0] + pb.cpu.counts()[0]))
    assert len(totals) > 20 and all(a == b for a, b in totals), totals[:8]


@pytest.mark.parametrize("case", [test_ring_wraps_many_times_bit_exact, test_irregular_dt_needs_no_forecast, test_growth_while_wrapped,
                                  test_instances_destroyed_and_aabb_on_a_wrapped_ring, test_nested_spawner_on_rings_bit_exact,
                                  test_nested_rings_with_attached_instances_and_idle_frames], ids=lambda f: f.__name__[5:])
@pytest.mark.parametrize("knob", ["FW_NT_MB", "FW_NT_WO_MB"])
def test_non_temporal_form_of_the_kernel(fw_path, monkeypatch, case, knob):
    """a ring launch that streams more than fw_ctx::nt_wo_bytes / nt_bytes (no longer fits the Infinity Cache / several times
    its size) runs a non-temporal instantiation of the kernel (fw_ld4w<NT>: the write-only planes / every plane access): the
    same results, bit for bit -- forced here at every size"""
    from bevy_firework_amd.system import ParticleSystem

    if fw_path == "general":
        pytest.skip("the compacting path has no non-temporal form")
    monkeypatch.setenv(knob, "0")
    with ParticleSystem(device=0, seed=SEED) as nt_system:  # (the knob is read when the context is created)
        nt_system.path = fw_path
        case(nt_system)
