"""The in-place RANGE ring path (fw_k_update_range: particle types whose lifetime is a range, in spawners without Nested
entries) against the CPU oracle: the young part of the list is updated in place, the part that may lose particles this
frame is compacted in place towards it, cohorts join the old part as they age, new particles appear at the tail.  The
ring wraps, grows while wrapped, leaves the mode when its premise breaks, and everything the ABI can observe (order,
state, destroyed records in list order, instance records, AABB, live totals) stays the reference's.  Needs an MI355X."""
import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair, assert_particles_match

pytestmark = pytest.mark.gpu
DT = np.float32(1.0 / 60.0)
SEED = workloads.SEED


# every test of the file with four-round OLD / YOUNG tiles at any size (FW_RANGE_SMALL=0) and with one-round tiles for small
# launches (the product's default: below 384 four-round tiles in all) -- fw_k_update_range<.., TR = 4 / 1>
@pytest.fixture(params=["four-round tiles", "one-round tiles when small"])
def system(monkeypatch, request):
    from bevy_firework_amd.system import ParticleSystem

    monkeypatch.setenv("FW_FIFO", "0")
    monkeypatch.setenv("FW_RANGE", "1")
    monkeypatch.setenv("FW_RANGE_MIN", "0")
    monkeypatch.setenv("FW_SMALL", "0")  # (a type that leaves its ring continues on the compacting kernels: what these tests are about)
    if request.param == "four-round tiles":
        monkeypatch.setenv("FW_RANGE_SMALL", "0")
    else:
        monkeypatch.delenv("FW_RANGE_SMALL", raising=False)
    with ParticleSystem(device=0, seed=SEED) as ps:
        yield ps


def _settings(**kw):
    base = dict(lifetime=S.RandF32(0.15, 0.45), initial_scale=S.RandF32(0.5, 2.0), linear_drag=0.2,
                scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]),
                base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
    base.update(kw)
    return S.ParticleSettings(**base)


def _emission(rate=11000.0, **kw):
    base = dict(emission_pacing=S.EmissionPacing.rate(rate),
                initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (0.0, 1.0, 0.0), 0.0), initial_velocity_radial=S.RandF32(0.0, 1.0))
    base.update(kw)
    return S.EmissionSettings(**base)


def test_small_ring_wraps_many_times_bit_exact(system):
    """capacity 4096, ~3300 live: the young boundary, the old part and the tail cross the end of the buffer again and again;
    deaths in every frame (lifetimes 0.15-0.45 s), no trig anywhere -> the whole state and the destroyed records bit-exact
    every frame"""
    ps = _settings(capacity=4096, particles_destroyed=lambda dead: None)
    pair = Pair(system, S.ParticleSpawner([ps], [_emission()]), S.Transform((1.0, 2.0, 3.0)), seed=SEED, uid=3)
    assert pair.gpu.update_path(0)[0] == "range"
    for fr in range(260):
        system.update(DT)
        pair.step_cpu(DT)
        pair.check(exact_all=True, what=f"frame {fr}")
        assert_particles_match(pair.gpu.destroyed(0), pair.cpu.destroyed(0), True, f"destroyed, frame {fr}")
    assert 2800 < pair.gpu.count(0) < 3800
    assert pair.gpu.update_path(0)[0] == "range"


def test_irregular_dt_zero_steps_and_spinning_particles(system):
    """any dt >= 0 below the shortest lifetime: no forecast exists on this path, nothing to lose; zero steps; a type whose
    particles spin (rotation + angular-velocity planes held across the old tiles' look-back) next to one that cannot"""
    still = _settings(lifetime=S.RandF32(0.2, 0.6))
    spin = _settings(lifetime=S.RandF32(0.3, 0.5), angular_acceleration=(0.1, 0.0, -0.2), angular_drag=0.3,
                     particles_destroyed=lambda dead: None)
    e0 = _emission(9000.0, particle_index=0)
    e1 = _emission(14000.0, particle_index=1, initial_angular_velocity=S.RandVec3(S.RandF32(1.0, 9.0), (0.0, 0.6, 0.8), 0.5),
                   emission_shape=S.EmissionShape.Sphere(0.7))
    pair = Pair(system, S.ParticleSpawner([still, spin], [e0, e1]), seed=SEED, uid=5)
    assert [pair.gpu.update_path(t)[0] for t in (0, 1)] == ["range", "range"]
    rng = np.random.default_rng(5)
    for fr in range(180):
        dt = np.float32(0.0 if fr % 17 == 3 else rng.uniform(0.002, 0.03))
        system.update(dt)
        pair.step_cpu(dt)
        if fr % 4 == 3:
            pair.check(what=f"frame {fr}")
            assert_particles_match(pair.gpu.destroyed(1), pair.cpu.destroyed(1), False, f"destroyed, frame {fr}")
    assert pair.gpu.count(0) > 2000 and pair.gpu.count(1) > 3000
    assert [pair.gpu.update_path(t)[0] for t in (0, 1)] == ["range", "range"]


def test_growth_while_wrapped_and_bursts(system):
    """a derived capacity that a OneShot burst and a rising OnDemand load outgrow: the ring is unwrapped into a larger one
    (old part first) with the young boundary re-based, several times"""
    ps = _settings(lifetime=S.RandF32(0.4, 1.1))
    es = [_emission(6000.0), S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand(), emission_shape=S.EmissionShape.Sphere(1.0))]
    pair = Pair(system, S.ParticleSpawner([ps], es), seed=SEED, uid=9)
    for fr in range(150):
        if fr in (30, 31, 60, 90):
            pair.queue(9000 * (1 + fr // 30))
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 5 == 4 or fr in (30, 31, 32, 60, 61, 90, 91):
            pair.check(what=f"frame {fr}")
    assert pair.gpu.update_path(0)[0] == "range" and pair.gpu.count(0) > 4000


def test_readers_see_list_order(system):
    """instance records (packing pass: the kernel derives particle 0's slot from the device count), AABB, last_emitted_age
    (never touched: f32::MIN) and the per-frame live totals in a registered device ring"""
    import torch

    ps = _settings(capacity=8192)
    pair = Pair(system, S.ParticleSpawner([ps], [_emission(13000.0)]), S.Transform((0.5, 0.0, -1.0)), seed=SEED, uid=21)
    live = torch.zeros(8, dtype=torch.int64, device="cuda")
    for fr in range(100):
        if fr == 40:
            system.live_count_ring(live.data_ptr(), 8)
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 9 == 8:
            parts, inst = pair.gpu.particles(0), pair.gpu.instances(0)
            ref = pair.cpu.particles(0)
            assert_particles_match(parts, ref, True, f"frame {fr}")
            for a, b in (("position", "position"), ("scale", "scale"), ("rotation", "rotation"), ("base_color", "base_color")):
                assert np.array_equal(inst[a], ref[b]), (fr, a)
            any_, mn, mx = pair.gpu.aabb()
            lo = (ref["position"] - ref["scale"][:, None]).min(axis=0)
            hi = (ref["position"] + ref["scale"][:, None]).max(axis=0)
            assert any_ and np.array_equal(mn, lo) and np.array_equal(mx, hi), fr
            assert (pair.gpu.last_emitted(0, 0) == np.float32(-3.40282347e+38)).all()
        if fr >= 40:
            torch.cuda.synchronize()
            assert int(live[(fr - 40) % 8].item()) == pair.cpu.count(0), fr
    system.live_count_ring(0, 0)


def test_leaving_the_mode(system):
    """what ends it: particles written by the caller, an attached instance buffer, a step as long as the shortest lifetime,
    a negative step -- each continues on the compacting path with the reference's state"""
    import torch

    def make(uid):
        return Pair(system, S.ParticleSpawner([_settings(lifetime=S.RandF32(0.3, 0.8), capacity=8192)], [_emission(9000.0)]), seed=SEED, uid=uid)

    pairs = [make(30 + k) for k in range(4)]
    buf = torch.full((8192 * 16,), float("nan"), dtype=torch.float32, device="cuda")
    for fr in range(90):
        if fr == 40:
            parts = pairs[0].cpu.particles(0).copy()
            parts["age"] = np.linspace(0.0, 0.29, len(parts), dtype=np.float32)  # no longer in spawn order
            pairs[0].gpu.write_particles(0, parts), pairs[0].cpu.write_particles(0, parts)
            pairs[1].gpu.attach_instances(buf.data_ptr(), 8192)
        dt = np.float32(0.31 if fr == 55 else DT)  # 0.31 s >= lifetime.min: new particles could die in their first step
        system.update(dt)
        for p in pairs:
            p.step_cpu(dt)
        if fr % 6 == 5 or fr in (40, 41, 55, 56):
            for k, p in enumerate(pairs):
                p.check(exact_all=True, what=f"frame {fr} spawner {k}")
        if fr == 39:
            assert all(p.gpu.update_path(0)[0] == "range" for p in pairs)
        if fr == 50:
            assert [p.gpu.update_path(0)[0] for p in pairs] == ["general", "general", "range", "range"]
            n = pairs[1].gpu.count(0)
            got = buf[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
            assert np.array_equal(got, pairs[1].gpu.instances(0).view(np.uint32).reshape(n, 16))
    assert all(p.gpu.update_path(0)[0] == "general" for p in pairs)
    system.update(np.float32(-0.01))
    for p in pairs:
        p.step_cpu(np.float32(-0.01))
        p.check(exact_all=True, what="negative step")


def test_many_segments_in_one_launch_with_churn(system):
    """forty spawners (two types each, one of them with a constant lifetime: more rings than the eight FIFO records of a
    launch could hold) in one context: one launch updates all of them; spawners are rebuilt and despawned along the way"""
    rng = np.random.default_rng(11)
    pairs = []
    for k in range(40):
        a = _settings(lifetime=S.RandF32(0.2 + 0.01 * k, 0.5 + 0.02 * k), linear_drag=0.1 + 0.01 * k)
        b = _settings(lifetime=S.RandF32.constant(0.3 + 0.01 * k), scale_curve=S.FireworkCurve.constant(1.0))
        es = [_emission(float(rng.uniform(500.0, 6000.0)), particle_index=0), _emission(float(rng.uniform(500.0, 3000.0)), particle_index=1)]
        pairs.append(Pair(system, S.ParticleSpawner([a, b], es), S.Transform((float(k), 0.0, 0.0)), seed=SEED, uid=100 + k))
    assert all(p.gpu.update_path(t)[0] == "range" for p in pairs for t in (0, 1))
    for fr in range(120):
        if fr == 50:
            pairs[3].gpu.update_settings(pairs[3].spawner)  # sync_spawner_data: emission state reset, particles dropped
            pairs[3].cpu.reset()
            system.despawn(pairs[7].gpu)
            pairs.pop(7)
        system.update(DT)
        for p in pairs:
            p.step_cpu(DT)
        if fr % 15 == 14 or fr in (50, 51):
            for k, p in enumerate(pairs):
                p.check(exact_all=True, what=f"frame {fr} spawner {k}")
    assert sum(sum(p.gpu.counts()) for p in pairs) > 50000


def test_steady_state_at_a_million(system):
    """1.1M particles with lifetimes 0.6-1.4 s in one ring (hundreds of old tiles look back in one chain) for 150 frames:
    counts, order and every field against the oracle at three frames"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.6, 1.4), linear_drag=0.1, scale_curve=S.FireworkCurve.even_samples([1.0, 2.0]),
                            base_color=S.FireworkGradient.even_samples([(1.0, 1.0, 1.0, 1.0), (0.0, 0.0, 0.0, 0.0)]))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(1.1e6), initial_velocity=S.RandVec3(S.RandF32(0.0, 10.0), (0.0, 1.0, 0.0), 0.0))
    pair = Pair(system, S.ParticleSpawner([ps], [es]), seed=SEED, uid=44)
    assert pair.gpu.update_path(0)[0] == "range"
    for fr in range(150):
        system.update(DT)
        pair.step_cpu(DT)
        if fr in (40, 100, 149):
            pair.check(exact_all=True, what=f"frame {fr}")
    assert pair.gpu.count(0) > 1_000_000


def test_product_defaults_put_each_type_on_its_path(monkeypatch):
    """no knob set at all (what a product process sees): a large one-lifetime type is a FIFO ring, a large lifetime-range type
    a range ring, small types and the types of a spawner with Nested entries that do not qualify stay on the compacting
    path -- all in one context, three launches per frame, against the oracle"""
    from bevy_firework_amd.system import ParticleSystem

    for k in ("FW_ENABLE_KNOBS", "FW_FIFO", "FW_FIFO_MIN", "FW_RANGE", "FW_RANGE_MIN", "FW_NOSPIN"):
        monkeypatch.delenv(k, raising=False)
    with ParticleSystem(device=0, seed=SEED) as system:
        big_const = S.ParticleSpawner([_settings(lifetime=S.RandF32.constant(0.6))], [_emission(90000.0)])
        big_range = S.ParticleSpawner([_settings(lifetime=S.RandF32(0.4, 0.9))], [_emission(40000.0, emission_shape=S.EmissionShape.Sphere(0.5))])
        small = S.ParticleSpawner([_settings(lifetime=S.RandF32(0.2, 0.5))], [_emission(2000.0)])  # (~700 live: past the wave kernel's bound)
        sparks, tf = workloads.nested(spark_rate=3000.0, smoke_per_spark=8.0)
        pairs = [Pair(system, sp, S.Transform((float(i), 0.5, 0.0)), seed=SEED, uid=500 + i) for i, sp in enumerate((big_const, big_range, small))]
        pairs.append(Pair(system, sparks, tf, seed=SEED, uid=510))
        assert [p.gpu.update_path(0)[0] for p in pairs[:3]] == ["fifo", "range", "general"]
        rng = np.random.default_rng(12)
        for fr in range(150):
            dt = np.float32(DT if fr < 80 else rng.uniform(0.005, 0.025))
            system.update(dt)
            for p in pairs:
                p.step_cpu(dt)
            if fr % 15 == 14:
                for k, p in enumerate(pairs):
                    p.check(what=f"frame {fr} spawner {k}")
        assert [p.gpu.update_path(0)[0] for p in pairs[:3]] == ["fifo", "range", "general"]
        assert pairs[0].gpu.count(0) > 40000 and pairs[1].gpu.count(0) > 20000


def test_windowed_instance_buffer_keeps_the_ring(system):
    """fw_spawner_attach_instances_window: the update of a range ring writes the ParticleInstance records itself, at indices
    counted from the particles the step destroys -- the live records are buffer[first : first + count] -- and the type stays a
    ring (the plain attach moves it to the compacting path).  Records against the packing pass and against the oracle; the
    scale / colour planes are not stored meanwhile (FW_TYPE_DERIVED) and come back when the buffer is detached.  A type that
    cannot turn and one that spins; rings that wrap; a buffer too small for one of them."""
    import torch

    still = _settings(lifetime=S.RandF32(0.2, 0.6), capacity=8192, particles_destroyed=lambda dead: None,
                      emissive_color=S.FireworkGradient.even_samples([(4.0, 2.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)]))
    spin = _settings(lifetime=S.RandF32(0.3, 0.5), capacity=8192, angular_acceleration=(0.1, 0.0, -0.2), angular_drag=0.3)
    e0 = _emission(12000.0, particle_index=0)
    e1 = _emission(14000.0, particle_index=1, initial_angular_velocity=S.RandVec3(S.RandF32(1.0, 9.0), (0.0, 0.6, 0.8), 0.0))
    pair = Pair(system, S.ParticleSpawner([still, spin], [e0, e1]), S.Transform((0.0, 1.0, 0.0)), seed=SEED, uid=61)
    caps = [8192, 3000]  # (the second buffer is smaller than the type's live count: records beyond it are dropped)
    guard = 64
    bufs = [torch.full(((c + guard) * 16,), float("nan"), dtype=torch.float32, device="cuda") for c in caps]
    rng = np.random.default_rng(9)
    for fr in range(200):
        if fr == 30:
            for t in (0, 1):
                pair.gpu.attach_instances_window(bufs[t].data_ptr(), caps[t], particle_type=t)
        if fr == 150:
            for t in (0, 1):
                pair.gpu.attach_instances_window(0, 0, particle_type=t)
            pair.check(what="right after detaching")
        dt = np.float32(DT if fr < 90 else rng.uniform(0.004, 0.03))
        system.update(dt)
        pair.step_cpu(dt)
        if fr % 6 == 5 or fr in (30, 31, 150, 151):
            pair.check(what=f"frame {fr}")
            assert_particles_match(pair.gpu.destroyed(0), pair.cpu.destroyed(0), True, f"destroyed, frame {fr}")
            assert [pair.gpu.update_path(t)[0] for t in (0, 1)] == ["range", "range"], fr
            if 30 <= fr < 150:
                for t in (0, 1):
                    first, n = pair.gpu.instance_window(t)
                    assert n == pair.cpu.count(t), (fr, t, first, n)
                    if t == 0:  # (the type with a particles_destroyed handler: the oracle says how many this step destroyed)
                        assert first == len(pair.cpu.destroyed(0)), (fr, first, len(pair.cpu.destroyed(0)))
                    m = min(n, max(0, caps[t] - first))
                    got = bufs[t][first * 16: (first + m) * 16].cpu().numpy().view(np.uint32).reshape(m, 16)
                    ref = pair.gpu.instances(t)[:m]
                    assert np.array_equal(got, ref.view(np.uint32).reshape(m, 16)), f"frame {fr} type {t}: window != packing pass"
                    rec = got.view(np.float32).view(S.INSTANCE_DTYPE).reshape(m)
                    cp = pair.cpu.particles(t)[:m]
                    for k in ("scale", "base_color", "emissive_color"):
                        assert np.array_equal(rec[k], cp[k]), (fr, t, k)
                    assert bool(torch.isnan(bufs[t][caps[t] * 16:]).all()), "wrote past the attached buffer"
    assert pair.gpu.count(0) > 3000 and pair.gpu.count(1) > 4000


@pytest.mark.parametrize("case", [test_small_ring_wraps_many_times_bit_exact, test_irregular_dt_zero_steps_and_spinning_particles,
                                  test_growth_while_wrapped_and_bursts, test_many_segments_in_one_launch_with_churn,
                                  test_windowed_instance_buffer_keeps_the_ring], ids=lambda f: f.__name__[5:])
@pytest.mark.parametrize("knob", ["FW_NT_MB", "FW_NT_WO_MB"])
def test_non_temporal_form_of_the_kernel(system, monkeypatch, case, knob):
    """a launch that streams more than fw_ctx::nt_wo_bytes / nt_bytes (no longer fits the Infinity Cache / several times its
    size) runs a non-temporal instantiation of the kernel (fw_ld4w<NT>: the write-only planes / every plane access): the same
    results, bit for bit -- forced here at every size"""
    from bevy_firework_amd.system import ParticleSystem

    monkeypatch.setenv(knob, "0")
    with ParticleSystem(device=0, seed=SEED) as nt_system:  # (the knob is read when the context is created)
        case(nt_system)


@pytest.mark.parametrize("case", [test_small_ring_wraps_many_times_bit_exact, test_irregular_dt_zero_steps_and_spinning_particles,
                                  test_growth_while_wrapped_and_bursts, test_many_segments_in_one_launch_with_churn], ids=lambda f: f.__name__[5:])
def test_young_tiles_of_512_slots(system, monkeypatch, case):
    """the YRP = 2 instantiations of fw_k_update_range (young tiles of 512 slots: the product's choice for large rings in round 5, a knob
    since the component planes of round 6 -- fw_ctx::range_young_big): the same results, forced here at every size that runs four-round
    tiles"""
    from bevy_firework_amd.system import ParticleSystem

    monkeypatch.setenv("FW_RANGE_YOUNG_BIG", "1")
    monkeypatch.setenv("FW_RANGE_SMALL", "0")  # (four-round tiles at every size: the 512-slot form belongs to them)
    with ParticleSystem(device=0, seed=SEED) as y_system:
        case(y_system)


def test_an_internal_error_is_sticky_for_its_spawner_until_it_is_rebuilt():
    """Fault injection (the `make ab` build, FW_DEBUG 256: the second OLD tile of every range ring never publishes its count):
    whoever waits for it times out -- an in-place update that went wrong cannot be redone.  The library must not carry on as
    if nothing happened: the NEXT fw_step refuses (FW_EHIP), every call that touches the spawner's particles refuses, the
    neighbour spawner keeps its state, and fw_spawner_update_settings brings the spawner back -- empty, on the compacting
    path -- after which both run and match the oracle again.  In a subprocess: the knobs and the library are per process."""
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, "bevy_firework_amd", "csrc", "libfirework_hip_ab.so")
    if not os.path.exists(ab):
        pytest.skip("libfirework_hip_ab.so not built (make -C bevy_firework_amd/csrc ab)")
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        import oracle
        from bevy_firework_amd import settings as S, workloads, _ffi
        from bevy_firework_amd.system import ParticleSystem, FwError
        from parity import Pair
        DT = np.float32(1 / 60)
        def spawner(lo, hi, rate):
            ps = S.ParticleSettings(lifetime=S.RandF32(lo, hi), linear_drag=0.2)
            es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(rate),
                                    initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0))
            return S.ParticleSpawner([ps], [es])
        with ParticleSystem(device=0, seed=7) as ps:
            victim = ps.spawn(spawner(0.3, 1.5, 40000.0), uid=1)     # old part: several tiles -> parallel OLD tiles, look-back
            bystander = Pair(ps, spawner(5.0, 9.0, 3000.0), seed=7, uid=2)  # nobody old for 5 s: one OLD tile, never waits
            assert victim.update_path(0)[0] == "range" and bystander.gpu.update_path(0)[0] == "range"
            frames, failed = 0, None
            for fr in range(200):
                try:
                    ps.update(DT)
                    frames += 1
                    if fr %% 10 == 9: ps.synchronize()
                except FwError as e:
                    failed = e
                    break
            assert failed is not None and "FW_EHIP" in str(failed), failed
            assert frames > 12                          # the first frames have no third OLD tile: nothing to wait for
            for call in (lambda: ps.step(DT), lambda: victim.counts(), lambda: victim.particles(0), lambda: victim.aabb()):
                try:
                    call(); raise SystemExit("a call on the invalid spawner went through")
                except FwError as e:
                    assert "FW_EHIP" in str(e) and "rebuild" in str(e), e
            n_by = bystander.gpu.count(0)               # the neighbour is readable and was not touched by the refused frames
            assert n_by > 0
            victim.update_settings(spawner(0.3, 1.5, 40000.0))   # fw_spawner_update_settings
            assert victim.update_path(0)[0] in ("general", "small") and victim.counts() == [0]
            for fr in range(120):
                ps.update(DT)
            g = victim.particles(0)   # (its RNG streams go on where they were -- they never replay -- so no fresh oracle matches it)
            assert 20000 < len(g) < 45000
            assert (g["age"] < g["lifetime"]).all() and (np.diff(g["age"]) <= 0).all() and (g["age"] > 0).all()
            assert np.isfinite(g["position"]).all() and (g["lifetime"] >= 0.3).all() and (g["lifetime"] <= 1.5).all()
            assert bystander.gpu.count(0) > n_by
        print("STICKY-OK", frames)
    """) % (root, root)
    env = dict(os.environ, FW_ENABLE_KNOBS="1", FW_LIB_PATH=ab, FW_DEBUG="256", FW_FIFO="0", FW_RANGE="1", FW_RANGE_MIN="0",
               FW_SPIN_LIMIT="4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "STICKY-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("rings", ["fifo", "range"])
def test_a_missing_cohort_report_costs_the_ring_not_the_particles(rings):
    """Fault injection (the `make ab` build, FW_DEBUG 512: in frame 60 the pinned report of every cohort that is due counts as
    missing).  A ring that receives Nested children needs that report -- the size of a cohort only the device knows -- when the
    cohort may start to die.  Until round 5 its absence was found in the middle of the frame's bookkeeping and poisoned the
    spawner (FW_ERR_FORECAST, unrecoverable).  Now fw_step looks for it BEFORE anything of the frame is committed: the ring
    continues on the compacting path, exact counts from the device, particles and order kept -- the run goes on and matches the
    oracle on every frame, before and after.  A one-lifetime child type (FIFO ring) and a lifetime range (range ring)."""
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, "bevy_firework_amd", "csrc", "libfirework_hip_ab.so")
    if not os.path.exists(ab):
        pytest.skip("libfirework_hip_ab.so not built (make -C bevy_firework_amd/csrc ab)")
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        import oracle
        from bevy_firework_amd import settings as S
        from bevy_firework_amd.system import ParticleSystem
        from parity import Pair
        DT = np.float32(1 / 60)
        rings = %r
        child_life = S.RandF32.constant(0.5) if rings == "fifo" else S.RandF32(0.4, 0.7)
        sparks = S.ParticleSettings(lifetime=S.RandF32.constant(0.6) if rings == "fifo" else S.RandF32(0.5, 0.8), linear_drag=0.2)
        smoke = S.ParticleSettings(lifetime=child_life, acceleration=(0.0, 0.5, 0.0))
        e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(4000.0),
                                initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0))
        e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0), inherit_parent_velocity=False,
                                emission_pacing=S.EmissionPacing.CountOverDuration(6.0, 0.0, 0.0, 0.5))
        with ParticleSystem(device=0, seed=11) as ps:
            pair = Pair(ps, S.ParticleSpawner([sparks, smoke], [e0, e1]), seed=11, uid=3)
            by = Pair(ps, S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32.constant(1.0))],
                                            [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(9000.0))]), seed=11, uid=4)
            assert pair.gpu.update_path(1)[0] == rings, pair.gpu.update_path(1)
            for fr in range(140):
                ps.update(DT)
                pair.step_cpu(DT), by.step_cpu(DT)
                if fr %% 10 == 9 or fr in (59, 60, 61):
                    pair.check(exact_all=True, what="frame %%d" %% fr)
                    by.check(exact_all=True, what="bystander, frame %%d" %% fr)
                if fr == 59:
                    assert ps.recovered_rings() == 0 and pair.gpu.update_path(1)[0] == rings
                if fr == 60:
                    assert ps.recovered_rings() >= 1 and pair.gpu.update_path(1)[0] == "general", (ps.recovered_rings(), pair.gpu.update_path(1))
            assert pair.gpu.count(1) > 5000 and pair.gpu.update_path(0)[0] in ("fifo", "range", "general")
        print("RECOVERED-OK")
    """) % (root, root, rings)
    env = dict(os.environ, FW_ENABLE_KNOBS="1", FW_LIB_PATH=ab, FW_DEBUG="512", FW_SMALL="0")
    env.update({"FW_FIFO": "1", "FW_FIFO_MIN": "0", "FW_RANGE": "0"} if rings == "fifo" else {"FW_FIFO": "0", "FW_RANGE": "1", "FW_RANGE_MIN": "0"})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RECOVERED-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


# ---- fw_ctx::range_few: in a context of few segments a small type runs on a range ring too ------------------------------------
def _defaults(monkeypatch, **env):
    for k in ("FW_ENABLE_KNOBS", "FW_FIFO", "FW_FIFO_MIN", "FW_RANGE", "FW_RANGE_MIN", "FW_RANGE_SMALL", "FW_RANGE_FEW", "FW_NOSPIN"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def test_few_small_types_run_on_range_rings_by_default(monkeypatch):
    """no knob set: the reference's own regime -- a handful of spawners of a few hundred particles (examples/sparks.rs) -- on
    range rings, whatever the lifetime (one value or a range); against the oracle through deaths.  (A type that receives Nested
    children gets a FIFO ring at any size -- its derived capacity is large -- and a FIFO ring ends the rule: the third test below)"""
    from bevy_firework_amd.system import ParticleSystem

    _defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        sparks, tf = workloads.example_sparks()
        two = S.ParticleSpawner([_settings(lifetime=S.RandF32.constant(0.4)), _settings(lifetime=S.RandF32(0.1, 0.6))],
                                [_emission(900.0), _emission(1400.0, particle_index=1, emission_shape=S.EmissionShape.Sphere(0.5))])
        pairs = [Pair(system, sparks, tf, seed=SEED, uid=700),
                 Pair(system, S.ParticleSpawner([_settings(lifetime=S.RandF32(0.2, 0.5))], [_emission(1500.0)]), seed=SEED, uid=701),
                 Pair(system, two, seed=SEED, uid=702)]
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["range", "range", "range"]
        assert pairs[2].gpu.update_path(1)[0] == "range"
        rng = np.random.default_rng(3)
        for fr in range(160):
            dt = np.float32(DT if fr < 70 else rng.uniform(0.004, 0.03))
            system.update(dt)
            for p in pairs:
                p.step_cpu(dt)
            if fr % 10 == 9:
                for k, p in enumerate(pairs):
                    p.check(what=f"frame {fr} spawner {k}")
        assert pairs[0].gpu.count(0) > 500 and pairs[2].gpu.count(1) > 300


def test_small_rings_leave_when_the_context_is_no_longer_one_of_few_segments(monkeypatch):
    """the spawner that takes the context past fw_ctx::range_few segments (here 5 instead of 64) sends every small range ring to
    the compacting path, particles and order kept; spawners created after that follow the usual thresholds; despawning does not
    bring rings back by itself, a NEW small type gets one once the context is back at half the limit"""
    from bevy_firework_amd.system import ParticleSystem

    _defaults(monkeypatch, FW_ENABLE_KNOBS="1", FW_RANGE_FEW="5")
    with ParticleSystem(device=0, seed=SEED) as system:
        def small(uid, lo=0.2, hi=0.5):
            return Pair(system, S.ParticleSpawner([_settings(lifetime=S.RandF32(lo, hi))], [_emission(2500.0)]),
                        S.Transform((float(uid % 7), 0.5, 0.0)), seed=SEED, uid=uid)

        pairs = [small(710 + k) for k in range(5)]
        assert all(p.gpu.update_path(0)[0] == "range" for p in pairs)

        def run(n, what):
            for fr in range(n):
                system.update(DT)
                for p in pairs:
                    p.step_cpu(DT)
                if fr % 8 == 7 or fr == n - 1:
                    for k, p in enumerate(pairs):
                        p.check(exact_all=True, what=f"{what} frame {fr} spawner {k}")

        run(40, "five rings")
        pairs.append(small(720, 0.3, 0.3))  # the sixth segment: rings with live particles become compacting segments
        assert all(p.gpu.update_path(0)[0] in ("general", "small") for p in pairs)  # (off their rings: compacting layout, by a workgroup or by a wave)
        for k, p in enumerate(pairs):
            p.check(exact_all=True, what=f"right after the change, spawner {k}")
        run(50, "six compacting segments")
        for p in pairs[:3]:
            system.despawn(p.gpu)
        del pairs[:3]
        pairs.append(small(729))  # four segments in use: still past HALF the limit, where the rule comes back (no ping-pong at the limit)
        assert all(p.gpu.update_path(0)[0] in ("general", "small") for p in pairs) and len(pairs) == 4
        for p in pairs[:2]:
            system.despawn(p.gpu)
        del pairs[:2]
        pairs.append(small(730))  # two were left: the context is one of few segments again
        assert [p.gpu.update_path(0)[0] in ("general", "small") for p in pairs[:2]] == [True, True] and pairs[2].gpu.update_path(0)[0] == "range"
        run(50, "two compacting segments and a ring")
        assert all(p.gpu.count(0) > 400 for p in pairs)


def test_small_rings_leave_when_a_fifo_ring_arrives(monkeypatch):
    """a FIFO launch and a range launch run one after the other, a FIFO launch and the compacting launch side by side: when a large
    one-lifetime type joins a context of small range rings, those continue on the compacting path"""
    from bevy_firework_amd.system import ParticleSystem

    _defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = [Pair(system, S.ParticleSpawner([_settings(lifetime=S.RandF32(0.2, 0.5))], [_emission(3000.0)]), seed=SEED, uid=740),
                 Pair(system, S.ParticleSpawner([_settings(lifetime=S.RandF32.constant(0.4))], [_emission(1000.0)]), seed=SEED, uid=741)]
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["range", "range"]
        for fr in range(45):
            system.update(DT)
            for p in pairs:
                p.step_cpu(DT)
        pairs.append(Pair(system, S.ParticleSpawner([_settings(lifetime=S.RandF32.constant(0.6))], [_emission(90000.0)]), seed=SEED, uid=742))
        assert [p.gpu.update_path(0)[0] in ("general", "small") for p in pairs[:2]] == [True, True] and pairs[2].gpu.update_path(0)[0] == "fifo"
        for fr in range(80):
            system.update(DT)
            for p in pairs:
                p.step_cpu(DT)
            if fr % 10 == 9:
                for k, p in enumerate(pairs):
                    p.check(exact_all=True, what=f"frame {fr} spawner {k}")
        assert pairs[2].gpu.count(0) > 40000 and pairs[0].gpu.count(0) > 500


def test_a_small_nested_spawner_stays_on_range_rings_in_a_context_of_few_segments(monkeypatch):
    """examples/textures.rs (55 bullet cases that leave 110 puffs) with no knob set: the type that receives the children has a
    derived capacity past the FIFO threshold (parents' CAPACITY x children), but next to its small parent type it takes a range
    ring too -- a frame is the Nested pass + ONE update launch -- against the oracle through the deaths of both types; a
    configs[3]-sized spawner (derived capacity in the millions) keeps its FIFO rings"""
    from bevy_firework_amd.system import ParticleSystem

    _defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        tex, tf, _ = workloads.example_textures(with_world=False)
        small, tfs = workloads.nested(spark_rate=300.0, smoke_per_spark=5.0)
        pairs = [Pair(system, tex, tf, seed=SEED, uid=760), Pair(system, small, tfs, seed=SEED, uid=761)]
        assert [(p.gpu.update_path(0)[0], p.gpu.update_path(1)[0]) for p in pairs] == [("range", "range")] * 2
        for fr in range(330):
            system.update(DT)
            for p in pairs:
                p.step_cpu(DT)
            if fr % 15 == 14:
                for k, p in enumerate(pairs):
                    p.check(what=f"frame {fr} spawner {k}")
                    assert np.array_equal(p.gpu.last_emitted(0, 1), p.cpu.last_emitted(0, 1))
        assert pairs[0].gpu.count(1) > 90 and pairs[1].gpu.count(1) > 2000
    with ParticleSystem(device=0, seed=SEED) as system:
        big, tfb = workloads.nested(spark_rate=20000.0, smoke_per_spark=20.0)
        d = system.spawn(big, tfb, uid=762)
        assert (d.update_path(0)[0], d.update_path(1)[0]) == ("fifo", "fifo")


def test_more_one_lifetime_types_than_one_fifo_launch_holds(monkeypatch):
    """product defaults (round 5, fw_ctx::n_spilled): eight large one-lifetime types are eight FIFO rings -- one launch; the NINTH
    takes a range ring and the eight follow it WHERE THEY STAND (fifo_to_range: the ring's head becomes the slot of the first
    young particle, the FIFO cohorts become the young cohorts, nothing is copied): one kind of launch per frame instead of two.
    Live particles in every ring at the moment of the change -- some of them a step from death -- irregular steps afterwards,
    types that spin and types that cannot turn (their lifetimes move into a plane), a type that reports its destroyed particles;
    despawning back to eight types and below brings no FIFO ring back while converted rings exist; when the last of them is
    gone a new large type is a FIFO ring again.  Whole state against the oracle throughout (exact fields bit for bit)."""
    from bevy_firework_amd.system import ParticleSystem

    _defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        def big(uid, life, spin=False, destroyed=False):
            ps = _settings(lifetime=S.RandF32.constant(life), particles_destroyed=(lambda dead: None) if destroyed else None)
            em = _emission(90000.0 + 1000.0 * (uid % 5), initial_angular_velocity=S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.0)) if spin \
                else _emission(90000.0 + 1000.0 * (uid % 5))
            return Pair(system, S.ParticleSpawner([ps], [em]), S.Transform((float(uid % 9), 0.5, 0.0)), seed=SEED, uid=uid)

        pairs = [big(800 + k, 0.5 + 0.01 * k, spin=k % 3 == 0, destroyed=k % 4 == 1) for k in range(8)]
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["fifo"] * 8
        rng = np.random.default_rng(8)

        def run(n, what, irregular=False):
            for fr in range(n):
                dt = np.float32(rng.uniform(0.004, 0.03)) if irregular and fr % 3 else DT
                system.update(dt)
                for p in pairs:
                    p.step_cpu(dt)
                if fr % 10 == 9 or fr == n - 1:
                    for k, p in enumerate(pairs):
                        p.check(what=f"{what} frame {fr} spawner {k}")
                        if p.spawner.particle_settings[0].particles_destroyed is not None:
                            assert_particles_match(p.gpu.destroyed(0), p.cpu.destroyed(0), False, f"{what} frame {fr} spawner {k}: destroyed")

        run(45, "eight FIFO rings")  # (lifetimes 0.50-0.57 s = 30-34 frames: deaths in every ring by now)
        pairs.append(big(820, 0.45))
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["range"] * 9
        for k, p in enumerate(pairs):
            p.check(what=f"right after the change, spawner {k}")
        run(60, "nine range rings", irregular=True)
        for p in pairs[:2]:
            system.despawn(p.gpu)
        del pairs[:2]
        pairs.append(big(821, 0.4, spin=True))  # eight types again: it joins the converted rings (no second kind of launch)
        assert [p.gpu.update_path(0)[0] for p in pairs] == ["range"] * 8
        run(40, "eight range rings")
        for p in pairs:
            system.despawn(p.gpu)
        pairs.clear()
        pairs.append(big(830, 0.5))  # nothing converted is left: FIFO rings are back
        assert pairs[0].gpu.update_path(0)[0] == "fifo"
        run(40, "one FIFO ring")
        assert pairs[0].gpu.count(0) > 40000
