"""The host half of a frame with many small emitters (round 5: solo segments, ops and their per-segment headers written in place
into the parameter slot -- csrc/fw_engine_step.cpp, fw_engine.h: OpList, fw_ctx::n_solo) at PRODUCT DEFAULTS: the bookkeeping the
fast path skips or moves (frame_spawn, the frame-begin pass, the sort + header pass) must not change a single particle.  Every
case against the oracle, bit for bit; the same world with FW_HOST_FAST=0 gives the same digest.  Needs an MI355X."""
import hashlib
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair

pytestmark = pytest.mark.gpu
DT = np.float32(1.0 / 60.0)
SEED = workloads.SEED


def _emitter(rate, life, k, entries=1, types=1, fed=0):
    ps = [S.ParticleSettings(lifetime=S.RandF32(life, life * 1.3), linear_drag=0.1 + 0.01 * (k % 7),
                             base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT)) for _ in range(types)]
    es = [S.EmissionSettings(particle_index=fed, emission_pacing=S.EmissionPacing.rate(rate * (1.0 + 0.5 * e)),
                             initial_velocity=S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.0)) for e in range(entries)]
    return S.ParticleSpawner(ps, es)


def _world(system, n_solo):
    pairs = []
    for k in range(n_solo):  # one type, one entry: solo segments; every third one emits only every few frames
        rate = 25.0 if k % 3 == 2 else 500.0 + 0.5 * k
        pairs.append(Pair(system, _emitter(rate, 0.2, k), S.Transform((float(k % 17), 0.0, float(k // 17))), seed=SEED, uid=1000 + k))
    # two entries feed ONE small type: not solo (its ops of a frame share a header)
    pairs.append(Pair(system, _emitter(300.0, 0.25, 7, entries=2), S.Transform((1.0, 2.0, 3.0)), seed=SEED, uid=5000))
    # emitters that sustain ~300 particles on average -- small types when they are built -- but emit a cycle's worth in its first
    # 0.3 s: 1100 live pass what a WAVE is given (the type continues as a wide one, on a workgroup, from the frame in which the op
    # that does it is made), 2600 also pass the derived capacity (2048: the segment grows in that frame)
    for j, (count, duration) in enumerate(((1100.0, 3.0), (2600.0, 8.0))):
        ps = S.ParticleSettings(lifetime=S.RandF32(0.6, 0.78), base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
        es = S.EmissionSettings(emission_pacing=S.EmissionPacing.CountOverDuration(count, duration, 0.0, 0.3 / duration),
                                initial_velocity=S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.0))
        pairs.append(Pair(system, S.ParticleSpawner([ps], [es]), S.Transform((3.0, float(j), 0.0)), seed=SEED, uid=5100 + j))
    return pairs


def _run(system, pairs, n, what, every=10, dt=DT, loose=()):
    """(`loose`: pairs whose spawn uses trigonometry -- cones, circles: their vector fields within the tolerance of tests/parity.py)"""
    for fr in range(n):
        system.update(dt)
        for p in pairs:
            p.step_cpu(dt)
        if fr % every == every - 1 or fr == n - 1:
            for k, p in enumerate(pairs):
                p.check(exact_all=not any(p is q for q in loose), what=f"{what}, frame {fr}, spawner {k}")


def _scenario(system, digest=None):
    from bevy_firework_amd.system import FwError

    pairs = _world(system, 360)  # (the wave-per-type kernel runs from 352 eligible types on: fw_ctx::small_min)
    assert {p.gpu.update_path(0)[0] for p in pairs} == {"small"}
    _run(system, pairs, 25, "steady", every=5)
    # (past the bound of a wave: a WIDE type, a workgroup of the same kernel -- mode 4; the others keep their wave -- mode 3)
    assert [p.gpu.update_mode(0) for p in pairs[-2:]] == [4, 4] and pairs[-1].gpu.count(0) > 2048
    assert {p.gpu.update_mode(0) for p in pairs[:-2]} == {3}
    # segment slots and spawner slots no longer run in step: a two-type spawner whose entry feeds its SECOND type takes the freed
    # slot 3 and a slot at the end, the spawner built after it the freed slot 40 -- ops arrive out of segment order from here on
    for k in (40, 3):
        system.despawn(pairs[k].gpu)
        del pairs[k]
    pairs.append(Pair(system, _emitter(450.0, 0.2, 1, types=2, fed=1), S.Transform((0.5, 0.0, 0.0)), seed=SEED, uid=6000))
    pairs.append(Pair(system, _emitter(520.0, 0.2, 2), S.Transform((0.0, 0.5, 0.0)), seed=SEED, uid=6001))
    _run(system, pairs, 25, "after the slots were shuffled")
    # a frame that cannot be enqueued (2^30 particles in one op) while ops already sit in the parameter slot: nothing changes
    bad = system.spawn(S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32.constant(0.5))],
                                         [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]), S.Transform(), uid=7000)
    bad.queue_particles((1 << 30) + 5)
    for _ in range(2):
        with pytest.raises(FwError) as e:
            system.update(DT)
        assert e.value.status == -4  # FW_ECAPACITY
    system.despawn(bad)
    for k, p in enumerate(pairs):
        p.check(exact_all=True, what=f"after the failed frames, spawner {k}")
    _run(system, pairs, 12, "after the failed frames")
    if digest is not None:
        for p in pairs:
            for t in range(p.n_types):
                digest.update(p.gpu.particles(t).tobytes())
    return pairs


def _product_defaults(monkeypatch):
    for k in list(os.environ):
        if k.startswith("FW_") and k != "FW_LIB_PATH":
            monkeypatch.delenv(k, raising=False)


def test_many_small_emitters_shuffled_slots_failed_frames_and_a_long_step(monkeypatch):
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        _scenario(system)


def test_the_same_world_without_the_fast_host_path(monkeypatch):
    """FW_HOST_FAST=0 (the frame-begin pass over every segment, ops through a list + the sort + header pass) and the default
    give the same particles, byte for byte"""
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    out = []
    for fast in ("1", "0"):
        monkeypatch.setenv("FW_ENABLE_KNOBS", "1"), monkeypatch.setenv("FW_HOST_FAST", fast)
        h = hashlib.sha256()
        with ParticleSystem(device=0, seed=SEED) as system:
            _scenario(system, h)
        out.append(h.hexdigest())
    assert out[0] == out[1]


def test_hundreds_of_mid_size_emitters_on_a_workgroup_each(monkeypatch):
    """720 emitters of 450-1300 particles (wide types: a workgroup of fw_k_update_small each, in a context of fw_ctx::wide_min = 768
    eligible types or more) next to 80 of ~150 (a wave each) in ONE launch; every fourth wide type reports its destroyed particles; lifetimes cross, so the workgroups compact across their waves"""
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = []
        for k in range(800):
            wide = k % 10 != 9
            life = 0.3 + 0.002 * (k % 50)
            ps = S.ParticleSettings(lifetime=S.RandF32(life, life * 1.25), linear_drag=0.1 + 0.01 * (k % 7), acceleration=(0.0, -2.0, 0.1 * (k % 5)),
                                    scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]), base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT),
                                    particles_destroyed=(lambda recs: None) if k % 4 == 0 else None)
            es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate((1500.0 + 1.5 * k) if wide else 400.0),
                                    initial_velocity=S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.0))
            pairs.append(Pair(system, S.ParticleSpawner([ps], [es]), S.Transform((float(k % 40), 0.0, float(k // 40))), seed=SEED, uid=9000 + k))
        modes = [p.gpu.update_mode(0) for p in pairs]
        assert modes.count(4) == 720 and modes.count(3) == 80, (modes.count(4), modes.count(3), set(modes))
        from parity import assert_particles_match
        for rep in range(3):
            _run(system, pairs, 15, "a workgroup per mid-size type", every=15)
            for k in range(0, 800, 4):  # the destroyed stream of the last frame (core.rs:596-599), in list order
                assert_particles_match(pairs[k].gpu.destroyed(0), pairs[k].cpu.destroyed(0), True, f"destroyed records, spawner {k}, round {rep}")
        assert min(p.gpu.count(0) for p in pairs) > 100 and max(p.gpu.count(0) for p in pairs) > 1000


def test_small_types_write_their_render_records(monkeypatch):
    """instance buffers attached to types the wave- / workgroup-per-type kernel updates (what a renderer does for every spawner): the
    types keep their kernel (its INST instantiation), the records the update leaves are the packed records of the survivors, the
    particles still match the oracle; detaching goes back to the plain instantiation"""
    import torch
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = _world(system, 360)
        _run(system, pairs, 12, "before", every=12)
        modes = [p.gpu.update_mode(0) for p in pairs]
        chosen = [0, 1, 5, 77, 200, 359, 360, 361, 362]  # waves, the type two entries feed, the two that burst (workgroups by now)
        bufs = {k: torch.full((4096 * 16,), float("nan"), dtype=torch.float32, device="cuda") for k in chosen}
        for k in chosen:
            pairs[k].gpu.attach_instances(bufs[k].data_ptr(), 4096, particle_type=0)
        assert [p.gpu.update_mode(0) for p in pairs] == modes  # nobody left the kernel
        for rep in range(3):
            _run(system, pairs, 9, "with instance buffers", every=9)
            for k in chosen:
                n = min(pairs[k].gpu.count(0), 4096)
                want = pairs[k].gpu.instances(0)[:n].view(np.uint32).reshape(n, 16)
                got = bufs[k][: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                assert n > 0 and np.array_equal(got, want), f"instance records of spawner {k}, round {rep}"
        for k in chosen[:4]:
            pairs[k].gpu.attach_instances(0, 0, particle_type=0)
        _run(system, pairs, 10, "some detached", every=10)


def test_colliding_small_types_among_hundreds(monkeypatch):
    """a few colliding emitters (examples/collision.rs: bouncing, and the same with destroy_on_collision) among 360 small ones: they run
    on the COLL instantiation of the small kernel -- a frame stays ONE launch for everybody (one such type used to send the whole
    context through the materialise -> count -> scan -> update passes: 16.7 -> 49 us per frame at 512 emitters,
    profiles/r05/coll_cliff_*.txt) --; particles and destroyed records against the oracle"""
    import copy
    from bevy_firework_amd.system import ParticleSystem
    from parity import assert_particles_match

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = _world(system, 360)
        sp, tf, world = workloads.example_collision()
        system.set_colliders(world)
        colliding = []
        for k in range(4):
            spk = copy.deepcopy(sp)
            cs = spk.particle_settings[0].collision_settings
            spk.particle_settings[0].collision_settings = S.ParticleCollisionSettings(cs.restitution, cs.friction, k >= 2, cs.filter_mask)
            spk.particle_settings[0].particles_destroyed = lambda recs: None
            spk.emission_settings[0].emission_pacing = S.EmissionPacing.rate(100.0 + 20.0 * k)
            p = Pair(system, spk, S.Transform((tf.translation[0] + 0.3 * k, tf.translation[1], tf.translation[2]), tf.rotation), seed=SEED, uid=8000 + k)
            p.cpu.set_colliders(world)
            colliding.append(p)
        pairs += colliding
        assert {p.gpu.update_path(0)[0] for p in pairs} == {"small"}
        for rep in range(6):
            _run(system, pairs, 30, "colliding types on the small kernel", every=30, loose=colliding)
            for p in colliding:
                assert_particles_match(p.gpu.destroyed(0), p.cpu.destroyed(0), False, f"destroyed records of a colliding type, round {rep}")
        # (the bouncing ones have met the slab by now; the destroy-on-collision ones have lost particles to it)
        assert colliding[2].gpu.count(0) < colliding[0].gpu.count(0)
        assert {p.gpu.update_path(0)[0] for p in pairs} == {"small"}


def test_small_types_next_to_a_nested_spawner(monkeypatch):
    """a spawner with a Nested entry (examples/textures.rs) among 360 small emitters: its frames run the separate spawn / nest passes,
    the small types keep spawning their own particles (virtual, from a table of their own) -- everything against the oracle; half way
    the slots are shuffled, so the table is no longer the list that was written in place"""
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        pairs = _world(system, 360)
        sp, tf, world = workloads.example_textures()
        system.set_colliders(world)
        nested = Pair(system, sp, tf, seed=SEED, uid=8100)
        nested.cpu.set_colliders(world)
        pairs.append(nested)
        _run(system, pairs, 40, "next to a Nested spawner", every=20, loose=[nested])
        assert system.nest_frames()[1] > 30  # (the separate passes ran)
        assert {p.gpu.update_mode(0) for p in pairs[:-3]} == {3}
        for k in (40, 3):
            system.despawn(pairs[k].gpu)
            del pairs[k]
        pairs.insert(0, Pair(system, _emitter(450.0, 0.2, 1, types=2, fed=1), S.Transform((0.5, 0.0, 0.0)), seed=SEED, uid=6000))
        pairs.insert(1, Pair(system, _emitter(520.0, 0.2, 2, entries=2), S.Transform((0.0, 0.5, 0.0)), seed=SEED, uid=6001))
        _run(system, pairs, 40, "slots shuffled, next to a Nested spawner", every=20, loose=[nested])
        assert nested.gpu.count(1) > 0


def test_small_types_next_to_a_large_destroy_on_collision_type(monkeypatch):
    """a destroy_on_collision type of ~20 000 particles (neither a ring nor small: the materialise -> count -> scan -> update passes)
    among 360 small emitters: the small launch runs next to those passes on the ring stream, from a table of its own that an event on
    that stream recycles; types leave the wave for a workgroup on the way (the list is re-sent: that frame runs on the main stream);
    an instance buffer is packed from a small type between frames (a reader on the main stream).  Everything against the oracle."""
    import copy
    import torch
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        sp, tf, world = workloads.example_collision()
        big = copy.deepcopy(sp)
        cs = big.particle_settings[0].collision_settings
        big.particle_settings[0].collision_settings = S.ParticleCollisionSettings(cs.restitution, cs.friction, True, cs.filter_mask)
        big.emission_settings[0].emission_pacing = S.EmissionPacing.rate(20000.0)
        system.set_colliders(world)
        bigp = Pair(system, big, tf, seed=SEED, uid=8200)
        bigp.cpu.set_colliders(world)
        pairs = [bigp] + _world(system, 360)
        # (the list of small types is re-sent in the frame a type leaves the wave -- that copy must come AFTER the previous
        # frame's small launch on the ring stream has read the old list.  A small emitter with an OnDemand entry: a queue of 3000
        # in the MIDDLE of the steady side-stream regime takes it past what a wave is given, in one frame)
        ps_q = S.ParticleSettings(lifetime=S.RandF32(0.5, 0.7), base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT))
        es_q = [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(300.0), initial_velocity=S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.0)),
                S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand(), initial_velocity=S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.0))]
        queued = Pair(system, S.ParticleSpawner([ps_q], es_q), S.Transform((5.0, 1.0, 0.0)), seed=SEED, uid=8300)
        pairs.append(queued)
        assert bigp.gpu.update_path(0)[0] == "general"
        buf = torch.empty(4096 * 16, dtype=torch.float32, device="cuda")
        for rep in range(5):
            if rep in (2, 3):
                assert queued.gpu.update_mode(0) == (3 if rep == 2 else 4)  # a wave before the burst, a workgroup after it
                queued.queue(3000 if rep == 2 else 500)
            _run(system, pairs, 16, "next to the collision passes", every=16, loose=[bigp])
            n = pairs[7].gpu.count(0)
            got = pairs[7].gpu.instances(0)  # (packed on the main stream from a type the ring stream's launch updates)
            assert len(got) == n and n > 0
        assert bigp.gpu.count(0) > 5000 and {p.gpu.update_mode(0) for p in pairs[1:]} <= {0, 3, 4} and queued.gpu.count(0) > 100


def test_from_about_a_thousand_types_on_every_small_type_is_walked_by_a_wave(monkeypatch):
    """fw_ctx::wave_all_min (round 6: tools/threshold_sweep.py found a wave per type 27-44 % faster than a workgroup per type at 1024
    emitters x 1000 particles): 930 emitters that sustain ~600 particles each -- wide types, a workgroup each among a few hundred --
    run in the WAVE role of the same kernel; below five sixths of the threshold they are workgroups again.  The list's partition is
    all that moves: particles against the oracle across both transitions."""
    from bevy_firework_amd.system import ParticleSystem

    _product_defaults(monkeypatch)
    with ParticleSystem(device=0, seed=SEED) as system:
        hs = [system.spawn(_emitter(1500.0, 0.4, k), S.Transform((float(k % 31), 0.0, float(k // 31))), uid=20000 + k) for k in range(880)]
        checked = [Pair(system, _emitter(1500.0 + 7.0 * k, 0.4, k), S.Transform((float(k), 1.0, 0.0)), seed=SEED, uid=21000 + k) for k in range(50)]
        assert {p.gpu.update_mode(0) for p in checked} == {3} and hs[0].update_mode(0) == 3  # 930 eligible types: waves
        _run(system, checked, 45, "930 mid-size types, a wave each", every=15)
        assert checked[0].gpu.count(0) > 500
        for h in hs[:250]:
            system.despawn(h)
        assert {p.gpu.update_mode(0) for p in checked} == {4} and hs[300].update_mode(0) == 4  # 680 < 747: workgroups
        _run(system, checked, 30, "680 mid-size types, a workgroup each", every=15)
        more = [system.spawn(_emitter(1500.0, 0.4, k), S.Transform((float(k % 31), 2.0, float(k // 31))), uid=23000 + k) for k in range(230)]
        assert {p.gpu.update_mode(0) for p in checked} == {3} and more[0].update_mode(0) == 3     # 910: waves again
        _run(system, checked, 30, "910 mid-size types, a wave each again", every=15)
