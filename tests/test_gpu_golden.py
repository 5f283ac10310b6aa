"""The HIP path against the committed golden vectors DIRECTLY -- no C oracle in between.

tests/golden/trajectories.npz holds multi-frame trajectories produced by the independent numpy restatement
(tests/golden/np_sim.py: array-oriented, written from the reference lines, numpy's libm).  A misreading of glam /
bevy_math shared by oracle/fw_oracle.c and csrc/fw_math.h (same author) passes test_gpu_parity.py; it does not pass
here.  Also: the reference-held vectors (core.rs:806-834, curve.rs:246-258) re-run on the GPU box, against the oracle
.so loaded there, against the product's host helper and against the device kernels.  Needs an MI355X."""
import json
import os
import sys

import numpy as np
import pytest

import parity
from bevy_firework_amd import settings as S

pytestmark = pytest.mark.gpu
G = parity.GOLDEN_DIR
sys.path.insert(0, G)
import scenarios  # noqa: E402

DT = np.float32(1.0 / 60.0)


@pytest.fixture()
def system():
    from bevy_firework_amd.system import ParticleSystem

    with ParticleSystem(device=0, seed=scenarios.SEED) as ps:
        yield ps


@pytest.mark.parametrize("env", [{}, {"FW_FORECAST": "0"}])
@pytest.mark.parametrize("name", list(scenarios.ALL))
def test_hip_follows_the_numpy_trajectories(monkeypatch, name, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from bevy_firework_amd.system import ParticleSystem

    sc = scenarios.ALL[name]()
    for p in sc["spawner"].particle_settings:
        p.particles_destroyed = lambda dead: None  # report_destroyed on: the destroyed stream is compared too
    n_types = len(sc["spawner"].particle_settings)
    n_em = len(sc["spawner"].emission_settings)
    g = parity.golden()
    with ParticleSystem(device=0, seed=scenarios.SEED) as system:
        h = system.spawn(sc["spawner"], sc["transform"], uid=sc["uid"], modifier=sc["modifier"])
        h.set_parent_velocity(sc["parent_velocity"])
        system.set_colliders(sc.get("colliders", []))

        def check(fr):
            for t in range(n_types):
                want = parity.golden_particles(name, fr, t)
                got = h.particles(t)
                parity.assert_particles_match(got, want, exact_all=bool(sc.get("exact")),
                                              what=f"{name} {env} frame {fr} type {t}")
                lea = g[f"{name}/f{fr}/t{t}/last_emitted_age"]
                for i in range(n_em):
                    assert np.array_equal(h.last_emitted(t, i), lea[:, i]), (name, fr, t, i)
                dead = h.destroyed(t)
                assert np.array_equal(dead["age"], g[f"{name}/f{fr}/t{t}/destroyed_age"]), (name, fr, t)
                if len(dead):
                    for k in ("position", "velocity"):  # collision deaths carry the NEW position / velocity
                        want_k = g[f"{name}/f{fr}/t{t}/destroyed_{k}"]
                        ok, _ = parity.trig_field_errors(dead[k], want_k)
                        assert ok.all() and (not sc.get("exact") or np.array_equal(dead[k], want_k)), (name, fr, t, k)
                    assert np.array_equal(dead["scale"], g[f"{name}/f{fr}/t{t}/destroyed_scale"])

        parity.run_scenario(sc, lambda: system.update, check)
        assert sum(h.counts()) > 500


# ---- the reference-held vectors, on the GPU box -----------------------------------------------------------------
def test_reference_pins_hold_for_the_oracle_loaded_here():
    """the GPU lease proves by itself that the oracle .so it loaded is the pinned one"""
    import test_oracle_golden as tg

    tg.test_reference_emission_unit_test()
    tg.test_reference_gradient_unit_test()
    tg.test_philox_published_kat_and_uniform_stream()
    tg.test_nested_count_kat()


def test_reference_emission_unit_test_through_the_product_library():
    """reference src/core.rs:806-834 through libfirework_hip.so's own arithmetic (fw_compute_emission_count is the
    function fw_step evaluates for Global entries)"""
    from bevy_firework_amd.system import compute_emission_count

    d = json.load(open(os.path.join(G, "emission_kat.json")))
    f = lambda bits: np.array([bits], dtype=np.uint32).view(np.float32)[0]
    total = 0
    for age_b, last_b, n, next_b in d["steps"]:
        got_n, got_next = compute_emission_count(f(age_b), f(last_b), 3.0, 0.0, 1.0, 23.0)
        assert got_n == n and int(np.float32(got_next).view(np.uint32)) == next_b
        total += got_n
    assert total in (22, 23)


def test_reference_gradient_unit_test_on_the_device(system):
    """reference src/curve.rs:246-258 (three even keys; t = 0 and t = 0.5 select keys exactly) evaluated by the
    kernels: t = 0 is the colour a particle is spawned with (core.rs:460), t = 0.5 the colour update_particles stores
    for age / lifetime = 0.5 (core.rs:652-655).  t = 1 cannot be sampled by a live particle (age >= lifetime)."""
    red, green, blue = (1.0, 0.0, 0.0, 1.0), (0.0, 1.0, 0.0, 1.0), (0.0, 0.0, 1.0, 1.0)
    ps = S.ParticleSettings(lifetime=S.RandF32.constant(1.0), base_color=S.FireworkGradient.even_samples([red, green, blue]),
                            emissive_color=S.FireworkGradient.even_samples([blue, red, green]))
    h = system.spawn(S.ParticleSpawner([ps], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())]))
    h.queue_particles(3)
    system.update(np.float32(0.0))  # spawned with gradient(0), then updated at age / lifetime = 0 -> the first key
    p = h.particles(0)
    assert [tuple(c) for c in p["base_color"]] == [red] * 3 and [tuple(c) for c in p["emissive_color"]] == [blue] * 3
    p["age"] = np.float32(0.25)
    h.write_particles(0, p)
    system.update(np.float32(0.25))  # age 0.5 of lifetime 1.0 -> exactly the middle key
    p = h.particles(0)
    assert [tuple(c) for c in p["base_color"]] == [green] * 3 and [tuple(c) for c in p["emissive_color"]] == [red] * 3


def test_nested_count_kat_on_the_device(system):
    """tests/golden/nested_count_kat.json (numpy restatement of core.rs:490-500): one parent per case, counted by the
    Nested kernels frame by frame; children per frame and the parent's last_emitted_age must match bit for bit"""
    d = json.load(open(os.path.join(G, "nested_count_kat.json")))
    for case in d["cases"]:
        parent = S.ParticleSettings(lifetime=S.RandF32.constant(case["lifetime"]))
        child = S.ParticleSettings(lifetime=S.RandF32.constant(1000.0))
        e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.OnDemand())
        e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                                emission_pacing=S.EmissionPacing.CountOverDuration(case["count"], 0.0, case["offset_start"],
                                                                                   case["offset_end"]))
        h = system.spawn(S.ParticleSpawner([parent, child], [e0, e1]))
        h.queue_particles(1)
        children = 0
        for age_b, n, last_b in case["rows"]:
            system.update(DT)
            c = h.counts()
            assert c[1] - children == n, (case["count"], age_b, c, children, n)
            children = c[1]
            if c[0]:
                assert int(h.last_emitted(0, 1)[0].view(np.uint32)) == last_b
        assert children == case["total"]
        system.despawn(h)


def test_global_burst_into_a_nested_fed_type_stays_in_bounds(system):
    """a Global (OnDemand) entry and a Nested entry feed the same particle type with a caller-given capacity: the
    burst is clamped to the capacity on the device and reported, nothing is read or written past the buffers
    (neighbouring spawners stay intact), and the state stays usable"""
    from bevy_firework_amd.system import FwError

    sparks = S.ParticleSettings(lifetime=S.RandF32.constant(1.0))
    mixed = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), capacity=4096)
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(600.0))
    e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                            emission_pacing=S.EmissionPacing.CountOverDuration(4.0, 0.0, 0.0, 1.0))
    e2 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.OnDemand())
    # a neighbour allocated right after: an out-of-bounds write would land in its planes
    h = system.spawn(S.ParticleSpawner([sparks, mixed], [e0, e1, e2]), uid=1)
    nb = parity.Pair(system, S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32.constant(2.0))],
                                               [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(20000.0))]),
                     seed=scenarios.SEED, uid=2)
    for fr in range(40):
        if fr in (5, 6, 20):
            h.queue_particles(30000)  # far beyond the 4096 slots
        system.update(DT)
        nb.step_cpu(DT)
        # (round 6: the burst used to ask for 118 new-particle tiles -- 30 000 / 256 -- when the per-tile arrays of the context,
        # sized from the capacities, held 36: forecast entries and status words of the tiles beyond were written out of bounds, and
        # with other processes on the GPU an update kernel's `check 2` fired.  What exceeds the capacity needs no tile.)
        table, scratch = system.tile_scratch()
        assert table <= scratch, (fr, table, scratch)
    with pytest.raises(FwError) as e:
        h.counts()
    assert e.value.status == -4  # FW_ECAPACITY
    c = h.counts()
    assert c[1] <= 4096 and c[0] > 300
    p = h.particles(1)
    assert np.isfinite(p["position"]).all() and (p["age"] < p["lifetime"]).all()
    nb.check(exact_all=True, what="neighbour of the overflowing type")
    for _ in range(40):  # everything of the burst expires; the type keeps working
        system.update(DT)
        nb.step_cpu(DT)
    assert h.counts()[1] < 4096
    nb.check(exact_all=True, what="neighbour, later")


def test_nested_fed_type_with_derived_capacity_grows(system):
    """the reference pushes children onto a Vec (core.rs:523); a Nested-fed type whose capacity was derived follows
    the live count seen in the snapshot rows and doubles before it can overflow: OnDemand parents far beyond what
    the capacity was derived from, no FW_ECAPACITY, counts identical to the oracle"""
    parent = S.ParticleSettings(lifetime=S.RandF32.constant(1.5))
    child = S.ParticleSettings(lifetime=S.RandF32.constant(0.6))
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.OnDemand())
    e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                            emission_pacing=S.EmissionPacing.CountOverDuration(30.0, 0.0, 0.0, 1.0))
    pair = parity.Pair(system, S.ParticleSpawner([parent, child], [e0, e1]), seed=scenarios.SEED, uid=8)
    for fr in range(120):
        if fr % 10 == 0 and fr < 60:
            pair.queue(600 * (1 + fr // 10))  # 600, 1200, ... parents per burst: ramps up gradually
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 20 == 19:
            pair.check(exact_all=True, what=f"frame {fr}")
    assert max(pair.gpu.counts()) > 20000


def _exact_rotation(w32, dt32, rot0, steps):
    """float64 ground truth of `steps` applications of from_scaled_axis(w dt) * rotation with a constant w"""
    v = w32.astype(np.float64) * np.float64(dt32)
    ln = np.linalg.norm(v, axis=1)
    safe = np.where(ln == 0, 1.0, ln)
    dq = np.concatenate([v / safe[:, None] * np.sin(ln / 2)[:, None], np.cos(ln / 2)[:, None]], axis=1)
    q = rot0.astype(np.float64).copy()
    for _ in range(steps):
        x0, y0, z0, w0 = dq.T
        x1, y1, z1, w1 = q.T
        q = np.stack([w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1, w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
                      w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1, w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1], axis=1)
    return q


def test_rotation_drift_over_2000_frames_is_no_worse_than_the_oracles(system):
    """rotation = from_scaled_axis(w dt) * rotation is never renormalised (core.rs:645-647).  With a CONSTANT angular
    velocity (drag 0) the same rounded step quaternion is applied every frame, so its half-ulp norm / angle error
    compounds linearly and ANY two implementations that differ in one rounding drift apart (numpy's sin/cos, 99.2 %
    bit-identical to glibc's, is 1e-4 away from the oracle after 2000 frames; the oracle itself is 6e-5 away from the
    float64 truth).  What can be asked of the kernels (small-angle polynomials instead of sinf/cosf + divisions) is
    that they are as close to the TRUTH as the oracle is: checked here against a float64 recurrence."""
    ps = S.ParticleSettings(lifetime=S.RandF32.constant(1000.0), angular_drag=0.0, angular_acceleration=(0.0, 0.0, 0.0),
                            linear_drag=1.0, acceleration=(0.0, 0.0, 0.0))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.OneShot(3000),
                            initial_angular_velocity=S.RandVec3(S.RandF32(0.5, 12.0), (0.3, 0.9, -0.3), 1.2))
    pair = parity.Pair(system, S.ParticleSpawner([ps], [es]), seed=scenarios.SEED, uid=77)
    w = rot0 = None
    for fr in range(2000):
        system.update(DT)
        pair.step_cpu(DT)
        if fr == 0:
            w = pair.cpu.particles(0)["angular_velocity"].copy()  # constant from here on (no drag, no acceleration)
            assert np.array_equal(pair.gpu.particles(0)["angular_velocity"], w) or parity.trig_field_errors(
                pair.gpu.particles(0)["angular_velocity"], w)[0].all()
        if fr + 1 in (250, 1000, 2000):
            truth = _exact_rotation(w, DT, np.tile(np.array([0, 0, 0, 1.0]), (len(w), 1)), fr + 1)
            got, want = pair.gpu.particles(0)["rotation"], pair.cpu.particles(0)["rotation"]
            e_hip, e_cpu = np.abs(got - truth).max(), np.abs(want - truth).max()
            print(f"frame {fr + 1}: |HIP - truth| {e_hip:.3g}  |oracle - truth| {e_cpu:.3g}  |HIP - oracle| {np.abs(got - want).max():.3g}")
            assert e_hip <= 1.5 * e_cpu + 1e-6, (fr, e_hip, e_cpu)
            assert e_hip < 4e-8 * (fr + 1) + 1e-6  # half an ulp per step, linear: 8e-5 at 2000 frames


def test_rotation_with_default_drag_stays_inside_the_tolerance_for_2000_frames(system):
    """the default angular_drag 0.2 (core.rs:203): the angular velocity changes every frame, the per-step roundings do
    not repeat, and HIP and oracle stay inside the usual 1e-5 for 2000 frames (33 s)"""
    ps = S.ParticleSettings(lifetime=S.RandF32.constant(1000.0), linear_drag=1.0, acceleration=(0.0, 0.0, 0.0))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.OneShot(3000),
                            initial_rotation=(0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))),
                            initial_angular_velocity=S.RandVec3(S.RandF32(0.5, 12.0), (0.3, 0.9, -0.3), 1.2))
    pair = parity.Pair(system, S.ParticleSpawner([ps], [es]), seed=scenarios.SEED, uid=78)
    worst = 0.0
    for fr in range(2000):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 250 == 249:
            ok, wst = parity.trig_field_errors(pair.gpu.particles(0)["rotation"], pair.cpu.particles(0)["rotation"])
            worst = max(worst, wst)
            assert ok.all(), f"frame {fr}: rotation at {wst:.2f}x the allowance"
    print(f"rotation, default drag, 2000 frames: worst {worst:.3f} of the allowance (rtol {parity.RTOL})")
    pair.check(what="after 2000 frames")


def test_sharded_system_feeds_the_exchange_from_the_device_ring():
    """bevy_firework_amd/sharding.py on the real backend: per-frame totals come from the ring the update kernel
    writes (no live_count() synchronisation in the frame), buckets incl. a partial one and a wrapped window"""
    import torch

    from bevy_firework_amd import sharding, workloads
    from bevy_firework_amd.system import ParticleSystem

    ems = workloads.many_emitters(6, live_per_emitter=3000)
    stream = torch.cuda.Stream()
    sh = sharding.ShardedParticleSystem(lambda: ParticleSystem(device=0, seed=workloads.SEED, stream=stream.cuda_stream),
                                        ems, rank=0, world=1, reduce_every=8, torch_stream=stream, exchange=True)
    assert sh._device_ring
    want = []
    import oracle

    cpu = [oracle.OracleSpawner(s, seed=workloads.SEED, uid=e, transform=tf) for e, (s, tf) in enumerate(ems)]
    for fr in range(45):
        sh.update(DT)
        for o in cpu:
            o.step(DT)
        want.append(sum(o.count(0) for o in cpu))
        if fr == 19:
            sh.flush()  # moves the bucket boundary: later windows wrap around the ring
    sh.flush()
    assert sh.global_live_history == want
    assert sh.global_live_count() == want[-1]
    sh.system.close()


def test_sharded_system_over_a_one_rank_rccl_group():
    """the same class with the nccl (= RCCL) backend: device buckets all-reduced on the GPU; run in a subprocess with
    a timeout so that a communicator problem cannot hang the suite"""
    import subprocess

    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from bevy_firework_amd import sharding, workloads
from bevy_firework_amd.system import ParticleSystem
stream = torch.cuda.Stream()
sh = sharding.ShardedParticleSystem(lambda: ParticleSystem(device=0, seed=workloads.SEED, stream=stream.cuda_stream),
                                    workloads.many_emitters(4, 2000), 0, 1, process_group=dist.group.WORLD,
                                    reduce_every=4, torch_stream=stream)
dt = np.float32(1 / 60)
local = []
for fr in range(24):
    sh.update(dt)
    local.append(sh.local_live_count())
assert all(b.is_cuda for b in sh._buckets), "buckets must stay on the device with the nccl backend"
assert sh.global_live_history == local, (sh.global_live_history, local)
sh.system.close(); dist.destroy_process_group(); print("RCCL_OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=240)
    assert "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_collisions_against_the_oracle_with_nested_and_attached_instances(system):
    """the collision path (count / scan / update-with-collisions launches) together with everything else a frame can
    hold: a Nested entry (children of bouncing parents), a non-colliding neighbour in the same context, an attached
    instance buffer on the colliding type, colliders replaced mid-run, destroyed streams of both death causes"""
    import torch

    bounce = S.ParticleSettings(lifetime=S.RandF32(0.8, 1.6), linear_drag=0.1,
                                collision_settings=S.ParticleCollisionSettings(0.7, 0.2))
    bounce.particles_destroyed = lambda d: None
    trail = S.ParticleSettings(lifetime=S.RandF32(0.2, 0.5), acceleration=(0.0, 0.0, 0.0),
                               collision_settings=S.ParticleCollisionSettings(0.0, 0.0, destroy_on_collision=True))
    trail.particles_destroyed = lambda d: None
    # no libm call anywhere (Point shape, zero spread; directions vary through three entries and a parent velocity that
    # changes every frame): a bounce amplifies differences, so this test is bit-exact or nothing
    dirs = [(0.2, -1.0, 0.0), (-0.6, -0.8, 0.3), (0.5, -0.2, -0.7)]
    e0 = [S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(1000.0),
                             initial_velocity=S.RandVec3(S.RandF32(1.0, 9.0), d, 0.0)) for d in dirs]
    e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                            emission_pacing=S.EmissionPacing.CountOverDuration(12.0, 0.0, 0.0, 1.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 2.0), (0.0, -1.0, 0.0), 0.0))
    pair = parity.Pair(system, S.ParticleSpawner([bounce, trail], e0 + [e1]), S.Transform((0.0, 2.5, 0.0)),
                       seed=scenarios.SEED, uid=31)
    plain = parity.Pair(system, S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32(0.3, 0.9))],
                                                  [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(20000.0))]),
                        seed=scenarios.SEED, uid=32)
    world_a = [S.Collider.Plane((0, 0, 0), (0, 1, 0)), S.Collider.Sphere((0.8, 0.6, 0.0), 0.5)]
    world_b = world_a + [S.Collider.Box((-0.9, 0.5, 0.2), (0.5, 0.5, 0.5), (0.0, 0.0, float(np.sin(0.2)), float(np.cos(0.2))))]
    system.set_colliders(world_a)
    pair.cpu.set_colliders(world_a)
    cap = 60000
    buf = torch.full((cap * 16,), float("nan"), dtype=torch.float32, device="cuda")
    pair.gpu.attach_instances(buf.data_ptr(), cap, particle_type=0)
    for fr in range(120):
        if fr == 60:
            system.set_colliders(world_b)
            pair.cpu.set_colliders(world_b)
        pv = (float(np.float32(((fr * 37) % 23) / 11.0 - 1.0)), 0.0, float(np.float32(((fr * 17) % 13) / 6.0 - 1.0)))
        pair.gpu.set_parent_velocity(pv)
        pair.cpu.set_parent_velocity(pv)
        system.update(DT)
        pair.step_cpu(DT)
        plain.step_cpu(DT)
        if fr % 20 == 19:
            pair.check(exact_all=True, what=f"colliding spawner, frame {fr}")
            plain.check(exact_all=True, what=f"neighbour, frame {fr}")
            for t in (0, 1):
                gd, cd = pair.gpu.destroyed(t), pair.cpu.destroyed(t)
                assert len(gd) == len(cd) and np.array_equal(gd["age"], cd["age"]) and np.array_equal(gd["scale"], cd["scale"])
                assert np.array_equal(gd["position"], cd["position"]) and np.array_equal(gd["velocity"], cd["velocity"])
            n = pair.gpu.count(0)
            rec = buf[: n * 16].cpu().numpy().view(S.INSTANCE_DTYPE).reshape(n)
            assert np.array_equal(rec["position"], pair.gpu.particles(0)["position"])
    c = pair.gpu.counts()
    assert c[0] > 2000 and c[1] > 1000, c
    # the ground really acts (without it nearly everything would be far below y = 0 by now); a few particles do end up
    # below it, as in the reference: a particle that starts a frame inside a solid is pushed along its velocity
    # (core.rs:766-776), and `delta - hit.distance` mixes a time with a length (core.rs:786)
    assert (pair.gpu.particles(0)["position"][:, 1] > -1e-3).mean() > 0.8


def test_particles_that_emit_onto_their_own_type(system):
    """Nested { target_particle_type } == particle_index: the parent bound `0..particles[t].len()` is fixed when the
    entry starts (core.rs:488), the children are appended behind it and become parents only in the NEXT frame.  On the
    device such an op commits its totals through the ticket path of fw_k_nest (a late workgroup must not see this
    frame's children as parents); several parent tiles, counts and state against the oracle, last_emitted_age too."""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 0.6), linear_drag=0.4)
    seedling = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(2500.0),
                                  initial_velocity=S.RandVec3(S.RandF32(0.5, 3.0), (0.0, 1.0, 0.0), 0.0))
    budding = S.EmissionSettings(particle_index=0, emission_mode=S.EmissionMode.Nested(0),
                                 emission_pacing=S.EmissionPacing.CountOverDuration(1.6, 0.0, 0.3, 0.95),
                                 initial_velocity=S.RandVec3(S.RandF32(0.2, 1.0), (1.0, 0.0, 0.0), 0.0),
                                 inherit_parent_velocity=True)
    pair = parity.Pair(system, S.ParticleSpawner([ps], [seedling, budding]), seed=scenarios.SEED, uid=41)
    for fr in range(150):
        system.update(DT)
        pair.step_cpu(DT)
        if fr % 15 == 14:
            pair.check(exact_all=True, what=f"self-nested frame {fr}")
            assert np.array_equal(pair.gpu.last_emitted(0, 1), pair.cpu.last_emitted(0, 1))
    assert pair.gpu.count(0) > 3000  # well beyond what the Global entry alone sustains (2500/s x 0.45 s)


def test_aabb_fused_into_the_update_matches_the_two_pass_query(system):
    """fw_ctx_track_aabbs: the update kernels leave the box of position -/+ scale of every tile's survivors and
    fw_spawner_aabb folds those tile boxes; the two-pass reduction over the stored planes (and numpy over the read-back
    particles) must give the same box bit for bit -- forecast frames, a changed dt (look-back kernel), a Nested spawner
    (materialised tiles), thousands of particles in several tiles, a tiny spawner (new particles riding in the live tile),
    an empty one, and the fallback after the state was touched outside fw_step"""
    from bevy_firework_amd import workloads

    big, tf = workloads.stress_test(rate=40000.0)
    nest, tfn = workloads.nested(spark_rate=2000.0, smoke_per_spark=10.0)
    tiny = S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32(0.2, 0.5))],
                             [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(300.0),
                                                 emission_shape=S.EmissionShape.Sphere(0.4))])
    idle = S.ParticleSpawner([S.ParticleSettings()], [S.EmissionSettings(emission_pacing=S.EmissionPacing.OnDemand())])
    hs = [system.spawn(big, tf, uid=1), system.spawn(nest, tfn, uid=2), system.spawn(tiny, S.Transform((5.0, 0.0, 1.0)), uid=3),
          system.spawn(idle, uid=4)]
    system.track_aabbs(True)
    dts = [1 / 60] * 20 + [1 / 45, 1 / 60, 1 / 75] + [1 / 60] * 15
    for i, dt in enumerate(dts):
        system.update(np.float32(dt))
        if i % 4 == 3 or i in (20, 21, 22):
            fused = [h.aabb() for h in hs]
            system.track_aabbs(False)  # drops the tile boxes: the same query now re-reads the particles
            twopass = [h.aabb() for h in hs]
            system.track_aabbs(True)
            for h, (a1, mn1, mx1), (a2, mn2, mx2) in zip(hs, fused, twopass):
                assert a1 == a2 and np.array_equal(mn1, mn2) and np.array_equal(mx1, mx2), (i, h.handle, mn1, mn2, mx1, mx2)
                if a1:
                    p = np.concatenate([h.particles(t) for t in range(len(h.settings.particle_settings))])
                    assert np.array_equal(mn1, (p["position"] - p["scale"][:, None]).min(axis=0))
                    assert np.array_equal(mx1, (p["position"] + p["scale"][:, None]).max(axis=0))
            # (tracking was re-enabled after a frame without boxes: the next query must not use stale ones)
            assert hs[0].aabb()[0]
    assert not hs[3].aabb()[0] and hs[1].counts()[1] > 1000
    # state rewritten by the caller: the boxes of the last update no longer describe it
    p = hs[0].particles(0)[:100].copy()
    p["position"] += np.float32(100.0)
    hs[0].write_particles(0, p)
    a, mn, mx = hs[0].aabb()
    assert a and mn[0] > 90.0
