"""Round 6: the compacting path under a dt that never repeats -- what an unmodified Bevy `Update` schedule delivers (plugin.rs:26-31) --
on the THRESHOLD FORECAST (csrc/fw_kernels.h: FwUpdateArgs::fc_theta, fw_k_fc_resolve): the previous update lists, per tile, the
survivors a step of theta = 1.25 dt would destroy; the next frame, whatever its dt below theta, rewrites the forecast entries from those
lists and runs the streaming schedule instead of the decoupled look-back.  Counts, order and every field against the oracle on EVERY
frame; the product only engages the scheme from 2048 tiles on -- here from 0 (FW_TF_MIN_TILES).  Needs an MI355X."""
import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from parity import Pair

pytestmark = pytest.mark.gpu
SEED = workloads.SEED


def _system(monkeypatch, **env):
    from bevy_firework_amd.system import ParticleSystem

    base = {"FW_ENABLE_KNOBS": "1", "FW_FIFO": "0", "FW_RANGE": "0", "FW_SMALL": "0", "FW_TF_MIN_TILES": "0"}
    base.update(env)
    for k, v in base.items():
        monkeypatch.setenv(k, v)
    return ParticleSystem(device=0, seed=SEED)


def _emitter(rate, lo, hi, **kw):
    ps = S.ParticleSettings(lifetime=S.RandF32(lo, hi), linear_drag=0.15, particles_destroyed=lambda dead: None,
                            base_color=S.FireworkGradient.uneven_samples(workloads.STRESS_GRADIENT), **kw)
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(rate), initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0))
    return S.ParticleSpawner([ps], [es])


def _dts(n, seed=3, spikes=()):
    rng = np.random.default_rng(seed)
    d = (1.0 / 60.0) * (1.0 + 0.12 * np.sin(0.7 * np.arange(n)) + 0.03 * rng.standard_normal(n))
    d = d.astype(np.float32)
    for k, v in spikes:
        d[k] = np.float32(v)
    return d


@pytest.mark.parametrize("stream", ["1", "0"])
def test_a_jittering_dt_runs_the_streaming_schedule_and_matches_the_oracle(monkeypatch, stream):
    """five emitters -- lifetimes of 0.3-1.1 s (a dozen risky survivors per tile), of 3-6 FRAMES (most of a tile is risky: the lists
    overflow and the resolve pass recounts from the particles), one of several tiles, one that spins -- under dt = 1/60 (1 + 12 % sine +
    noise), with spikes beyond theta (the look-back takes that frame), a repeated dt (the classic forecast) and a zero dt"""
    with _system(monkeypatch, FW_STREAM=stream) as system:
        pairs = [Pair(system, _emitter(9000.0, 0.3, 1.1), seed=SEED, uid=1),
                 Pair(system, _emitter(30000.0, 0.05, 0.1), S.Transform((2.0, 0.0, 0.0)), seed=SEED, uid=2),
                 Pair(system, _emitter(4000.0, 0.5, 0.5001), S.Transform((4.0, 0.0, 0.0)), seed=SEED, uid=3),
                 Pair(system, _emitter(700.0, 0.2, 0.9), S.Transform((6.0, 0.0, 0.0)), seed=SEED, uid=4)]
        spin = _emitter(5000.0, 0.4, 0.8, angular_acceleration=(0.0, 0.3, 0.0))
        spin.emission_settings[0].initial_angular_velocity = S.RandVec3(S.RandF32(1.0, 4.0), (0.0, 1.0, 0.0), 0.3)
        pairs.append(Pair(system, spin, S.Transform((8.0, 0.0, 0.0)), seed=SEED, uid=5))
        assert {p.gpu.update_path(0)[0] for p in pairs} == {"general"}
        dts = _dts(170, spikes=((40, 0.05), (41, 1.0 / 60.0), (42, 1.0 / 60.0), (43, 1.0 / 60.0), (90, 0.0), (120, 0.03)))
        for fr, dt in enumerate(dts):
            system.update(dt)
            for p in pairs:
                p.step_cpu(dt)
            if fr % 6 == 5 or fr in (40, 41, 42, 43, 44, 90, 91, 120, 121):
                for k, p in enumerate(pairs):
                    p.check(exact_all=(k != 4), what=f"frame {fr} (dt {dt}), emitter {k}")
                    from parity import assert_particles_match
                    assert_particles_match(p.gpu.destroyed(0), p.cpu.destroyed(0), k != 4, f"destroyed records, frame {fr}, emitter {k}")
        assert pairs[0].gpu.count(0) > 5000 and pairs[1].gpu.count(0) > 1500
        if stream == "1":
            assert system.tf_frames() > 120, system.tf_frames()  # (all but the spikes, the repeated dt and the frames right after them)
        else:
            assert system.tf_frames() == 0  # (FW_STREAM=0: no streaming schedule to hand the entries to)


def test_threshold_forecast_with_a_nested_spawner_and_attached_instances(monkeypatch):
    """frames with a Nested entry (materialised new particles: the streaming kernel takes them as loaded tiles when they all survive)
    and a type whose update also writes instance records, under a jittering dt"""
    import torch

    with _system(monkeypatch) as system:
        sparks = S.ParticleSettings(lifetime=S.RandF32(0.5, 0.9), linear_drag=0.2)
        smoke = S.ParticleSettings(lifetime=S.RandF32(0.4, 0.8), acceleration=(0.0, 0.5, 0.0))
        e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(3000.0),
                                initial_velocity=S.RandVec3(S.RandF32(1.0, 5.0), (0.0, 1.0, 0.0), 0.0))
        e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0), inherit_parent_velocity=False,
                                emission_pacing=S.EmissionPacing.CountOverDuration(5.0, 0.0, 0.0, 0.6))
        nested = Pair(system, S.ParticleSpawner([sparks, smoke], [e0, e1]), seed=SEED, uid=11)
        inst = Pair(system, _emitter(12000.0, 0.3, 0.9), S.Transform((3.0, 0.0, 0.0)), seed=SEED, uid=12)
        cap = 32768
        buf = torch.full((cap * 16,), float("nan"), dtype=torch.float32, device="cuda")
        inst.gpu.attach_instances(buf.data_ptr(), cap, particle_type=0)
        dts = _dts(120, seed=9)
        for fr, dt in enumerate(dts):
            system.update(dt)
            nested.step_cpu(dt), inst.step_cpu(dt)
            if fr % 8 == 7:
                nested.check(exact_all=True, what=f"nested, frame {fr}")
                inst.check(exact_all=True, what=f"attached, frame {fr}")
                n = inst.gpu.count(0)
                got = buf[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                assert np.array_equal(got, inst.gpu.instances(0).view(np.uint32).reshape(n, 16)), fr
        assert nested.gpu.count(1) > 3000 and system.tf_frames() > 60, system.tf_frames()


def test_product_defaults_a_million_particles_with_a_plain_attach_under_a_jittering_dt(monkeypatch):
    """NO knob set: a lifetime-range type of ~1M particles runs on its range ring under a dt that never repeats; at frame 45 its renderer
    uses the plain fw_spawner_attach_instances and the type leaves the ring for the compacting path (records counted from 0: DESIGN.md
    4.0b) -- ~980 tiles, beyond fw_ctx::tf_min_tiles: the product itself picks fw_k_fc_resolve + the streaming schedule.  Counts, order,
    every field and the records against the oracle, before, at and after the change of paths."""
    import torch
    from bevy_firework_amd.system import ParticleSystem

    for k in list(__import__("os").environ):
        if k.startswith("FW_") and k != "FW_LIB_PATH":
            monkeypatch.delenv(k)
    with ParticleSystem(device=0, seed=SEED) as system:
        pair = Pair(system, _emitter(1.0e6, 0.8, 1.2), seed=SEED, uid=21)
        cap = 1 << 21
        buf = torch.full((cap * 16,), float("nan"), dtype=torch.float32, device="cuda")
        assert pair.gpu.update_path(0)[0] == "range"
        dts = _dts(110, seed=5, spikes=((80, 0.03),))
        for fr, dt in enumerate(dts):
            if fr == 45:
                # (round 6: the ring -- ~740 000 particles by now, component planes, wrapped or not -- is unwrapped AND transposed into the
                # float4 planes of the compacting path here: realloc_segment)
                pair.gpu.attach_instances(buf.data_ptr(), cap, particle_type=0)
                assert pair.gpu.update_path(0)[0] == "general"
                pair.check(exact_all=True, what="right after the attach")
            system.update(dt)
            pair.step_cpu(dt)
            if fr in (30, 44, 46, 65, 79, 80, 81, 95, 109):
                pair.check(exact_all=True, what=f"frame {fr} (dt {dt})")
                if fr > 45:
                    n = pair.gpu.count(0)
                    got = buf[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
                    assert np.array_equal(got, pair.gpu.instances(0).view(np.uint32).reshape(n, 16)), fr
        assert pair.gpu.count(0) > 900000
        assert system.tf_frames() > 30, system.tf_frames()  # (the first ~46 frames hold fewer than tf_min_tiles tiles)
