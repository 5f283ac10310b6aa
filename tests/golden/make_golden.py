#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (committed as JSON).

The reference (Rust + bevy 0.19) cannot be built or imported in this image, so
these vectors come from (a) the reference's own two unit tests, restated with
their literal inputs (src/core.rs:806-834, src/curve.rs:246-258), (b) the
published Random123 known-answer vectors for Philox4x32-10, and (c) the
independent numpy-float32 restatement in np_restatement.py for trajectories the
reference has no test for.  Floats are stored as uint32 bit patterns so the
comparison is bit-exact.  Run from the repo root:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import np_restatement as R  # noqa: E402

f32 = np.float32


def bits(x) -> int:
    return int(np.asarray(x, dtype=np.float32).view(np.uint32))


def vbits(v):
    return [bits(x) for x in np.asarray(v, dtype=np.float32).ravel()]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name)


def emission_kat():
    """reference src/core.rs:806-834 with its literal inputs."""
    timestep, age, last = f32(0.016), f32(0.0), R.F32_MIN
    duration, per = f32(3.0), f32(23.0)
    steps, total = [], 0
    while age <= duration:
        n, nxt = R.compute_emission_count(age, last, duration, 0.0, 1.0, per)
        steps.append([bits(age), bits(last), n, bits(nxt)])
        total += n
        last = nxt
        age = f32(age + timestep)
    assert total in (22, 23), total  # the reference's own assertion
    return {"source": "reference src/core.rs:806-834", "duration": 3.0, "count": 23.0, "offset_start": 0.0,
            "offset_end": 1.0, "steps": steps, "total": total, "reference_accepts": [22, 23]}


def emission_wrap():
    """Global CountOverDuration pacing exactly as spawn_particles drives it
    (src/core.rs:412-424): rem_euclid of the cycle clock, then the count."""
    out = []
    dt = f32(1.0) / f32(60.0)
    for rate, dur, start, end, frames in [
        (160000.0, 1.0, 0.0, 1.0, 150),   # examples/stress_test.rs:113
        (1.0e6, 1.0, 0.0, 1.0, 150),      # BASELINE config 2
        (50000.0, 1.0, 0.0, 1.0, 150),    # BASELINE config 1 "~50k"
        (12.0, 1.0, 0.0, 1.0, 200),       # examples/textures.rs:129
        (300.0, 2.5, 0.25, 0.75, 400),    # offset window
    ]:
        tpc, last = f32(0.0), f32(0.0)
        rows = []
        for _ in range(frames):
            tpc = R.rem_euclid(f32(tpc + dt), dur)
            n, last = R.compute_emission_count(tpc, last, dur, start, end, rate)
            rows.append([bits(tpc), n, bits(last)])
        out.append({"count": rate, "duration": dur, "offset_start": start, "offset_end": end, "dt_bits": bits(dt),
                    "frames": rows})
    # the figures quoted in SURVEY.md §6/§8: one frame of emission is lost per cycle wrap
    per_cycle_160k = sum(r[1] for r in out[0]["frames"][60:120])
    per_cycle_1m = sum(r[1] for r in out[1]["frames"][60:120])
    assert per_cycle_160k == 157334, per_cycle_160k
    assert per_cycle_1m in (983333, 983334), per_cycle_1m
    return {"cases": out, "per_cycle_160k": per_cycle_160k, "per_cycle_1m": per_cycle_1m}


def nested_count_kat():
    """Nested pacing as src/core.rs:490-500 drives it: per-parent age clock,
    last_emitted_age starts at f32::MIN, duration = the parent's lifetime
    (examples/textures.rs:144-156: count 6, window [0, 0.1])."""
    cases = []
    dt = f32(1.0) / f32(60.0)
    for count, start, end, life in [(6.0, 0.0, 0.1, 3.0), (20.0, 0.0, 0.5, 2.0), (5.0, 0.2, 0.9, 1.3)]:
        age, last = f32(0.0), R.F32_MIN
        rows = []
        while age < f32(life):
            n, last = R.compute_emission_count(age, last, life, start, end, count)
            rows.append([bits(age), n, bits(last)])
            age = f32(age + dt)
        cases.append({"count": count, "offset_start": start, "offset_end": end, "lifetime": life, "rows": rows,
                      "total": sum(r[1] for r in rows)})
    return {"cases": cases}


STRESS_GRADIENT = [  # examples/stress_test.rs:100-106
    (0.0, (10.0, 7.0, 1.0, 1.0)), (0.7, (3.0, 1.0, 1.0, 1.0)), (0.8, (1.0, 0.3, 0.3, 1.0)),
    (0.9, (0.3, 0.3, 0.3, 1.0)), (1.0, (0.1, 0.1, 0.1, 0.0)),
]


def curve_kat():
    red, green, blue = (1.0, 0.0, 0.0, 1.0), (0.0, 1.0, 0.0, 1.0), (0.0, 0.0, 1.0, 1.0)
    ref = []
    for t, want in [(0.0, red), (0.5, green), (1.0, blue)]:  # reference src/curve.rs:255-257
        got = R.gradient_sample(1, [red, green, blue], [], t)
        assert list(got) == list(np.asarray(want, dtype=f32)), (t, got)
        ref.append({"t": t, "rgba_bits": vbits(got)})
    ts = [0.0, 0.1, 0.35, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 1.0, -0.5, 1.5, 0.69999999, 0.70000005]
    times = [t for t, _ in STRESS_GRADIENT]
    cols = [c for _, c in STRESS_GRADIENT]
    uneven = [{"t_bits": bits(t), "rgba_bits": vbits(R.gradient_sample(2, cols, times, t))} for t in ts]
    even5 = [{"t_bits": bits(t), "rgba_bits": vbits(R.gradient_sample(1, cols, [], t))} for t in ts]
    lin = [{"t_bits": bits(t), "v_bits": bits(R.curve_sample(1, [1.0, 2.0], [], t))} for t in ts]
    e3 = [{"t_bits": bits(t), "v_bits": bits(R.curve_sample(1, [0.0, 1.0, 0.25], [], t))} for t in ts]
    un = [{"t_bits": bits(t), "v_bits": bits(R.curve_sample(2, [1.0, 1.2, 0.0], [0.0, 0.8, 1.0], t))} for t in ts]
    # unsorted + duplicate + non-finite times exercise UnevenCore::new normalisation
    messy_t, messy_v = [0.5, 0.0, 1.0, 0.5, float("inf")], [3.0, 1.0, 2.0, 9.0, 7.0]
    messy = [{"t_bits": bits(t), "v_bits": bits(R.curve_sample(2, messy_v, messy_t, t))} for t in ts]
    return {"reference_test": {"source": "reference src/curve.rs:246-258", "colors": [red, green, blue], "samples": ref},
            "stress_gradient": {"times": times, "colors": cols, "uneven": uneven, "even": even5},
            "f32_even_2": {"values": [1.0, 2.0], "samples": lin},
            "f32_even_3": {"values": [0.0, 1.0, 0.25], "samples": e3},
            "f32_uneven_3": {"values": [1.0, 1.2, 0.0], "times": [0.0, 0.8, 1.0], "samples": un},
            "f32_uneven_messy": {"values": messy_v, "times": [t if np.isfinite(t) else "inf" for t in messy_t],
                                 "samples": messy}}


def philox_kat():
    # Random123 kat_vectors, philox4x32-10 (published known answers)
    published = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for c, k, want in published:
        assert R.philox4x32_10(c, k) == want, (c, k, [hex(x) for x in R.philox4x32_10(c, k)])
    uniforms = []
    for seed, uid, ei, serial in [(0xC0FFEE, 0, 0, 0), (0xC0FFEE, 0, 0, 1), (0xC0FFEE, 7, 2, 123456789),
                                  (1, 4095, 1, (1 << 33) + 5)]:
        u = []
        for b in range(3):
            o = R.philox4x32_10([serial & 0xFFFFFFFF, serial >> 32, ei, b], [seed, uid])
            u += [bits(R.unit_f32(x)) for x in o]
        uniforms.append({"seed": seed, "uid": uid, "emission_index": ei, "serial": serial, "u_bits": u})
    return {"source": "Random123 kat_vectors (philox4x32 10 rounds)",
            "published": [{"ctr": c, "key": k, "out": w} for c, k, w in published], "spawn_uniforms": uniforms}


def update_kat():
    """Hand-derivable single-particle updates (src/core.rs:591-658)."""
    dt = f32(1.0) / f32(60.0)
    default_ps = {"scale_curve": (0, [1.0], []), "acceleration": (0.0, -9.81, 0.0), "linear_drag": 0.2,
                  "angular_acceleration": (0.0, 0.0, 0.0), "angular_drag": 0.2,
                  "base_color": (0, [(1.0, 1.0, 1.0, 1.0)], []), "emissive_color": (0, [(0.0, 0.0, 0.0, 1.0)], [])}
    stress_ps = dict(default_ps, linear_drag=0.1,
                     base_color=(2, [c for _, c in STRESS_GRADIENT], [t for t, _ in STRESS_GRADIENT]))
    lin_ps = dict(default_ps, scale_curve=(1, [1.0, 2.0], []),
                  base_color=(1, [(1.0, 1.0, 1.0, 1.0), (0.0, 0.0, 0.0, 0.0)], []),
                  emissive_color=(1, [(4.0, 2.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)], []))
    base = {"position": (0.0, 0.0, 0.0), "velocity": (0.0, 10.0, 0.0), "angular_velocity": (0.0, 0.0, 0.0),
            "initial_scale": 0.05, "scale": 0.05, "age": 0.0, "lifetime": 1.0}
    cases = []
    for name, ps, p in [
        ("defaults_first_frame", default_ps, base),
        ("stress_mid", stress_ps, dict(base, position=(0.1, 2.0, -0.3), velocity=(1.5, 3.25, -0.75), age=0.7216)),
        ("linear_curves", lin_ps, dict(base, position=(1.0, 2.0, 3.0), velocity=(-2.0, 0.5, 4.0), age=0.31, lifetime=1.7)),
        # death boundary: 0.98333335 + 1/60 rounds to exactly 1.0 >= lifetime
        ("dies_exactly_at_lifetime", default_ps, dict(base, age=float(f32(1.0) - dt))),
        ("survives_one_ulp_below", default_ps, dict(base, age=float(np.nextafter(f32(f32(1.0) - dt), f32(0))))),
    ]:
        alive, dead = R.update_one(p, ps, dt)
        rec = {"name": name, "dt_bits": bits(dt), "in": {k: (vbits(v) if isinstance(v, tuple) else bits(v)) for k, v in p.items()},
               "settings": {"scale_curve": list(ps["scale_curve"]), "acceleration": list(ps["acceleration"]),
                            "linear_drag": ps["linear_drag"], "angular_drag": ps["angular_drag"],
                            "base_color": [ps["base_color"][0], [list(c) for c in ps["base_color"][1]], list(ps["base_color"][2])],
                            "emissive_color": [ps["emissive_color"][0], [list(c) for c in ps["emissive_color"][1]], list(ps["emissive_color"][2])]}}
        if alive is None:
            rec["alive"] = False
            rec["out"] = {"age": bits(dead["age"])}
        else:
            rec["alive"] = True
            rec["out"] = {k: (vbits(alive[k]) if np.ndim(alive[k]) else bits(alive[k]))
                          for k in ("position", "velocity", "angular_velocity", "scale", "age", "base_color", "emissive_color")}
        cases.append(rec)
    assert cases[3]["alive"] is False and cases[4]["alive"] is True
    return {"source": "reference src/core.rs:591-658 (hand-derived, numpy float32)", "cases": cases}


def trajectories():
    """Multi-frame trajectories of the array-oriented numpy restatement (np_sim.py) over scenarios.py: particle
    state of every type at the checkpoint frames, stored as raw float32 (bit patterns preserved) in one .npz."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import np_sim
    import scenarios

    out = {}
    summary = {}
    for name, make in scenarios.ALL.items():
        sc = make()
        sim = np_sim.Spawner(sc["spawner"], scenarios.SEED, sc["uid"], sc["transform"], sc["modifier"])
        sim.parent_velocity = np.asarray(sc["parent_velocity"], dtype=f32)
        sim.colliders = sc.get("colliders", [])
        counts = []
        for fr in range(sc["frames"]):
            sim.step(f32(sc["dts"][fr % len(sc["dts"])]))
            if fr in sc["checkpoints"]:
                for t, p in enumerate(sim.particles):
                    for k, v in p.items():
                        out[f"{name}/f{fr}/t{t}/{k}"] = np.ascontiguousarray(v, dtype=f32)
                    d = sim.destroyed[t]
                    out[f"{name}/f{fr}/t{t}/destroyed_age"] = np.ascontiguousarray(d["age"], dtype=f32)
                    out[f"{name}/f{fr}/t{t}/destroyed_position"] = np.ascontiguousarray(d["position"], dtype=f32)
                    out[f"{name}/f{fr}/t{t}/destroyed_velocity"] = np.ascontiguousarray(d["velocity"], dtype=f32)
                    out[f"{name}/f{fr}/t{t}/destroyed_scale"] = np.ascontiguousarray(d["scale"], dtype=f32)
                counts.append([fr] + [sim.count(t) for t in range(len(sim.particles))])
        summary[name] = counts
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **out)
    print("wrote trajectories.npz", {k: v[-1] for k, v in summary.items()})
    return {"source": "tests/golden/np_sim.py over tests/golden/scenarios.py", "counts_at_checkpoints": summary}


if __name__ == "__main__":
    dump("trajectories.json", trajectories())
    dump("emission_kat.json", emission_kat())
    dump("emission_wrap.json", emission_wrap())
    dump("nested_count_kat.json", nested_count_kat())
    dump("curve_kat.json", curve_kat())
    dump("philox_kat.json", philox_kat())
    dump("update_kat.json", update_kat())
