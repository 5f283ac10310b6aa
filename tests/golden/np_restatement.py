"""Independent numpy-float32 restatement of the scalar pieces of the hot path.

Second opinion for the C oracle (oracle/fw_oracle.c): written separately, in a
different language, from the same reference lines.  Used only to GENERATE and
CROSS-CHECK the golden vectors under tests/golden/ (SURVEY.md §8c).  Every
operation is an explicit np.float32 op (IEEE single, no FMA).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
F32_MIN = f32(np.finfo(np.float32).min)
PI = f32(np.pi)


def rem_euclid(a, b):
    a, b = f32(a), f32(b)
    r = np.fmod(a, b)
    return f32(r + np.abs(b)) if r < 0 else f32(r)


def div_euclid(a, b):
    a, b = f32(a), f32(b)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.trunc(f32(a / b))
        if np.fmod(a, b) < 0:
            return f32(q - f32(1)) if b > 0 else f32(q + f32(1))
    return f32(q)


def as_usize(x):
    x = f32(x)
    if np.isnan(x) or x <= 0:
        return 0
    if x >= f32(18446744073709551616.0):
        return (1 << 64) - 1
    return int(x)


def compute_emission_count(t, last, dur, start, end, count):
    """reference src/core.rs:553-575"""
    t, last, dur, start, end, count = map(f32, (t, last, dur, start, end, count))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        percent_passed = f32(t / dur)
        last_percent = f32(last / dur)
        since = f32(np.fmin(percent_passed, end) - np.fmax(last_percent, start))
        between = f32(f32(end - start) / count)
        times = div_euclid(since, between)
        n = as_usize(times)
        next_percent = f32(np.fmax(last_percent, start) + f32(times * between))
        return n, f32(next_percent * dur)


def _even_interp(n, t):
    subdivs = n - 1
    step = f32(f32(1.0) / f32(subdivs))
    steps = f32(f32(t - f32(0)) / step)
    if steps <= 0:
        return 0, None
    if steps >= f32(subdivs):
        return n - 1, None
    lo = int(np.floor(steps))
    return lo, f32(steps - np.trunc(steps))


def _uneven_interp(times, t):
    times = [f32(x) for x in times]
    idx = sum(1 for x in times if x < t)
    if idx < len(times) and times[idx] == t:
        return idx, None
    if idx == 0:
        return 0, None
    if idx >= len(times):
        return len(times) - 1, None
    lo, hi = times[idx - 1], times[idx]
    return idx - 1, f32(f32(t - lo) / f32(hi - lo))


def normalize_uneven(times, values):
    pairs = [(f32(t), v) for t, v in zip(times, values) if np.isfinite(t)]
    pairs.sort(key=lambda p: p[0])  # python sort is stable
    out = []
    for t, v in pairs:
        if out and out[-1][0] == t:
            continue
        out.append((t, v))
    return [p[0] for p in out], [p[1] for p in out]


def curve_sample(kind, values, times, t):
    """FireworkCurve<f32>::sample_clamped, reference src/curve.rs:26-32"""
    t = f32(t)
    values = [f32(v) for v in values]
    if kind == 0 or len(values) == 1:
        return values[0]
    if kind == 1:
        t = f32(min(max(t, f32(0)), f32(1)))
        lo, s = _even_interp(len(values), t)
    else:
        times, values = normalize_uneven(times, values)
        t = f32(min(max(t, times[0]), times[-1]))
        lo, s = _uneven_interp(times, t)
    if s is None:
        return values[lo]
    a, b = values[lo], values[lo + 1]
    return f32(f32(a * f32(f32(1) - s)) + f32(b * s))  # VectorSpace::lerp: self * (1. - t) + rhs * t


def gradient_sample(kind, colors, times, t):
    """FireworkGradient<LinearRgba>::sample_clamped, reference src/curve.rs:111-114,156-158"""
    t = f32(t)
    colors = [np.asarray(c, dtype=f32) for c in colors]
    if kind == 0 or len(colors) == 1:
        return colors[0]
    if kind == 1:
        lo, s = _even_interp(len(colors), t)
    else:
        times, colors = normalize_uneven(times, colors)
        lo, s = _uneven_interp(times, t)
    if s is None:
        return colors[lo]
    nf = f32(f32(1) - s)
    return (colors[lo] * nf + colors[lo + 1] * s).astype(f32)


def philox4x32_10(ctr, key):
    c = [int(x) & 0xFFFFFFFF for x in ctr]
    k = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


def unit_f32(u32):
    return f32(f32(u32 >> 8) * f32(2.0 ** -24))


def update_one(p, ps, dt):
    """update_particles body for one particle with zero angular velocity
    (reference src/core.rs:591-658, non-avian arm); p/ps are dicts."""
    dt = f32(dt)
    q = dict(p)
    q["age"] = f32(f32(p["age"]) + dt)
    if q["age"] >= f32(p["lifetime"]):
        return None, q
    age_percent = f32(q["age"] / f32(p["lifetime"]))
    sf = curve_sample(*ps["scale_curve"], age_percent)
    q["scale"] = f32(f32(p["initial_scale"]) * sf)
    pos = np.asarray(p["position"], dtype=f32)
    vel = np.asarray(p["velocity"], dtype=f32)
    acc = np.asarray(ps["acceleration"], dtype=f32)
    q["position"] = (pos + (vel * dt).astype(f32)).astype(f32)
    q["velocity"] = (vel + ((acc - (vel * f32(ps["linear_drag"])).astype(f32)).astype(f32) * dt).astype(f32)).astype(f32)
    w = np.asarray(p["angular_velocity"], dtype=f32)
    assert not w.any(), "np restatement covers the zero-rotation path only"
    aa = np.asarray(ps["angular_acceleration"], dtype=f32)
    q["angular_velocity"] = (w + ((aa - (f32(ps["angular_drag"]) * w).astype(f32)).astype(f32) * dt).astype(f32)).astype(f32)
    q["base_color"] = gradient_sample(*ps["base_color"], age_percent)
    q["emissive_color"] = gradient_sample(*ps["emissive_color"], age_percent)
    return q, None
