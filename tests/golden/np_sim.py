"""Array-oriented numpy-float32 restatement of the whole hot path (one spawner).

A SECOND, INDEPENDENT statement of the reference algorithm, written from the reference
source lines (paths relative to /root/reference) and the published third-party
algorithms -- not from oracle/fw_oracle.c or csrc/fw_math.h.  It differs from those in
language (numpy), in data shape (whole-array operations over structure-of-arrays state
instead of a per-particle loop over records) and in libm (numpy's own float32 sin/cos),
so a misreading shared by the C oracle and the HIP kernels does not pass silently here.

  spawn_particles          src/core.rs:367-551   -> Spawner.spawn
  compute_emission_count   src/core.rs:553-575   -> emission_count (vectorised)
  update_particles         src/core.rs:577-670   -> Spawner.update
  active()                 src/core.rs:288-302   -> Spawner.active
  EmissionShape            src/emission_shape.rs:18-39
  curves / gradients       src/curve.rs + bevy_math cores (np_restatement.py, scalar; vectorised here)
  glam 0.32 scalar Quat / Vec3 algorithms (from_scaled_axis, Hamilton product, Quat * Vec3,
  from_rotation_arc, from_rotation_y, normalize_or_zero)

What is a DEFINITION rather than a restatement (the reference draws from an unseeded
thread-local RNG, so spawn attributes are unpinnable in principle): the Philox4x32-10
stream (key = seed, spawner uid; counter = serial lo, serial hi, emission index, block;
12 uniforms per particle in the reference's draw order core.rs:438-466) and our reading
of bevy_utilitarian's RandVec3 cone / PitchYaw::to_unit_vec.

Every value is np.float32; numpy never fuses a*b+c.  Used only by make_golden.py to
generate fixtures and by tests to cross-check; never imported by the product.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
F32_MIN = f32(np.finfo(np.float32).min)
PI = f32(np.pi)
TWO = f32(2.0)
ONE = f32(1.0)
HALF = f32(0.5)
ZERO = f32(0.0)


def _a(x):
    return np.asarray(x, dtype=f32)


# ---------------------------------------------------------------- Rust scalar semantics (vectorised)
def rem_euclid(a, b):
    a, b = _a(a), _a(b)
    r = np.fmod(a, b)
    return np.where(r < 0, (r + np.abs(b)).astype(f32), r).astype(f32)


def div_euclid(a, b):
    a, b = _a(a), _a(b)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.trunc((a / b).astype(f32))
        r = np.fmod(a, b)
    adj = np.where(b > 0, q - ONE, q + ONE).astype(f32)
    return np.where(r < 0, adj, q).astype(f32)


def as_usize(x):
    """`x as usize`: NaN -> 0, negative -> 0, saturating (values here stay far below 2^63)."""
    x = _a(x)
    ok = np.isfinite(x) & (x > 0)
    big = x >= f32(2.0 ** 63)
    out = np.zeros(x.shape, dtype=np.uint64)
    out[ok & ~big] = x[ok & ~big].astype(np.uint64)
    out[(x > 0) & (big | np.isinf(x))] = np.uint64(2 ** 63)  # never reached by the fixtures
    return out


def emission_count(t, last, dur, start, end, count):
    """src/core.rs:553-575 for arrays of (t, last, dur)."""
    t, last, dur = _a(t), _a(last), _a(dur)
    start, end, count = f32(start), f32(end), f32(count)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        percent_passed = (t / dur).astype(f32)
        last_percent = (last / dur).astype(f32)
        lo = np.fmax(last_percent, start).astype(f32)
        since = (np.fmin(percent_passed, end).astype(f32) - lo).astype(f32)
        between = f32(f32(end - start) / count)
        times = div_euclid(since, between)
        n = as_usize(times)
        nxt = ((lo + (times * between).astype(f32)).astype(f32) * dur).astype(f32)
    return n, nxt


# ---------------------------------------------------------------- glam scalar algorithms, columns of arrays
def dot3(a, b):
    return ((a[..., 0] * b[..., 0]).astype(f32) + (a[..., 1] * b[..., 1]).astype(f32)).astype(f32) + (a[..., 2] * b[..., 2]).astype(f32)


def cross3(a, b):
    return np.stack([
        (a[..., 1] * b[..., 2]).astype(f32) - (b[..., 1] * a[..., 2]).astype(f32),
        (a[..., 2] * b[..., 0]).astype(f32) - (b[..., 2] * a[..., 0]).astype(f32),
        (a[..., 0] * b[..., 1]).astype(f32) - (b[..., 0] * a[..., 1]).astype(f32),
    ], axis=-1).astype(f32)


def normalize_or_zero(v):
    with np.errstate(divide="ignore", invalid="ignore"):
        rcp = (ONE / np.sqrt(dot3(v, v)).astype(f32)).astype(f32)
    ok = np.isfinite(rcp) & (rcp > 0)
    safe = np.where(ok, rcp, ZERO).astype(f32)
    return np.where(ok[..., None], (v * safe[..., None]).astype(f32), f32(0)).astype(f32)


def quat_mul(a, b):
    """Hamilton product, glam scalar Quat::mul_quat."""
    x0, y0, z0, w0 = (a[..., i] for i in range(4))
    x1, y1, z1, w1 = (b[..., i] for i in range(4))
    m = lambda p, q: (p * q).astype(f32)
    return np.stack([
        ((m(w0, x1) + m(x0, w1)).astype(f32) + m(y0, z1)).astype(f32) - m(z0, y1),
        ((m(w0, y1) - m(x0, z1)).astype(f32) + m(y0, w1)).astype(f32) + m(z0, x1),
        ((m(w0, z1) + m(x0, y1)).astype(f32) - m(y0, x1)).astype(f32) + m(z0, w1),
        ((m(w0, w1) - m(x0, x1)).astype(f32) - m(y0, y1)).astype(f32) - m(z0, z1),
    ], axis=-1).astype(f32)


def quat_mul_vec3(q, v):
    """glam scalar Quat::mul_vec3: v*(w^2 - b.b) + b*(2 v.b) + (b x v)*(2 w)."""
    b = q[..., :3]
    w = q[..., 3]
    b2 = dot3(b, b)
    k0 = ((w * w).astype(f32) - b2).astype(f32)
    k1 = (dot3(v, b) * TWO).astype(f32)
    c = cross3(b, v)
    k2 = (w * TWO).astype(f32)
    t0 = (v * k0[..., None]).astype(f32)
    t1 = (b * k1[..., None]).astype(f32)
    t2 = (c * k2[..., None]).astype(f32)
    return ((t0 + t1).astype(f32) + t2).astype(f32)


def quat_from_axis_angle(axis, angle):
    h = (angle * HALF).astype(f32)
    s, c = np.sin(h).astype(f32), np.cos(h).astype(f32)
    return np.concatenate([(axis * s[..., None]).astype(f32), c[..., None]], axis=-1).astype(f32)


def quat_from_scaled_axis(v):
    """glam Quat::from_scaled_axis: identity for a zero vector, else axis = v / |v|, angle = |v|."""
    length = np.sqrt(dot3(v, v)).astype(f32)
    zero = length == 0
    safe = np.where(zero, ONE, length).astype(f32)
    q = quat_from_axis_angle((v / safe[..., None]).astype(f32), length)
    ident = np.zeros(q.shape, dtype=f32)
    ident[..., 3] = ONE
    return np.where(zero[..., None], ident, q).astype(f32)


def quat_from_rotation_arc(frm, to):
    """glam Quat::from_rotation_arc for single vectors (settings-time)."""
    frm, to = _a(frm), _a(to)
    one_minus_eps = f32(ONE - f32(TWO * f32(np.finfo(np.float32).eps)))
    d = dot3(frm, to)
    if d > one_minus_eps:
        return np.array([0, 0, 0, 1], dtype=f32)
    if d < -one_minus_eps:
        sign = f32(np.copysign(ONE, frm[2]))  # Vec3::any_orthonormal_vector
        a = f32(f32(-1.0) / f32(sign + frm[2]))
        b = f32(f32(frm[0] * frm[1]) * a)
        axis = np.array([b, f32(sign + f32(f32(frm[1] * frm[1]) * a)), f32(-frm[1])], dtype=f32)
        return quat_from_axis_angle(axis, _a(PI))
    c = cross3(frm, to)
    w = f32(ONE + d)
    ln = np.sqrt(f32(f32(f32(f32(c[0] * c[0]) + f32(c[1] * c[1])) + f32(c[2] * c[2])) + f32(w * w))).astype(f32)
    inv = f32(ONE / ln)
    return np.array([c[0] * inv, c[1] * inv, c[2] * inv, w * inv], dtype=f32)


# ---------------------------------------------------------------- RNG stream (definition, see module docstring)
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & mask for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        c0, c1, c2, c3 = n0, p1 & mask, n2, p0 & mask
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c0, c1, c2, c3


def spawn_uniforms(seed, uid, emission_index, serials):
    """[n, 12] uniforms in [0, 1): rand 0.9's f32 = (u32 >> 8) * 2^-24."""
    serials = np.asarray(serials, dtype=np.uint64)
    lo, hi = serials & np.uint64(0xFFFFFFFF), serials >> np.uint64(32)
    cols = []
    for blk in range(3):
        outs = philox4x32_10(lo, hi, np.full_like(lo, emission_index), np.full_like(lo, blk), seed, uid)
        cols += [((o >> np.uint64(8)).astype(f32) * f32(2.0 ** -24)).astype(f32) for o in outs]
    return np.stack(cols, axis=-1).astype(f32)


def randf32(r, u):
    """RandF32::generate = random * (max - min) + min."""
    return ((u * f32(f32(r.max) - f32(r.min))).astype(f32) + f32(r.min)).astype(f32)


def randvec3(r, u_angle, u_radius, u_mag):
    n = len(u_mag)
    if f32(r.spread) > 0:
        ang = ((u_angle * TWO).astype(f32) * PI).astype(f32)
        rad = (u_radius * f32(r.spread)).astype(f32)
        sr, cr = np.sin(rad).astype(f32), np.cos(rad).astype(f32)
        local = np.stack([(sr * np.cos(ang).astype(f32)).astype(f32), cr, (sr * np.sin(ang).astype(f32)).astype(f32)], axis=-1)
        arc = quat_from_rotation_arc((0.0, 1.0, 0.0), r.direction)
        d = quat_mul_vec3(np.broadcast_to(arc, (n, 4)), local)
    else:
        d = np.broadcast_to(_a(r.direction), (n, 3)).astype(f32)
    m = randf32(r.magnitude, u_mag)
    return (d * m[:, None]).astype(f32)


def shape_points(shape, u):
    """EmissionShape::generate_point (emission_shape.rs:18-39); u = first three uniforms."""
    n = len(u)
    if shape.kind == 1:  # Sphere
        pitch = ((u[:, 0] * TWO).astype(f32) * PI).astype(f32)
        yaw = (u[:, 1] * PI).astype(f32)
        cp, sp = np.cos(pitch).astype(f32), np.sin(pitch).astype(f32)
        unit = np.stack([(cp * np.sin(yaw).astype(f32)).astype(f32), sp, (cp * np.cos(yaw).astype(f32)).astype(f32)], axis=-1)
        return ((unit * u[:, 2][:, None]).astype(f32) * f32(shape.radius)).astype(f32)
    if shape.kind == 2:  # Circle
        ang = ((u[:, 0] * TWO).astype(f32) * PI).astype(f32)
        h = (ang * HALF).astype(f32)
        z = np.zeros(n, dtype=f32)
        qy = np.stack([z, np.sin(h).astype(f32), z, np.cos(h).astype(f32)], axis=-1)  # Quat::from_rotation_y
        arc = quat_from_rotation_arc((0.0, 1.0, 0.0), shape.normal)
        q = quat_mul(np.broadcast_to(arc, (n, 4)), qy)
        v = np.stack([(u[:, 1] * f32(shape.radius)).astype(f32), z, z], axis=-1)
        return quat_mul_vec3(q, v)
    return np.zeros((n, 3), dtype=f32)


# ---------------------------------------------------------------- curves over arrays
def _normalize_uneven(times, values):
    pairs = [(f32(t), v) for t, v in zip(times, values) if np.isfinite(t)]
    pairs.sort(key=lambda p: p[0])
    out = []
    for t, v in pairs:
        if out and out[-1][0] == t:
            continue
        out.append((t, v))
    return [p[0] for p in out], [p[1] for p in out]


def _interp_index(kind, n_keys, times, t):
    """-> (lo index, s, between mask) per element; EvenCore / UnevenCore::sample_with."""
    t = _a(t)
    if kind == 1:
        subdivs = n_keys - 1
        step = f32(ONE / f32(subdivs))
        steps = ((t - ZERO).astype(f32) / step).astype(f32)
        left, right = steps <= 0, steps >= f32(subdivs)
        lo = np.clip(np.floor(steps), 0, n_keys - 2).astype(np.int64)
        s = (steps - np.trunc(steps)).astype(f32)
        between = ~(left | right)
        lo = np.where(left, 0, np.where(right, n_keys - 1, lo))
        return lo, s, between
    ts = np.asarray(times, dtype=f32)
    idx = np.searchsorted(ts, t, side="left")  # number of times < t
    exact = (idx < n_keys) & (ts[np.minimum(idx, n_keys - 1)] == t)
    left, right = idx == 0, idx >= n_keys
    between = ~(exact | left | right)
    lo_b = np.clip(idx - 1, 0, n_keys - 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = ((t - ts[lo_b]).astype(f32) / (ts[lo_b + 1] - ts[lo_b]).astype(f32)).astype(f32)
    lo = np.where(exact, np.minimum(idx, n_keys - 1), np.where(left, 0, np.where(right, n_keys - 1, lo_b)))
    return lo, s, between


def curve_sample(curve, t):
    """FireworkCurve<f32>::sample_clamped (curve.rs:26-32): clamp to the domain, VectorSpace::lerp."""
    t = _a(t)
    vals, times, kind = list(curve.values), list(curve.times), curve.kind
    if kind == 2:
        times, vals = _normalize_uneven(times, vals)
    if kind == 0 or len(vals) == 1:
        return np.full(t.shape, f32(vals[0]), dtype=f32)
    v = np.asarray(vals, dtype=f32)
    if kind == 1:
        t = np.minimum(np.maximum(t, ZERO), ONE).astype(f32)
    else:
        t = np.minimum(np.maximum(t, f32(times[0])), f32(times[-1])).astype(f32)
    lo, s, between = _interp_index(kind, len(v), times, t)
    a, b = v[lo], v[np.minimum(lo + 1, len(v) - 1)]
    mixed = ((a * (ONE - s).astype(f32)).astype(f32) + (b * s).astype(f32)).astype(f32)
    return np.where(between, mixed, a).astype(f32)


def gradient_sample(grad, t):
    """FireworkGradient<LinearRgba>::sample_clamped (curve.rs:111-114,156-158); Mix: a*(1-f) + b*f."""
    t = _a(t)
    cols, times, kind = [tuple(c) for c in grad.colors], list(grad.times), grad.kind
    if kind == 2:
        times, cols = _normalize_uneven(times, cols)
    c = np.asarray(cols, dtype=f32)
    if kind == 0 or len(c) == 1:
        return np.broadcast_to(c[0], t.shape + (4,)).astype(f32)
    lo, s, between = _interp_index(kind, len(c), times, t)
    a, b = c[lo], c[np.minimum(lo + 1, len(c) - 1)]
    nf = (ONE - s).astype(f32)
    mixed = ((a * nf[..., None]).astype(f32) + (b * s[..., None]).astype(f32)).astype(f32)
    return np.where(between[..., None], mixed, a).astype(f32)


# ---------------------------------------------------------------- particle_collision (core.rs:744-800)
def _len3(v):
    return np.sqrt(dot3(v, v)).astype(f32)


def _normalize3(v):  # glam Vec3::normalize = self * length_recip()
    with np.errstate(divide="ignore", invalid="ignore"):
        return (v * (ONE / _len3(v)).astype(f32)[..., None]).astype(f32)


def _project_onto(a, rhs):  # glam: rhs * self.dot(rhs) * rhs.dot(rhs).recip()
    with np.errstate(divide="ignore", invalid="ignore"):
        rcp = (ONE / dot3(rhs, rhs)).astype(f32)
    return ((rhs * dot3(a, rhs)[..., None]).astype(f32) * rcp[..., None]).astype(f32)


def cast_ray(colliders, mask, origin, d, max_distance):
    """nearest `solid = true` hit of the rays origin + t d over the analytic colliders (include/firework_hip.h:
    fw_collider) -> (hit mask, distance, normal); a ray starting inside a solid hits at 0 with a zero normal"""
    n = len(origin)
    best_t = np.full(n, np.inf, dtype=f32)
    best_n = np.zeros((n, 3), dtype=f32)
    found = np.zeros(n, dtype=bool)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for c in colliders:
            if not (int(c.layers) & int(mask)):
                continue
            cpos = _a(c.position)
            hit = np.zeros(n, dtype=bool)
            t = np.zeros(n, dtype=f32)
            nrm = np.zeros((n, 3), dtype=f32)
            if c.kind == 0:  # plane
                nn = np.broadcast_to(_a(c.normal), (n, 3))
                dnd = dot3(nn, (cpos - origin).astype(f32))
                inside = dnd > 0
                tt = (dnd / dot3(nn, d)).astype(f32)
                ok = ~inside & (tt >= 0) & (tt <= max_distance)
                hit = inside | ok
                t = np.where(inside, ZERO, tt).astype(f32)
                nrm = np.where(ok[:, None], nn, ZERO).astype(f32)
            elif c.kind == 1:  # sphere
                dc = (origin - cpos).astype(f32)
                a_, b_ = dot3(d, d), dot3(dc, d)
                cc = (dot3(dc, dc) - f32(f32(c.radius) * f32(c.radius))).astype(f32)
                inside = cc <= 0
                delta = ((b_ * b_).astype(f32) - (a_ * cc).astype(f32)).astype(f32)
                tt = (((-b_).astype(f32) - np.sqrt(np.maximum(delta, ZERO)).astype(f32)).astype(f32) / a_).astype(f32)
                ok = ~inside & ~(b_ > 0) & (delta >= 0) & (tt >= 0) & (tt <= max_distance)
                p_ = ((origin + (d * tt[:, None]).astype(f32)).astype(f32) - cpos).astype(f32)
                hit = inside | ok
                t = np.where(inside, ZERO, tt).astype(f32)
                nrm = np.where(ok[:, None], _normalize3(p_), ZERO).astype(f32)
            elif c.kind == 3:  # cylinder (avian Collider::cylinder(radius, height), examples/textures.rs:195): axis = local Y
                q = _a(c.rotation)
                qi = np.broadcast_to(np.array([-q[0], -q[1], -q[2], q[3]], dtype=f32), (n, 4))
                ol = quat_mul_vec3(qi, (origin - cpos).astype(f32))
                dl = quat_mul_vec3(qi, d)
                hh, rr = f32(c.half_extents[1]), f32(f32(c.radius) * f32(c.radius))
                ox, oy, oz, dx, dy, dz = ol[:, 0], ol[:, 1], ol[:, 2], dl[:, 0], dl[:, 1], dl[:, 2]
                c2 = (((ox * ox).astype(f32) + (oz * oz).astype(f32)).astype(f32) - rr).astype(f32)
                inside = (np.abs(oy) <= hh) & (c2 <= 0)
                tnear = np.full(n, -np.inf, dtype=f32)
                tfar = np.full(n, np.inf, dtype=f32)
                side = np.zeros(n, dtype=np.int64)
                sign = np.zeros(n, dtype=f32)
                par = dy == 0
                dead = par & (np.abs(oy) > hh)
                inv = (ONE / dy).astype(f32)
                t1 = ((-hh - oy).astype(f32) * inv).astype(f32)
                t2 = ((hh - oy).astype(f32) * inv).astype(f32)
                sw = t1 > t2
                lo_, hi_ = np.where(sw, t2, t1), np.where(sw, t1, t2)
                live = ~dead & ~par
                upd = live & (lo_ > tnear)
                tnear = np.where(upd, lo_, tnear).astype(f32)
                sign = np.where(upd, np.where(sw, ONE, f32(-1.0)), sign).astype(f32)
                tfar = np.where(live & (hi_ < tfar), hi_, tfar).astype(f32)
                dead |= live & (tnear > tfar)
                a_ = ((dx * dx).astype(f32) + (dz * dz).astype(f32)).astype(f32)
                b_ = ((ox * dx).astype(f32) + (oz * dz).astype(f32)).astype(f32)
                apar = a_ == 0
                dead |= ~dead & apar & (c2 > 0)
                disc = ((b_ * b_).astype(f32) - (a_ * c2).astype(f32)).astype(f32)
                live = ~dead & ~apar
                dead |= live & ~(disc >= 0)
                live = ~dead & ~apar
                sq = np.sqrt(np.maximum(disc, ZERO)).astype(f32)
                s1 = (((-b_).astype(f32) - sq).astype(f32) / a_).astype(f32)
                s2 = (((-b_).astype(f32) + sq).astype(f32) / a_).astype(f32)
                upd = live & (s1 > tnear)
                tnear = np.where(upd, s1, tnear).astype(f32)
                side = np.where(upd, 1, side)
                tfar = np.where(live & (s2 < tfar), s2, tfar).astype(f32)
                dead |= live & (tnear > tfar)
                ok = ~inside & ~dead & (tnear >= 0) & (tnear <= max_distance)
                rad = np.stack([(ox + (dx * tnear).astype(f32)).astype(f32), np.zeros(n, dtype=f32), (oz + (dz * tnear).astype(f32)).astype(f32)], axis=1)
                cap = np.stack([np.zeros(n, dtype=f32), sign, np.zeros(n, dtype=f32)], axis=1)
                nl = np.where((side == 1)[:, None], _normalize3(rad), cap).astype(f32)
                hit = inside | ok
                t = np.where(inside, ZERO, tnear).astype(f32)
                nrm = np.where(ok[:, None], quat_mul_vec3(np.broadcast_to(q, (n, 4)), nl), ZERO).astype(f32)
            elif c.kind == 4:  # cone (avian Collider::cone(radius, height), examples/textures.rs:211): base at y = -h/2, apex at +h/2
                q = _a(c.rotation)
                qi = np.broadcast_to(np.array([-q[0], -q[1], -q[2], q[3]], dtype=f32), (n, 4))
                ol = quat_mul_vec3(qi, (origin - cpos).astype(f32))
                dl = quat_mul_vec3(qi, d)
                hh, rr = f32(c.half_extents[1]), f32(f32(c.radius) * f32(c.radius))
                k = f32(f32(c.radius) / f32(hh + hh))
                k2 = f32(k * k)
                ox, oy, oz, dx, dy, dz = ol[:, 0], ol[:, 1], ol[:, 2], dl[:, 0], dl[:, 1], dl[:, 2]
                wy = (oy - hh).astype(f32)
                cq = (((ox * ox).astype(f32) + (oz * oz).astype(f32)).astype(f32) - (k2 * (wy * wy).astype(f32)).astype(f32)).astype(f32)
                inside = (oy >= -hh) & (wy <= 0) & (cq <= 0)
                best = np.full(n, np.inf, dtype=f32)
                side = np.full(n, -1, dtype=np.int64)
                tb = ((-hh - oy).astype(f32) / dy).astype(f32)
                px, pz = (ox + (dx * tb).astype(f32)).astype(f32), (oz + (dz * tb).astype(f32)).astype(f32)
                base = (dy > 0) & (oy < -hh) & (((px * px).astype(f32) + (pz * pz).astype(f32)).astype(f32) <= rr)
                best = np.where(base, tb, best).astype(f32)
                side = np.where(base, 0, side)
                a_ = (((dx * dx).astype(f32) + (dz * dz).astype(f32)).astype(f32) - (k2 * (dy * dy).astype(f32)).astype(f32)).astype(f32)
                b_ = (((ox * dx).astype(f32) + (oz * dz).astype(f32)).astype(f32) - (k2 * (wy * dy).astype(f32)).astype(f32)).astype(f32)
                lin = a_ == 0
                disc = ((b_ * b_).astype(f32) - (a_ * cq).astype(f32)).astype(f32)
                sq = np.sqrt(np.maximum(disc, ZERO)).astype(f32)
                quad = ~lin & (disc >= 0)
                ta = np.where(lin, np.where(b_ != 0, ((-cq).astype(f32) / (b_ + b_).astype(f32)).astype(f32), f32(np.inf)),
                              np.where(quad, (((-b_).astype(f32) - sq).astype(f32) / a_).astype(f32), f32(np.inf))).astype(f32)
                tb2 = np.where(quad, (((-b_).astype(f32) + sq).astype(f32) / a_).astype(f32), f32(np.inf)).astype(f32)
                for tt in (ta, tb2):
                    yy = (oy + (dy * tt).astype(f32)).astype(f32)
                    good = (tt >= 0) & (tt < np.inf) & (yy >= -hh) & (yy <= hh) & (tt < best)
                    best = np.where(good, tt, best).astype(f32)
                    side = np.where(good, 1, side)
                ok = ~inside & (side >= 0) & (best <= max_distance)
                w = np.stack([(ox + (dx * best).astype(f32)).astype(f32), ((oy + (dy * best).astype(f32)).astype(f32) - hh).astype(f32),
                              (oz + (dz * best).astype(f32)).astype(f32)], axis=1)
                g = np.stack([w[:, 0], (-(k2 * w[:, 1]).astype(f32)).astype(f32), w[:, 2]], axis=1)
                gz = (g == 0).all(axis=1)
                up = np.broadcast_to(np.array([0.0, 1.0, 0.0], dtype=f32), (n, 3))
                down = np.broadcast_to(np.array([0.0, -1.0, 0.0], dtype=f32), (n, 3))
                nl = np.where((side == 1)[:, None], np.where(gz[:, None], up, _normalize3(g)), down).astype(f32)
                hit = inside | ok
                t = np.where(inside, ZERO, best).astype(f32)
                nrm = np.where(ok[:, None], quat_mul_vec3(np.broadcast_to(q, (n, 4)), nl), ZERO).astype(f32)
            else:  # box
                q = _a(c.rotation)
                qi = np.broadcast_to(np.array([-q[0], -q[1], -q[2], q[3]], dtype=f32), (n, 4))
                ol = quat_mul_vec3(qi, (origin - cpos).astype(f32))
                dl = quat_mul_vec3(qi, d)
                h = _a(c.half_extents)
                inside = (np.abs(ol) <= h).all(axis=1)
                tnear = np.full(n, -np.inf, dtype=f32)
                tfar = np.full(n, np.inf, dtype=f32)
                axis = np.zeros(n, dtype=np.int64)
                sign = np.zeros(n, dtype=f32)
                dead = np.zeros(n, dtype=bool)
                for i in range(3):
                    par = dl[:, i] == 0
                    dead |= ~dead & par & (np.abs(ol[:, i]) > h[i])
                    inv = (ONE / dl[:, i]).astype(f32)
                    t1 = ((-h[i] - ol[:, i]).astype(f32) * inv).astype(f32)
                    t2 = ((h[i] - ol[:, i]).astype(f32) * inv).astype(f32)
                    sw = t1 > t2
                    lo_, hi_ = np.where(sw, t2, t1), np.where(sw, t1, t2)
                    sg = np.where(sw, ONE, f32(-1.0)).astype(f32)
                    live = ~dead & ~par
                    upd = live & (lo_ > tnear)
                    tnear = np.where(upd, lo_, tnear).astype(f32)
                    axis = np.where(upd, i, axis)
                    sign = np.where(upd, sg, sign).astype(f32)
                    tfar = np.where(live & (hi_ < tfar), hi_, tfar).astype(f32)
                    dead |= live & (tnear > tfar)
                ok = ~inside & ~dead & (tnear >= 0) & (tnear <= max_distance)
                nl = np.zeros((n, 3), dtype=f32)
                nl[np.arange(n), axis] = sign
                hit = inside | ok
                t = np.where(inside, ZERO, tnear).astype(f32)
                nrm = np.where(ok[:, None], quat_mul_vec3(np.broadcast_to(q, (n, 4)), nl), ZERO).astype(f32)
            better = hit & (~found | (t < best_t))
            best_t = np.where(better, t, best_t).astype(f32)
            best_n = np.where(better[:, None], nrm, best_n).astype(f32)
            found |= hit
    return found, best_t, best_n


def particle_collision(pos, vel, delta, cs, colliders):
    """core.rs:744-800 for arrays of particles -> (pos, vel, should_destroy)"""
    n = len(pos)
    pos, vel = pos.astype(f32).copy(), vel.astype(f32).copy()
    orig = f32(delta)
    delta = np.full(n, orig, dtype=f32)
    steps = np.zeros(n, dtype=np.int64)
    destroy = np.zeros(n, dtype=bool)
    returned = np.zeros(n, dtype=bool)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for _ in range(4):
            act = ~returned & (delta > 0) & (steps < 4)
            if not act.any():
                break
            ln = _len3(vel)
            okdir = np.isfinite(ln) & (ln > 0)
            d = np.where(okdir[:, None], (vel / np.where(okdir, ln, ONE)[:, None]).astype(f32), _a((0.0, 1.0, 0.0))).astype(f32)
            found, dist, normal = cast_ray(colliders, cs.filter_mask, pos, d, (ln * delta).astype(f32))
            hit = act & found
            miss = act & ~found
            # ---- distance == 0 (core.rs:766-776)
            z = hit & (dist == 0)
            nz = (normal == 0).all(axis=1)
            vz = (vel == 0).all(axis=1)
            nfix = np.where((nz & ~vz)[:, None], _normalize3(vel), np.where((nz & vz)[:, None], _a((0.0, 1.0, 0.0)), normal)).astype(f32)
            k = np.fmax(ln, ONE).astype(f32)
            push = ((nfix * k[:, None]).astype(f32) * delta[:, None]).astype(f32)
            pos = np.where(z[:, None], (pos + push).astype(f32), pos).astype(f32)
            # ---- a real hit (core.rs:777-787)
            r = hit & ~z
            pos_r = (pos + (normalize_or_zero(vel) * dist[:, None]).astype(f32)).astype(f32)
            proj = _project_onto(vel, normal)
            rej = (vel - proj).astype(f32)
            fdv = (np.fmin(_len3(proj), _len3(rej)).astype(f32) * f32(cs.friction)).astype(f32)
            vel_r = ((rej - (normalize_or_zero(rej) * fdv[:, None]).astype(f32)).astype(f32)
                     - (proj * f32(cs.restitution)).astype(f32)).astype(f32)
            pos_r = (pos_r + (normal * f32(0.0001)).astype(f32)).astype(f32)
            nd = (delta - dist).astype(f32)
            nd = np.where(nd < 0, ZERO, nd).astype(f32)
            nd = np.where(nd > orig, orig, nd).astype(f32)
            pos = np.where(r[:, None], pos_r, pos).astype(f32)
            vel = np.where(r[:, None], vel_r, vel).astype(f32)
            delta = np.where(r, nd, delta).astype(f32)
            if cs.destroy_on_collision:
                destroy |= hit
                returned |= hit
            # ---- no hit
            pos = np.where(miss[:, None], (pos + (vel * delta[:, None]).astype(f32)).astype(f32), pos).astype(f32)
            delta = np.where(miss, ZERO, delta).astype(f32)
            steps = steps + act
    return pos, vel, destroy


# ---------------------------------------------------------------- the spawner
FIELDS = {"position": 3, "velocity": 3, "rotation": 4, "angular_velocity": 3, "initial_scale": 0, "scale": 0, "age": 0,
          "lifetime": 0, "base_color": 4, "emissive_color": 4}


def _empty(n_em):
    d = {k: np.zeros((0, w) if w else (0,), dtype=f32) for k, w in FIELDS.items()}
    d["last_emitted_age"] = np.zeros((0, n_em), dtype=f32)
    return d


def _cat(a, b):
    return {k: np.concatenate([a[k], b[k]], axis=0) for k in a}


def _take(a, mask):
    return {k: v[mask] for k, v in a.items()}


class Spawner:
    """ParticleSpawner + ParticleSpawnerData of one entity (core.rs:178-185, 269-303)."""

    def __init__(self, spawner, seed, uid, transform=None, modifier=None):
        self.s = spawner
        self.seed, self.uid = int(seed), int(uid)
        self.n_em = len(spawner.emission_settings)
        self.origin_t = _a(transform.translation if transform else (0, 0, 0))
        self.origin_r = _a(transform.rotation if transform else (0, 0, 0, 1))
        self.parent_velocity = np.zeros(3, dtype=f32)
        self.mod_scale = f32(modifier.scale if modifier else 1.0)
        self.mod_speed = f32(modifier.speed if modifier else 1.0)
        self.queued = 0
        self.colliders = []
        self.serial = [0] * self.n_em
        self.reset()

    def reset(self):  # sync_spawner_data core.rs:343-365
        self.particles = [_empty(self.n_em) for _ in self.s.particle_settings]
        self.destroyed = [_empty(self.n_em) for _ in self.s.particle_settings]
        self.last_emission = [ZERO] * self.n_em
        self.time_passed = [ZERO] * self.n_em
        self.enabled = [bool(self.s.starts_enabled)] * self.n_em
        self.nested = [e.emission_mode.kind == 1 for e in self.s.emission_settings]

    def count(self, t):
        return len(self.particles[t]["age"])

    def active(self):  # core.rs:288-302
        any_p = any(self.count(t) for t in range(len(self.particles)))
        return any((self.enabled[i] and any_p) if self.nested[i] else self.enabled[i] for i in range(self.n_em))

    # ---- one batch of new ParticleData (core.rs:437-469 Global, 506-544 Nested)
    def _make(self, ei, n, origin_pos, origin_rot, inherit_vel):
        e = self.s.emission_settings[ei]
        ps = self.s.particle_settings[e.particle_index]
        u = spawn_uniforms(self.seed, self.uid, ei, np.arange(self.serial[ei], self.serial[ei] + n, dtype=np.uint64))
        self.serial[ei] += n
        off = shape_points(e.emission_shape, u[:, 0:3])
        vr = randvec3(e.initial_velocity, u[:, 3], u[:, 4], u[:, 5])
        rv = quat_mul_vec3(np.broadcast_to(origin_rot, (n, 4)), vr)
        radial = randf32(e.initial_velocity_radial, u[:, 6])
        inner = (rv + (normalize_or_zero(off) * radial[:, None]).astype(f32)).astype(f32)
        inh = np.broadcast_to(inherit_vel, (n, 3)).astype(f32) if e.inherit_parent_velocity else np.zeros((n, 3), dtype=f32)
        vel = ((inner * self.mod_speed).astype(f32) + inh).astype(f32)
        iscale = (randf32(ps.initial_scale, u[:, 7]) * self.mod_scale).astype(f32)
        life = randf32(ps.lifetime, u[:, 8])
        w = randvec3(e.initial_angular_velocity, u[:, 9], u[:, 10], u[:, 11])
        zero_t = np.zeros(n, dtype=f32)
        return {
            "position": (np.broadcast_to(origin_pos, (n, 3)).astype(f32) + off).astype(f32), "velocity": vel,
            "rotation": np.broadcast_to(_a(e.initial_rotation), (n, 4)).astype(f32).copy(), "angular_velocity": w,
            "initial_scale": iscale, "scale": iscale.copy(), "age": zero_t, "lifetime": life,
            "base_color": gradient_sample(ps.base_color, zero_t), "emissive_color": gradient_sample(ps.emissive_color, zero_t),
            "last_emitted_age": np.full((n, self.n_em), F32_MIN, dtype=f32),
        }

    def spawn(self, dt):  # core.rs:367-551
        dt = f32(dt)
        if not self.active():
            return
        for i, e in enumerate(self.s.emission_settings):
            if not self.enabled[i]:
                continue
            pc = e.emission_pacing
            if e.emission_mode.kind == 0:
                if pc.kind == 0:
                    self.enabled[i] = False
                    n = pc.oneshot_count
                elif pc.kind == 1:
                    n, self.queued = self.queued, 0
                else:
                    self.time_passed[i] = f32(rem_euclid(f32(self.time_passed[i] + dt), pc.duration))
                    cnt, nxt = emission_count(self.time_passed[i], self.last_emission[i], pc.duration, pc.offset_start,
                                              pc.offset_end, pc.count)
                    n, self.last_emission[i] = int(cnt), f32(nxt)
                if n:
                    new = self._make(i, n, self.origin_t, self.origin_r, self.parent_velocity)
                    self.particles[e.particle_index] = _cat(self.particles[e.particle_index], new)
            else:
                if pc.kind != 2:
                    continue  # warn_once + continue (core.rs:474-485)
                par = self.particles[e.emission_mode.target_particle_type]  # the bound 0..len is fixed here (core.rs:488)
                cnt, nxt = emission_count(par["age"], par["last_emitted_age"][:, i], par["lifetime"], pc.offset_start,
                                          pc.offset_end, pc.count)
                par["last_emitted_age"][:, i] = nxt  # core.rs:500
                who = np.repeat(np.arange(len(cnt)), cnt.astype(np.int64))  # parent-major child order (core.rs:488-544)
                if len(who):
                    new = self._make(i, len(who), par["position"][who], par["rotation"][who], par["velocity"][who])
                    self.particles[e.particle_index] = _cat(self.particles[e.particle_index], new)

    def update(self, dt):  # core.rs:577-670 (non-avian arm)
        dt = f32(dt)
        for t, ps in enumerate(self.s.particle_settings):
            p = self.particles[t]
            age = (p["age"] + dt).astype(f32)
            dead = age >= p["lifetime"]
            gone = _take(p, dead)
            gone["age"] = age[dead]  # the clone already carries the advanced age (core.rs:592-599)
            self.destroyed[t] = gone
            q = _take(p, ~dead)
            q["age"] = age[~dead]
            pct = (q["age"] / q["lifetime"]).astype(f32)
            q["scale"] = (q["initial_scale"] * curve_sample(ps.scale_curve, pct)).astype(f32)
            if ps.collision_settings is not None:  # the physics_avian arm (core.rs:607-624, 633-639)
                npos, nvel, kill = particle_collision(q["position"], q["velocity"], dt, ps.collision_settings, self.colliders)
                q["position"], q["velocity"] = npos, nvel
                hit = _take(q, kill)  # destroyed with the NEW position / velocity / scale, old rotation and colours
                self.destroyed[t] = self._merge_destroyed(gone, hit, dead, kill)
                q = _take(q, ~kill)
                pct = pct[~kill]
                vel = q["velocity"]
            else:
                vel = q["velocity"]
                q["position"] = (q["position"] + (vel * dt).astype(f32)).astype(f32)
            acc = _a(ps.acceleration)
            q["velocity"] = (vel + ((acc - (vel * f32(ps.linear_drag)).astype(f32)).astype(f32) * dt).astype(f32)).astype(f32)
            w = q["angular_velocity"]
            q["rotation"] = quat_mul(quat_from_scaled_axis((w * dt).astype(f32)), q["rotation"])
            aa = _a(ps.angular_acceleration)
            q["angular_velocity"] = (w + ((aa - (f32(ps.angular_drag) * w).astype(f32)).astype(f32) * dt).astype(f32)).astype(f32)
            q["base_color"] = gradient_sample(ps.base_color, pct)
            q["emissive_color"] = gradient_sample(ps.emissive_color, pct)
            self.particles[t] = q

    @staticmethod
    def _merge_destroyed(gone, hit, dead, kill):
        """the reference pushes destroyed particles in iteration order: age deaths and collision deaths interleaved"""
        n = len(dead)
        order = np.zeros(n, dtype=np.int64)  # 1 = died of age, 2 = destroyed by a collision, 0 = alive
        order[dead] = 1
        alive_idx = np.flatnonzero(~dead)
        order[alive_idx[kill]] = 2
        out = {}
        for k in gone:
            width = gone[k].shape[1:]
            buf = np.zeros((n,) + width, dtype=f32)
            buf[order == 1] = gone[k]
            buf[order == 2] = hit[k]
            out[k] = buf[order != 0]
        return out

    def step(self, dt):  # plugin.rs:46-60: spawn_particles then update_particles
        self.spawn(dt)
        self.update(dt)
