"""Python binding of the CPU oracle (oracle/fw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Nothing in bevy_firework_amd/ imports this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

import numpy as np

from bevy_firework_amd import settings as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfw_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fw_oracle.c")
    hdr = os.path.join(_HERE, "fw_oracle.h")
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return LIB_PATH


class _RandF32(C.Structure):
    _fields_ = [("min", C.c_float), ("max", C.c_float)]


class _RandVec3(C.Structure):
    _fields_ = [("magnitude", _RandF32), ("direction", C.c_float * 3), ("spread", C.c_float)]


class _Curve(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("times", C.POINTER(C.c_float)), ("values", C.POINTER(C.c_float))]


class _Gradient(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("times", C.POINTER(C.c_float)), ("rgba", C.POINTER(C.c_float))]


class _ParticleSettings(C.Structure):
    _fields_ = [
        ("lifetime", _RandF32),
        ("scale_curve", _Curve),
        ("initial_scale", _RandF32),
        ("acceleration", C.c_float * 3),
        ("angular_acceleration", C.c_float * 3),
        ("linear_drag", C.c_float),
        ("angular_drag", C.c_float),
        ("base_color", _Gradient),
        ("emissive_color", _Gradient),
        ("pbr", C.c_int32),
        ("coll_enabled", C.c_int32),
        ("coll_restitution", C.c_float),
        ("coll_friction", C.c_float),
        ("coll_destroy_on_collision", C.c_int32),
        ("coll_filter_mask", C.c_uint32),
    ]


class _Collider(C.Structure):
    _fields_ = [("kind", C.c_int32), ("layers", C.c_uint32), ("position", C.c_float * 3), ("rotation", C.c_float * 4),
                ("normal", C.c_float * 3), ("radius", C.c_float), ("half_extents", C.c_float * 3)]


def make_colliders(colliders):
    arr = (_Collider * max(len(colliders), 1))()
    for d, c in zip(arr, colliders):
        d.kind, d.layers, d.radius = int(c.kind), int(c.layers) & 0xFFFFFFFF, float(c.radius)
        d.position[:] = [float(x) for x in c.position]
        d.rotation[:] = [float(x) for x in c.rotation]
        d.normal[:] = [float(x) for x in c.normal]
        d.half_extents[:] = [float(x) for x in c.half_extents]
    return arr


class _EmissionSettings(C.Structure):
    _fields_ = [
        ("particle_index", C.c_int32),
        ("pacing_kind", C.c_int32),
        ("oneshot_count", C.c_uint64),
        ("count", C.c_float),
        ("duration", C.c_float),
        ("offset_start", C.c_float),
        ("offset_end", C.c_float),
        ("mode", C.c_int32),
        ("target_particle_type", C.c_int32),
        ("shape_kind", C.c_int32),
        ("shape_radius", C.c_float),
        ("shape_normal", C.c_float * 3),
        ("initial_velocity", _RandVec3),
        ("initial_velocity_radial", _RandF32),
        ("inherit_parent_velocity", C.c_int32),
        ("initial_rotation", C.c_float * 4),
        ("initial_angular_velocity", _RandVec3),
    ]


_lib: Optional[C.CDLL] = None
_FP = C.POINTER(C.c_float)
_VP = C.c_void_p


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    L.fwo_compute_emission_count.restype = C.c_uint64
    L.fwo_compute_emission_count.argtypes = [C.c_float] * 6 + [_FP]
    L.fwo_rem_euclid.restype = C.c_float
    L.fwo_rem_euclid.argtypes = [C.c_float, C.c_float]
    L.fwo_div_euclid.restype = C.c_float
    L.fwo_div_euclid.argtypes = [C.c_float, C.c_float]
    L.fwo_uneven_normalize.restype = C.c_int32
    L.fwo_uneven_normalize.argtypes = [_FP, _FP, C.c_int32, C.c_int32]
    L.fwo_curve_sample_clamped.restype = C.c_float
    L.fwo_curve_sample_clamped.argtypes = [C.POINTER(_Curve), C.c_float]
    L.fwo_gradient_sample_clamped.restype = None
    L.fwo_gradient_sample_clamped.argtypes = [C.POINTER(_Gradient), C.c_float, _FP]
    L.fwo_philox4x32_10.restype = None
    L.fwo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.fwo_spawn_uniforms.restype = None
    L.fwo_spawn_uniforms.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, _FP]
    L.fwo_shape_generate.restype = None
    L.fwo_shape_generate.argtypes = [C.POINTER(_EmissionSettings), _FP, _FP]
    L.fwo_randvec3_generate.restype = None
    L.fwo_randvec3_generate.argtypes = [C.POINTER(_RandVec3), C.c_float, C.c_float, C.c_float, _FP]
    for name in ("fwo_quat_from_scaled_axis",):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [_FP, _FP]
    for name in ("fwo_quat_mul", "fwo_quat_mul_vec3", "fwo_quat_from_rotation_arc"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [_FP, _FP, _FP]
    L.fwo_particle_collision.restype = C.c_int32
    L.fwo_particle_collision.argtypes = [_FP, _FP, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_uint32,
                                         C.POINTER(_Collider), C.c_int32]
    L.fwo_spawner_set_colliders.restype = None
    L.fwo_spawner_set_colliders.argtypes = [_VP, C.POINTER(_Collider), C.c_int32]
    L.fwo_spawner_create.restype = _VP
    L.fwo_spawner_create.argtypes = [
        C.POINTER(_ParticleSettings), C.c_int32, C.POINTER(_EmissionSettings), C.c_int32, C.c_int32, C.c_uint32,
        C.c_uint32,
    ]
    L.fwo_spawner_destroy.restype = None
    L.fwo_spawner_destroy.argtypes = [_VP]
    L.fwo_spawner_reset.restype = None
    L.fwo_spawner_reset.argtypes = [_VP]
    L.fwo_spawner_set_origin.restype = None
    L.fwo_spawner_set_origin.argtypes = [_VP, _FP, _FP]
    L.fwo_spawner_set_parent_velocity.restype = None
    L.fwo_spawner_set_parent_velocity.argtypes = [_VP, _FP]
    L.fwo_spawner_set_modifier.restype = None
    L.fwo_spawner_set_modifier.argtypes = [_VP, C.c_float, C.c_float]
    L.fwo_spawner_queue.restype = None
    L.fwo_spawner_queue.argtypes = [_VP, C.c_uint64]
    L.fwo_spawner_active.restype = C.c_int32
    L.fwo_spawner_active.argtypes = [_VP]
    L.fwo_spawner_poll_finished.restype = C.c_int32
    L.fwo_spawner_poll_finished.argtypes = [_VP]
    for name in ("fwo_spawner_spawn", "fwo_spawner_update", "fwo_spawner_step"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [_VP, C.c_float]
    L.fwo_spawner_count.restype = C.c_uint64
    L.fwo_spawner_count.argtypes = [_VP, C.c_int32]
    L.fwo_spawner_read.restype = C.c_uint64
    L.fwo_spawner_read.argtypes = [_VP, C.c_int32, _VP, C.c_uint64]
    L.fwo_spawner_read_destroyed.restype = C.c_uint64
    L.fwo_spawner_read_destroyed.argtypes = [_VP, C.c_int32, _VP, C.c_uint64]
    L.fwo_spawner_read_last_emitted.restype = C.c_uint64
    L.fwo_spawner_read_last_emitted.argtypes = [_VP, C.c_int32, C.c_int32, _VP, C.c_uint64]
    L.fwo_spawner_write.restype = None
    L.fwo_spawner_write.argtypes = [_VP, C.c_int32, _VP, C.c_uint64]
    L.fwo_spawner_write_last_emitted.restype = None
    L.fwo_spawner_write_last_emitted.argtypes = [_VP, C.c_int32, C.c_int32, _VP, C.c_uint64]
    L.fwo_spawner_aabb.restype = C.c_int32
    L.fwo_spawner_aabb.argtypes = [_VP, _FP, _FP]
    _lib = L
    return L


def _farr(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def _fill_randf32(dst, r):
    dst.min, dst.max = float(r.min), float(r.max)


def _fill_randvec3(dst, r):
    _fill_randf32(dst.magnitude, r.magnitude)
    dst.direction[:] = [float(c) for c in r.direction]
    dst.spread = float(r.spread)


def make_curve(c: S.FireworkCurve, keep: List) -> _Curve:
    d = _Curve()
    d.kind, d.n = int(c.kind), len(c.values)
    v = _farr(c.values)
    keep.append(v)
    d.values = C.cast(v, _FP)
    if c.times:
        t = _farr(c.times)
        keep.append(t)
        d.times = C.cast(t, _FP)
        if d.kind == S.CURVE_UNEVEN:
            d.n = lib().fwo_uneven_normalize(d.times, d.values, d.n, 1)
    return d


def make_gradient(g: S.FireworkGradient, keep: List) -> _Gradient:
    d = _Gradient()
    d.kind, d.n = int(g.kind), len(g.colors)
    v = _farr([ch for col in g.colors for ch in col])
    keep.append(v)
    d.rgba = C.cast(v, _FP)
    if g.times:
        t = _farr(g.times)
        keep.append(t)
        d.times = C.cast(t, _FP)
        if d.kind == S.CURVE_UNEVEN:
            d.n = lib().fwo_uneven_normalize(d.times, d.rgba, d.n, 4)
    return d


def make_emission(e: S.EmissionSettings) -> _EmissionSettings:
    d = _EmissionSettings()
    d.particle_index = int(e.particle_index)
    p = e.emission_pacing
    d.pacing_kind, d.oneshot_count = int(p.kind), int(p.oneshot_count)
    d.count, d.duration = float(p.count), float(p.duration)
    d.offset_start, d.offset_end = float(p.offset_start), float(p.offset_end)
    d.mode = int(e.emission_mode.kind)
    d.target_particle_type = int(e.emission_mode.target_particle_type)
    d.shape_kind = int(e.emission_shape.kind)
    d.shape_radius = float(e.emission_shape.radius)
    d.shape_normal[:] = [float(c) for c in e.emission_shape.normal]
    _fill_randvec3(d.initial_velocity, e.initial_velocity)
    _fill_randf32(d.initial_velocity_radial, e.initial_velocity_radial)
    d.inherit_parent_velocity = 1 if e.inherit_parent_velocity else 0
    d.initial_rotation[:] = [float(c) for c in e.initial_rotation]
    _fill_randvec3(d.initial_angular_velocity, e.initial_angular_velocity)
    return d


# ---- unit-function wrappers -------------------------------------------------------

def compute_emission_count(t, last, dur, start, end, count):
    nxt = C.c_float()
    n = lib().fwo_compute_emission_count(t, last, dur, start, end, count, C.byref(nxt))
    return int(n), np.float32(nxt.value)


def curve_sample(c: S.FireworkCurve, t: float) -> np.float32:
    keep: List = []
    d = make_curve(c, keep)
    return np.float32(lib().fwo_curve_sample_clamped(C.byref(d), float(t)))


def gradient_sample(g: S.FireworkGradient, t: float) -> np.ndarray:
    keep: List = []
    d = make_gradient(g, keep)
    out = (C.c_float * 4)()
    lib().fwo_gradient_sample_clamped(C.byref(d), float(t), out)
    return np.array(out[:], dtype=np.float32)


def philox(ctr, key) -> np.ndarray:
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (C.c_uint32 * 4)()
    lib().fwo_philox4x32_10(c, k, o)
    return np.array(o[:], dtype=np.uint32)


def spawn_uniforms(seed, uid, emission_index, serial) -> np.ndarray:
    u = (C.c_float * 12)()
    lib().fwo_spawn_uniforms(seed, uid, emission_index, serial, u)
    return np.array(u[:], dtype=np.float32)


def _vec_call(fn, *arrs, n_out):
    out = (C.c_float * n_out)()
    fn(*[_farr(a) for a in arrs], out)
    return np.array(out[:], dtype=np.float32)


def quat_from_scaled_axis(v):
    return _vec_call(lib().fwo_quat_from_scaled_axis, v, n_out=4)


def quat_mul(a, b):
    return _vec_call(lib().fwo_quat_mul, a, b, n_out=4)


def quat_mul_vec3(q, v):
    return _vec_call(lib().fwo_quat_mul_vec3, q, v, n_out=3)


def quat_from_rotation_arc(a, b):
    return _vec_call(lib().fwo_quat_from_rotation_arc, a, b, n_out=4)


def shape_generate(e: S.EmissionSettings, u3) -> np.ndarray:
    d = make_emission(e)
    out = (C.c_float * 3)()
    lib().fwo_shape_generate(C.byref(d), _farr(u3), out)
    return np.array(out[:], dtype=np.float32)


def randvec3_generate(r: S.RandVec3, ua, ur, um) -> np.ndarray:
    d = _RandVec3()
    _fill_randvec3(d, r)
    out = (C.c_float * 3)()
    lib().fwo_randvec3_generate(C.byref(d), ua, ur, um, out)
    return np.array(out[:], dtype=np.float32)


def particle_collision(pos, vel, delta, settings: S.ParticleCollisionSettings, colliders):
    """core.rs:744-800 -> (pos, vel, should_destroy)"""
    p, v = _farr(pos), _farr(vel)
    d = lib().fwo_particle_collision(p, v, float(delta), float(settings.restitution), float(settings.friction),
                                     1 if settings.destroy_on_collision else 0, int(settings.filter_mask) & 0xFFFFFFFF,
                                     make_colliders(colliders), len(colliders))
    return np.array(p[:], dtype=np.float32), np.array(v[:], dtype=np.float32), bool(d)


# ---- spawner ------------------------------------------------------------------------

class OracleSpawner:
    """One ParticleSpawner + ParticleSpawnerData pair simulated by the C oracle."""

    def __init__(self, spawner: S.ParticleSpawner, seed: int = 0, uid: int = 0,
                 transform: Optional[S.Transform] = None):
        L = lib()
        keep: List = []
        n_ps, n_es = len(spawner.particle_settings), len(spawner.emission_settings)
        ps = (_ParticleSettings * max(n_ps, 1))()
        es = (_EmissionSettings * max(n_es, 1))()
        for i, p in enumerate(spawner.particle_settings):
            d = ps[i]
            _fill_randf32(d.lifetime, p.lifetime)
            d.scale_curve = make_curve(p.scale_curve, keep)
            _fill_randf32(d.initial_scale, p.initial_scale)
            d.acceleration[:] = [float(c) for c in p.acceleration]
            d.angular_acceleration[:] = [float(c) for c in p.angular_acceleration]
            d.linear_drag, d.angular_drag = float(p.linear_drag), float(p.angular_drag)
            d.base_color = make_gradient(p.base_color, keep)
            d.emissive_color = make_gradient(p.emissive_color, keep)
            d.pbr = 1 if p.pbr else 0
            cs = p.collision_settings
            d.coll_enabled = 1 if cs is not None else 0
            if cs is not None:
                d.coll_restitution, d.coll_friction = float(cs.restitution), float(cs.friction)
                d.coll_destroy_on_collision = 1 if cs.destroy_on_collision else 0
                d.coll_filter_mask = int(cs.filter_mask) & 0xFFFFFFFF
        for i, e in enumerate(spawner.emission_settings):
            es[i] = make_emission(e)
        self.n_types, self.n_emissions = n_ps, n_es
        self._h = L.fwo_spawner_create(ps, n_ps, es, n_es, 1 if spawner.starts_enabled else 0,
                                       int(seed) & 0xFFFFFFFF, int(uid) & 0xFFFFFFFF)
        if not self._h:
            raise ValueError("invalid spawner settings (reference would panic)")
        if transform is not None:
            self.set_origin(transform.translation, transform.rotation)

    def close(self):
        if self._h:
            lib().fwo_spawner_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_origin(self, translation, rotation=S.QUAT_IDENTITY):
        lib().fwo_spawner_set_origin(self._h, _farr(translation), _farr(rotation))

    def set_parent_velocity(self, v):
        lib().fwo_spawner_set_parent_velocity(self._h, _farr(v))

    def set_modifier(self, m: S.EffectModifier):
        lib().fwo_spawner_set_modifier(self._h, float(m.scale), float(m.speed))

    def set_colliders(self, colliders):
        lib().fwo_spawner_set_colliders(self._h, make_colliders(colliders), len(colliders))

    def queue_particles(self, n: int):
        lib().fwo_spawner_queue(self._h, int(n))

    def reset(self):
        lib().fwo_spawner_reset(self._h)

    def active(self) -> bool:
        return bool(lib().fwo_spawner_active(self._h))

    def poll_finished(self) -> bool:
        return bool(lib().fwo_spawner_poll_finished(self._h))

    def spawn(self, dt: float):
        lib().fwo_spawner_spawn(self._h, float(dt))

    def update(self, dt: float):
        lib().fwo_spawner_update(self._h, float(dt))

    def step(self, dt: float):
        lib().fwo_spawner_step(self._h, float(dt))

    def count(self, t: int = 0) -> int:
        return int(lib().fwo_spawner_count(self._h, t))

    def counts(self) -> List[int]:
        return [self.count(t) for t in range(self.n_types)]

    def particles(self, t: int = 0) -> np.ndarray:
        n = self.count(t)
        out = np.zeros(n, dtype=S.PARTICLE_DTYPE)
        lib().fwo_spawner_read(self._h, t, out.ctypes.data_as(_VP), n)
        return out

    def destroyed(self, t: int = 0) -> np.ndarray:
        n = int(lib().fwo_spawner_read_destroyed(self._h, t, None, 0))
        out = np.zeros(n, dtype=S.PARTICLE_DTYPE)
        lib().fwo_spawner_read_destroyed(self._h, t, out.ctypes.data_as(_VP), n)
        return out

    def last_emitted(self, t: int, emission_index: int) -> np.ndarray:
        n = self.count(t)
        out = np.zeros(n, dtype=np.float32)
        lib().fwo_spawner_read_last_emitted(self._h, t, emission_index, out.ctypes.data_as(_VP), n)
        return out

    def write_particles(self, t: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=S.PARTICLE_DTYPE)
        lib().fwo_spawner_write(self._h, t, arr.ctypes.data_as(_VP), len(arr))

    def write_last_emitted(self, t: int, emission_index: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        lib().fwo_spawner_write_last_emitted(self._h, t, emission_index, arr.ctypes.data_as(_VP), len(arr))

    def aabb(self):
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        any_ = lib().fwo_spawner_aabb(self._h, mn, mx)
        return bool(any_), np.array(mn[:], dtype=np.float32), np.array(mx[:], dtype=np.float32)
