"""ctypes binding of include/firework_hip.h (libfirework_hip.so).

There is no fallback: if the shared library is missing or does not export the
ABI, importing the backend raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C bevy_firework_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List

from . import settings as S

_HERE = os.path.dirname(os.path.abspath(__file__))
# FW_LIB_PATH lets A/B experiments load an alternative build of the same library
LIB_PATH = os.environ.get("FW_LIB_PATH") or os.path.join(_HERE, "csrc", "libfirework_hip.so")

FW_OK, FW_EINVAL, FW_ENOMEM, FW_EHIP, FW_ECAPACITY, FW_ENODEV, FW_ESMALL = 0, -1, -2, -3, -4, -5, -6
STATUS_NAMES = {
    0: "FW_OK", -1: "FW_EINVAL", -2: "FW_ENOMEM", -3: "FW_EHIP", -4: "FW_ECAPACITY", -5: "FW_ENODEV", -6: "FW_ESMALL",
}


class FwError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class RandF32(C.Structure):
    _fields_ = [("min", C.c_float), ("max", C.c_float)]


class RandVec3(C.Structure):
    _fields_ = [("magnitude", RandF32), ("direction", C.c_float * 3), ("spread", C.c_float)]


class Curve(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("times", C.POINTER(C.c_float)), ("values", C.POINTER(C.c_float))]


class Gradient(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("times", C.POINTER(C.c_float)), ("rgba", C.POINTER(C.c_float))]


class CollisionSettings(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("restitution", C.c_float), ("friction", C.c_float),
                ("destroy_on_collision", C.c_int32), ("filter_mask", C.c_uint32)]


class Collider(C.Structure):
    _fields_ = [("kind", C.c_int32), ("layers", C.c_uint32), ("position", C.c_float * 3), ("rotation", C.c_float * 4),
                ("normal", C.c_float * 3), ("radius", C.c_float), ("half_extents", C.c_float * 3)]


def make_colliders(colliders):
    arr = (Collider * max(len(colliders), 1))()
    for d, c in zip(arr, colliders):
        d.kind, d.layers, d.radius = int(c.kind), int(c.layers) & 0xFFFFFFFF, float(c.radius)
        d.position[:] = [float(x) for x in c.position]
        d.rotation[:] = [float(x) for x in c.rotation]
        d.normal[:] = [float(x) for x in c.normal]
        d.half_extents[:] = [float(x) for x in c.half_extents]
    return arr


class ParticleSettings(C.Structure):
    _fields_ = [
        ("lifetime", RandF32),
        ("scale_curve", Curve),
        ("initial_scale", RandF32),
        ("acceleration", C.c_float * 3),
        ("angular_acceleration", C.c_float * 3),
        ("linear_drag", C.c_float),
        ("angular_drag", C.c_float),
        ("base_color", Gradient),
        ("emissive_color", Gradient),
        ("pbr", C.c_int32),
        ("report_destroyed", C.c_int32),
        ("capacity", C.c_uint32),
        ("collision", CollisionSettings),
    ]


class EmissionSettings(C.Structure):
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
        ("initial_velocity", RandVec3),
        ("initial_velocity_radial", RandF32),
        ("inherit_parent_velocity", C.c_int32),
        ("initial_rotation", C.c_float * 4),
        ("initial_angular_velocity", RandVec3),
    ]


class SpawnerDesc(C.Structure):
    _fields_ = [
        ("particle_settings", C.POINTER(ParticleSettings)),
        ("n_particle_settings", C.c_uint32),
        ("emission_settings", C.POINTER(EmissionSettings)),
        ("n_emission_settings", C.c_uint32),
        ("starts_enabled", C.c_int32),
        ("uid", C.c_uint32),
    ]


def _farr(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def fill_randf32(dst, r: S.RandF32):
    dst.min, dst.max = float(r.min), float(r.max)


def fill_randvec3(dst, r: S.RandVec3):
    fill_randf32(dst.magnitude, r.magnitude)
    dst.direction[:] = [float(c) for c in r.direction]
    dst.spread = float(r.spread)


def fill_curve(dst, c: S.FireworkCurve, keep: List):
    dst.kind, dst.n = int(c.kind), len(c.values)
    v = _farr(c.values)
    keep.append(v)
    dst.values = C.cast(v, C.POINTER(C.c_float))
    if c.times:
        t = _farr(c.times)
        keep.append(t)
        dst.times = C.cast(t, C.POINTER(C.c_float))


def fill_gradient(dst, g: S.FireworkGradient, keep: List):
    dst.kind, dst.n = int(g.kind), len(g.colors)
    flat = [ch for col in g.colors for ch in col]
    v = _farr(flat)
    keep.append(v)
    dst.rgba = C.cast(v, C.POINTER(C.c_float))
    if g.times:
        t = _farr(g.times)
        keep.append(t)
        dst.times = C.cast(t, C.POINTER(C.c_float))


def fill_emission(dst, e: S.EmissionSettings):
    dst.particle_index = int(e.particle_index)
    p = e.emission_pacing
    dst.pacing_kind = int(p.kind)
    dst.oneshot_count = int(p.oneshot_count)
    dst.count, dst.duration = float(p.count), float(p.duration)
    dst.offset_start, dst.offset_end = float(p.offset_start), float(p.offset_end)
    dst.mode = int(e.emission_mode.kind)
    dst.target_particle_type = int(e.emission_mode.target_particle_type)
    dst.shape_kind = int(e.emission_shape.kind)
    dst.shape_radius = float(e.emission_shape.radius)
    dst.shape_normal[:] = [float(c) for c in e.emission_shape.normal]
    fill_randvec3(dst.initial_velocity, e.initial_velocity)
    fill_randf32(dst.initial_velocity_radial, e.initial_velocity_radial)
    dst.inherit_parent_velocity = 1 if e.inherit_parent_velocity else 0
    dst.initial_rotation[:] = [float(c) for c in e.initial_rotation]
    fill_randvec3(dst.initial_angular_velocity, e.initial_angular_velocity)


def make_desc(spawner: S.ParticleSpawner, uid: int):
    """Flatten a ParticleSpawner into a SpawnerDesc; returns (desc, keepalive)."""
    keep: List = []
    n_ps, n_es = len(spawner.particle_settings), len(spawner.emission_settings)
    ps = (ParticleSettings * max(n_ps, 1))()
    es = (EmissionSettings * max(n_es, 1))()
    for i, p in enumerate(spawner.particle_settings):
        d = ps[i]
        fill_randf32(d.lifetime, p.lifetime)
        fill_curve(d.scale_curve, p.scale_curve, keep)
        fill_randf32(d.initial_scale, p.initial_scale)
        d.acceleration[:] = [float(c) for c in p.acceleration]
        d.angular_acceleration[:] = [float(c) for c in p.angular_acceleration]
        d.linear_drag, d.angular_drag = float(p.linear_drag), float(p.angular_drag)
        fill_gradient(d.base_color, p.base_color, keep)
        fill_gradient(d.emissive_color, p.emissive_color, keep)
        d.pbr = 1 if p.pbr else 0
        d.report_destroyed = 1 if p.particles_destroyed is not None else 0
        d.capacity = int(p.capacity)
        cs = p.collision_settings
        d.collision.enabled = 1 if cs is not None else 0
        if cs is not None:
            d.collision.restitution, d.collision.friction = float(cs.restitution), float(cs.friction)
            d.collision.destroy_on_collision = 1 if cs.destroy_on_collision else 0
            d.collision.filter_mask = int(cs.filter_mask) & 0xFFFFFFFF
    for i, e in enumerate(spawner.emission_settings):
        fill_emission(es[i], e)
    desc = SpawnerDesc()
    desc.particle_settings = C.cast(ps, C.POINTER(ParticleSettings))
    desc.n_particle_settings = n_ps
    desc.emission_settings = C.cast(es, C.POINTER(EmissionSettings))
    desc.n_emission_settings = n_es
    desc.starts_enabled = 1 if spawner.starts_enabled else 0
    desc.uid = int(uid) & 0xFFFFFFFF
    keep += [ps, es]
    return desc, keep


# every symbol include/firework_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_F3 = C.POINTER(C.c_float)
SYMBOLS = [
    ("fw_abi_version", C.c_int, []),
    ("fw_ctx_create", C.c_int, [C.c_int, C.c_uint32, _P, C.POINTER(_P)]),
    ("fw_ctx_destroy", C.c_int, [_P]),
    ("fw_last_error", C.c_char_p, [_P]),
    ("fw_ctx_stream", _P, [_P]),
    ("fw_ctx_synchronize", C.c_int, [_P]),
    ("fw_ctx_set_colliders", C.c_int, [_P, C.POINTER(Collider), C.c_uint32]),
    ("fw_spawner_create", C.c_int, [_P, C.POINTER(SpawnerDesc), C.POINTER(C.c_int32)]),
    ("fw_spawner_update_settings", C.c_int, [_P, C.c_int32, C.POINTER(SpawnerDesc)]),
    ("fw_spawner_destroy", C.c_int, [_P, C.c_int32]),
    ("fw_spawner_set_origin", C.c_int, [_P, C.c_int32, _F3, _F3]),
    ("fw_ctx_set_origins", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int32), _F3, _F3]),
    ("fw_spawner_set_parent_velocity", C.c_int, [_P, C.c_int32, _F3]),
    ("fw_spawner_set_modifier", C.c_int, [_P, C.c_int32, C.c_float, C.c_float]),
    ("fw_spawner_queue", C.c_int, [_P, C.c_int32, C.c_uint64]),
    ("fw_ctx_set_parent_velocities", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int32), _F3]),
    ("fw_ctx_set_modifiers", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int32), _F3, _F3]),
    ("fw_ctx_queue", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    ("fw_step", C.c_int, [_P, C.c_float]),
    ("fw_spawner_counts", C.c_int, [_P, C.c_int32, C.POINTER(C.c_uint32), C.c_uint32]),
    ("fw_spawner_active", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    ("fw_spawner_poll_finished", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    ("fw_spawner_read_particles", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_spawner_read_last_emitted", C.c_int, [_P, C.c_int32, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_spawner_write_particles", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64]),
    ("fw_spawner_write_last_emitted", C.c_int, [_P, C.c_int32, C.c_uint32, C.c_uint32, _P, C.c_uint64]),
    ("fw_spawner_read_destroyed", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_spawner_pack_instances", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_spawner_pack_instances_device", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_spawner_attach_instances", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64]),
    ("fw_spawner_attach_instances_window", C.c_int, [_P, C.c_int32, C.c_uint32, _P, C.c_uint64]),
    ("fw_spawner_instance_window", C.c_int, [_P, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fw_spawner_aabb", C.c_int, [_P, C.c_int32, _F3, _F3, C.POINTER(C.c_int32)]),
    ("fw_ctx_track_aabbs", C.c_int, [_P, C.c_int32]),
    ("fw_ctx_live_count", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("fw_ctx_live_count_device", C.c_int, [_P, _P]),
    ("fw_ctx_live_count_ring", C.c_int, [_P, _P, C.c_uint32]),
    ("fw_ctx_last_step_updated", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("fw_ctx_kernel_timing", C.c_int, [_P, C.c_int32]),
    ("fw_ctx_kernel_timing_read", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fw_ctx_kernel_timing_overhead", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("fw_ctx_measure_copy_bandwidth", C.c_int, [_P, C.c_uint64, C.c_int32, C.POINTER(C.c_double)]),
    ("fw_debug_read_timestamps", C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_debug_read_timestamps2", C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_debug_read_launches", C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    ("fw_debug_read_range_timestamps", C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fw_debug_update_path", C.c_int, [_P, C.c_int, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("fw_debug_nest_frames", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fw_debug_param_bar", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("fw_debug_tile_scratch", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fw_debug_recovered_rings", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("fw_debug_tf_frames", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("fw_compute_emission_count", C.c_uint64,
     [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]),
]

# entry points added after round 4 (ABI 5 + a measurement hook): absent from the older builds the A/B tools load through FW_LIB_PATH
NEWER_THAN_R04 = {"fw_debug_nest_frames", "fw_debug_param_bar", "fw_debug_tile_scratch", "fw_debug_recovered_rings", "fw_debug_tf_frames", "fw_ctx_set_parent_velocities", "fw_ctx_set_modifiers", "fw_ctx_queue"}
_lib = None


def load() -> C.CDLL:
    """Load libfirework_hip.so and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not built: run `make -C bevy_firework_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback for the particle path."
        )
    lib = C.CDLL(LIB_PATH)
    # An OLDER build loaded for an A/B measurement (tools/, through FW_LIB_PATH) lacks what was added since and reports ABI 4: accepted
    # only on the explicit FW_ALLOW_OLD_ABI=1 (FW_LIB_PATH alone used to switch both checks off, so a stale or
    # mismatched library bound with ABI-5 argument types)
    allow_old = os.environ.get("FW_ALLOW_OLD_ABI") == "1" and bool(os.environ.get("FW_LIB_PATH"))
    for name, res, args in SYMBOLS:
        if allow_old and name in NEWER_THAN_R04 and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.fw_abi_version() != 5 and not (allow_old and lib.fw_abi_version() == 4):
        raise ImportError(f"{LIB_PATH}: ABI version {lib.fw_abi_version()}, this binding is for 5"
                          " (an older A/B build: FW_ALLOW_OLD_ABI=1 accepts version 4)")
    _lib = lib
    return lib
