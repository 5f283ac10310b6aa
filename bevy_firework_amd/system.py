"""Host-side mirror of the reference's plugin surface over the HIP backend.

``ParticleSystem`` plays the role of ``ParticleSystemPlugin`` (reference
src/plugin.rs:22-61): it owns one GPU context and ``update(dt)`` runs the
chained per-frame systems -- sync_spawner_data, spawn_particles,
update_particles, notify_finished_particle_spawners -- for every spawner, on
the device, through the C ABI of include/firework_hip.h.

``SpawnerData`` mirrors ``ParticleSpawnerData`` (src/core.rs:269-303):
``queue_particles``, ``active`` and read access to ``particles``.

No simulation arithmetic happens in Python and there is no CPU fallback: the
shared library must be built (hipcc, gfx950) and a GPU must be present.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import numpy as np

from . import _ffi
from . import settings as S
from ._ffi import FwError


class SpawnerData:
    """Runtime state handle of one spawner (ParticleSpawnerData, core.rs:269-303)."""

    def __init__(self, system: "ParticleSystem", handle: int, spawner: S.ParticleSpawner, uid: int):
        self._sys = system
        self.handle = handle
        self.settings = spawner
        self.uid = uid
        self.transform = S.Transform()
        self.global_transform: Optional[S.Transform] = None
        self.on_finished: List[Callable[["SpawnerData"], None]] = []

    # -- inputs ---------------------------------------------------------------------
    def queue_particles(self, count: int) -> None:  # core.rs:284-286
        self._sys._check(self._sys._lib.fw_spawner_queue(self._sys._ctx, self.handle, int(count)))

    def set_transform(self, transform: S.Transform, global_transform: Optional[S.Transform] = None) -> None:
        self.transform = transform
        self.global_transform = global_transform

    def set_parent_velocity(self, v) -> None:  # core.rs:276
        arr = (C.c_float * 3)(*[float(c) for c in v])
        self._sys._check(self._sys._lib.fw_spawner_set_parent_velocity(self._sys._ctx, self.handle, arr))

    def set_modifier(self, m: S.EffectModifier) -> None:  # core.rs:323-327
        self._sys._check(self._sys._lib.fw_spawner_set_modifier(self._sys._ctx, self.handle, float(m.scale), float(m.speed)))

    def update_settings(self, spawner: S.ParticleSpawner) -> None:
        """Changed<ParticleSpawner>: sync_spawner_data resets state and drops particles (core.rs:343-365)."""
        desc, keep = _ffi.make_desc(spawner, self.uid)
        self._sys._check(self._sys._lib.fw_spawner_update_settings(self._sys._ctx, self.handle, C.byref(desc)))
        self.settings = spawner

    # -- outputs (synchronise) -----------------------------------------------------------
    def counts(self) -> List[int]:
        n = len(self.settings.particle_settings)
        out = (C.c_uint32 * max(n, 1))()
        self._sys._check(self._sys._lib.fw_spawner_counts(self._sys._ctx, self.handle, out, n))
        return [int(out[i]) for i in range(n)]

    def count(self, particle_type: int = 0) -> int:
        return self.counts()[particle_type]

    def active(self) -> bool:  # core.rs:288-302
        out = C.c_int32()
        self._sys._check(self._sys._lib.fw_spawner_active(self._sys._ctx, self.handle, C.byref(out)))
        return bool(out.value)

    def poll_finished(self) -> bool:  # core.rs:674-688
        out = C.c_int32()
        self._sys._check(self._sys._lib.fw_spawner_poll_finished(self._sys._ctx, self.handle, C.byref(out)))
        return bool(out.value)

    def particles(self, particle_type: int = 0) -> np.ndarray:
        """``data.particles[particle_type]`` as a structured array (settings.PARTICLE_DTYPE)."""
        n = C.c_uint64()
        L, ctx = self._sys._lib, self._sys._ctx
        self._sys._check(L.fw_spawner_read_particles(ctx, self.handle, particle_type, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=S.PARTICLE_DTYPE)
        if n.value:
            self._sys._check(L.fw_spawner_read_particles(ctx, self.handle, particle_type, out.ctypes.data_as(C.c_void_p),
                                                         n.value, C.byref(n)))
        return out

    def last_emitted(self, particle_type: int, emission_index: int) -> np.ndarray:
        n = C.c_uint64()
        L, ctx = self._sys._lib, self._sys._ctx
        self._sys._check(L.fw_spawner_read_last_emitted(ctx, self.handle, particle_type, emission_index, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.float32)
        if n.value:
            self._sys._check(L.fw_spawner_read_last_emitted(ctx, self.handle, particle_type, emission_index,
                                                            out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def write_particles(self, particle_type: int, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr, dtype=S.PARTICLE_DTYPE)
        self._sys._check(self._sys._lib.fw_spawner_write_particles(self._sys._ctx, self.handle, particle_type,
                                                                   arr.ctypes.data_as(C.c_void_p), len(arr)))

    def write_last_emitted(self, particle_type: int, emission_index: int, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self._sys._check(self._sys._lib.fw_spawner_write_last_emitted(self._sys._ctx, self.handle, particle_type,
                                                                      emission_index, arr.ctypes.data_as(C.c_void_p), len(arr)))

    def destroyed(self, particle_type: int = 0) -> np.ndarray:
        n = C.c_uint64()
        L, ctx = self._sys._lib, self._sys._ctx
        self._sys._check(L.fw_spawner_read_destroyed(ctx, self.handle, particle_type, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=S.PARTICLE_DTYPE)
        if n.value:
            self._sys._check(L.fw_spawner_read_destroyed(ctx, self.handle, particle_type, out.ctypes.data_as(C.c_void_p),
                                                         n.value, C.byref(n)))
        return out

    def instances(self, particle_type: int = 0) -> np.ndarray:
        """ParticleInstance records for the render extract (render.rs:95-115)."""
        n = C.c_uint64()
        L, ctx = self._sys._lib, self._sys._ctx
        self._sys._check(L.fw_spawner_pack_instances(ctx, self.handle, particle_type, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=S.INSTANCE_DTYPE)
        if n.value:
            self._sys._check(L.fw_spawner_pack_instances(ctx, self.handle, particle_type, out.ctypes.data_as(C.c_void_p),
                                                         n.value, C.byref(n)))
        return out

    def attach_instances(self, device_ptr: int, capacity: int, particle_type: int = 0) -> None:
        """From the next step on the update kernel also writes this type's ParticleInstance records (render.rs:95-115)
        into the caller's device buffer (`capacity` 64-byte records); 0 detaches."""
        self._sys._check(self._sys._lib.fw_spawner_attach_instances(
            self._sys._ctx, self.handle, particle_type, C.c_void_p(device_ptr) if device_ptr else None, int(capacity)))

    def attach_instances_window(self, device_ptr: int, capacity: int, particle_type: int = 0) -> None:
        """The same for a host that can draw an instance sub-range: the records of a step's survivors are
        buffer[first : first + count] with (first, count) = ``instance_window()``; lets a lifetime-range type keep its ring."""
        self._sys._check(self._sys._lib.fw_spawner_attach_instances_window(
            self._sys._ctx, self.handle, particle_type, C.c_void_p(device_ptr) if device_ptr else None, int(capacity)))

    def instance_window(self, particle_type: int = 0):
        """(first, count) of the attached windowed buffer's live records after the last step (synchronises)"""
        first, count = C.c_uint64(), C.c_uint64()
        self._sys._check(self._sys._lib.fw_spawner_instance_window(self._sys._ctx, self.handle, particle_type, C.byref(first), C.byref(count)))
        return int(first.value), int(count.value)

    def update_path(self, particle_type: int = 0):
        """("fifo" | "range" | "general" | "small" (one wave per type, the compacting layout), bytes one update of a live particle moves, of which algorithmic) -- which kernel family
        updates this type and what it costs per particle (bench.py's roofline accounting)"""
        mode, moved, algo = C.c_int32(), C.c_uint32(), C.c_uint32()
        self._sys._check(self._sys._lib.fw_debug_update_path(self._sys._ctx, self.handle, particle_type, C.byref(mode),
                                                             C.byref(moved), C.byref(algo)))
        # (4: a WIDE small type -- a workgroup of the same kernel instead of a wave; update_mode() tells the two apart)
        return {0: "general", 1: "fifo", 2: "range", 3: "small", 4: "small"}[mode.value], int(moved.value), int(algo.value)

    def update_mode(self, particle_type: int = 0) -> int:
        """the raw mode of fw_debug_update_path: 0 compacting, 1 FIFO ring, 2 range ring, 3 small (a wave), 4 small (a workgroup)"""
        mode = C.c_int32()
        self._sys._check(self._sys._lib.fw_debug_update_path(self._sys._ctx, self.handle, particle_type, C.byref(mode), None, None))
        return int(mode.value)

    def aabb(self):
        """(any, min, max) of position -/+ scale over all particle types (render.rs:677-703)."""
        mn, mx, any_ = (C.c_float * 3)(), (C.c_float * 3)(), C.c_int32()
        self._sys._check(self._sys._lib.fw_spawner_aabb(self._sys._ctx, self.handle, mn, mx, C.byref(any_)))
        return bool(any_.value), np.array(mn[:], dtype=np.float32), np.array(mx[:], dtype=np.float32)


class ParticleSystem:
    """One GPU context + its spawners; ``update(dt)`` is one frame of the plugin's system chain."""

    def __init__(self, device: int = 0, seed: int = 0, stream: Optional[int] = None):
        self._lib = _ffi.load()
        ctx = C.c_void_p()
        st = self._lib.fw_ctx_create(int(device), int(seed) & 0xFFFFFFFF, C.c_void_p(stream) if stream else None, C.byref(ctx))
        if st != _ffi.FW_OK:
            msg = self._lib.fw_last_error(None)
            raise FwError(st, msg.decode() if msg else "fw_ctx_create failed")
        self._ctx = ctx
        self.device = device
        self.seed = seed
        self.spawners: Dict[int, SpawnerData] = {}
        self._next_uid = 0
        self._keep: List = []

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, status: int) -> None:
        if status != _ffi.FW_OK:
            msg = self._lib.fw_last_error(self._ctx)
            raise FwError(status, msg.decode() if msg else "")

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.fw_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def stream(self) -> int:
        return int(self._lib.fw_ctx_stream(self._ctx) or 0)

    def synchronize(self) -> None:
        self._check(self._lib.fw_ctx_synchronize(self._ctx))

    def set_colliders(self, colliders) -> None:
        """The world particle_collision casts its rays into (core.rs:744-800): a device-resident set of analytic
        colliders (settings.Collider) standing in for avian's SpatialQuery."""
        self._check(self._lib.fw_ctx_set_colliders(self._ctx, _ffi.make_colliders(colliders), len(colliders)))

    # -- ECS-like surface ------------------------------------------------------------------------
    def spawn(self, spawner: S.ParticleSpawner, transform: Optional[S.Transform] = None,
              global_transform: Optional[S.Transform] = None, modifier: Optional[S.EffectModifier] = None,
              uid: Optional[int] = None) -> SpawnerData:
        """commands.spawn((ParticleSpawner {..}, Transform)): returns the ParticleSpawnerData handle."""
        if uid is None:
            uid = self._next_uid
        self._next_uid = max(self._next_uid, uid + 1)
        desc, keep = _ffi.make_desc(spawner, uid)
        h = C.c_int32(-1)
        self._check(self._lib.fw_spawner_create(self._ctx, C.byref(desc), C.byref(h)))
        data = SpawnerData(self, h.value, spawner, uid)
        if transform is not None:
            data.set_transform(transform, global_transform)
        if modifier is not None:
            data.set_modifier(modifier)
        self.spawners[h.value] = data
        return data

    def despawn(self, data: SpawnerData) -> None:
        self._check(self._lib.fw_spawner_destroy(self._ctx, data.handle))
        self.spawners.pop(data.handle, None)

    def _push_origins(self) -> None:
        """the transforms of ALL spawners in one call (fw_ctx_set_origins): spawn_particles reads them for every spawner
        entity of the query in one system (core.rs:377, 432-435)"""
        n = len(self.spawners)
        if not n:
            return
        handles = (C.c_int32 * n)()
        tr = (C.c_float * (3 * n))()
        ro = (C.c_float * (4 * n))()
        for i, d in enumerate(self.spawners.values()):
            # SpawnTransformMode (core.rs:432-435): Global -> GlobalTransform.compute_transform(), Local -> Transform
            t = d.transform
            if d.settings.spawn_transform_mode == S.SpawnTransformMode.Global and d.global_transform is not None:
                t = d.global_transform
            handles[i] = d.handle
            tr[3 * i:3 * i + 3] = [float(c) for c in t.translation]
            ro[4 * i:4 * i + 4] = [float(c) for c in t.rotation]
        self._check(self._lib.fw_ctx_set_origins(self._ctx, n, handles, tr, ro))

    # the other per-frame inputs of MANY spawners in one FFI call each (ABI 5): what sync_parent_velocity (core.rs:706-736),
    # propagate_particle_spawner_modifier (core.rs:690-703) and a gameplay system that queues on its OnDemand spawners write
    def set_parent_velocities(self, spawners, velocities) -> None:
        n = len(spawners)
        handles = (C.c_int32 * max(n, 1))(*[d.handle for d in spawners])
        v = (C.c_float * max(3 * n, 1))(*[float(c) for vel in velocities for c in vel])
        self._check(self._lib.fw_ctx_set_parent_velocities(self._ctx, n, handles, v))

    def set_modifiers(self, spawners, modifiers) -> None:
        n = len(spawners)
        handles = (C.c_int32 * max(n, 1))(*[d.handle for d in spawners])
        sc = (C.c_float * max(n, 1))(*[float(m.scale) for m in modifiers])
        sp = (C.c_float * max(n, 1))(*[float(m.speed) for m in modifiers])
        self._check(self._lib.fw_ctx_set_modifiers(self._ctx, n, handles, sc, sp))

    def queue_particles(self, spawners, counts) -> None:
        n = len(spawners)
        handles = (C.c_int32 * max(n, 1))(*[d.handle for d in spawners])
        cn = (C.c_uint64 * max(n, 1))(*[int(c) for c in counts])
        self._check(self._lib.fw_ctx_queue(self._ctx, n, handles, cn))

    def step(self, dt: float) -> None:
        """Enqueue one frame (spawn_particles + update_particles) without touching transforms or callbacks."""
        self._check(self._lib.fw_step(self._ctx, float(dt)))

    def update(self, dt: float) -> None:
        """One run of the plugin's chained systems (plugin.rs:46-60)."""
        self._push_origins()
        self.step(dt)
        for d in self.spawners.values():
            handlers = [(i, p.particles_destroyed) for i, p in enumerate(d.settings.particle_settings)
                        if p.particles_destroyed is not None]
            for i, fn in handlers:  # commands.run_system_with(handler, destroyed) (core.rs:660-667)
                dead = d.destroyed(i)
                if len(dead):
                    fn(dead)
            if d.on_finished and d.poll_finished():  # ParticleSpawnerFinished observers (core.rs:674-688)
                for fn in d.on_finished:
                    fn(d)

    def track_aabbs(self, enable: bool = True) -> None:
        """AABB fused into the update (render.rs:677-703): every tile leaves the box of its survivors; `aabb()` folds
        those instead of re-reading the particles."""
        self._check(self._lib.fw_ctx_track_aabbs(self._ctx, 1 if enable else 0))

    # -- statistics / measurement ----------------------------------------------------------------
    def live_count(self) -> int:
        out = C.c_uint64()
        self._check(self._lib.fw_ctx_live_count(self._ctx, C.byref(out)))
        return int(out.value)

    def live_count_device(self, device_ptr: int) -> None:
        self._check(self._lib.fw_ctx_live_count_device(self._ctx, C.c_void_p(device_ptr)))

    def live_count_ring(self, device_ptr: int, n_slots: int) -> None:
        """Every later step leaves its frame's total live count in ring[k % n_slots] (device uint64), no extra launch."""
        self._check(self._lib.fw_ctx_live_count_ring(self._ctx, C.c_void_p(device_ptr) if device_ptr else None, int(n_slots)))

    def updated_total(self) -> int:
        out = C.c_uint64()
        self._check(self._lib.fw_ctx_last_step_updated(self._ctx, C.byref(out)))
        return int(out.value)

    def kernel_timing(self, enable: bool) -> None:
        self._check(self._lib.fw_ctx_kernel_timing(self._ctx, 1 if enable else 0))

    def kernel_timing_read(self):
        ms, n, parts = C.c_double(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.fw_ctx_kernel_timing_read(self._ctx, C.byref(ms), C.byref(n), C.byref(parts)))
        return ms.value, int(n.value), int(parts.value)

    def kernel_timing_overhead_us(self) -> float:
        out = C.c_double()
        self._check(self._lib.fw_ctx_kernel_timing_overhead(self._ctx, C.byref(out)))
        return out.value * 1e3

    def param_bar(self) -> bool:
        """per-frame records and small op tables live in device memory the host writes through the large BAR (else: pinned host memory)"""
        on = C.c_int32()
        if not hasattr(self._lib, "fw_debug_param_bar"):
            return False
        self._check(self._lib.fw_debug_param_bar(self._ctx, C.byref(on)))
        return bool(on.value)

    def tile_scratch(self):
        """(tiles of the compacting launch's table, entries every per-tile array of the context holds)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.fw_debug_tile_scratch(self._ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def tf_frames(self) -> int:
        """frames whose dt differed from the previous one's and that ran fw_k_fc_resolve + the streaming schedule (threshold forecast)"""
        n = C.c_uint64()
        self._check(self._lib.fw_debug_tf_frames(self._ctx, C.byref(n)))
        return int(n.value)

    def recovered_rings(self) -> int:
        """rings moved to the compacting path (particles kept) because a cohort report was missing when it was due"""
        n = C.c_uint64()
        self._check(self._lib.fw_debug_recovered_rings(self._ctx, C.byref(n)))
        return int(n.value)

    def nest_frames(self):
        """(frames whose Nested entries ran inside the FIFO ring launch, frames that ran the separate fw_k_spawn / fw_k_nest passes)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.fw_debug_nest_frames(self._ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def measure_copy_bandwidth(self, nbytes: int = 1 << 30, iters: int = 20) -> float:
        out = C.c_double()
        self._check(self._lib.fw_ctx_measure_copy_bandwidth(self._ctx, int(nbytes), int(iters), C.byref(out)))
        return out.value


def compute_emission_count(t, last, duration, start, end, count):
    """compute_emission_count (core.rs:553-575) as the library's host side evaluates it."""
    nxt = C.c_float()
    n = _ffi.load().fw_compute_emission_count(t, last, duration, start, end, count, C.byref(nxt))
    return int(n), np.float32(nxt.value)
