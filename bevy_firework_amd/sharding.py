"""Multi-GPU partitioning of spawners.

Spawners are independent units: nothing in the reference's spawn/update path reads
another spawner's state (the reference itself relies on this for ``par_iter_mut``,
src/core.rs:583-585), and Nested emission stays inside one spawner.  So spawner
``e`` lives on rank ``e mod world`` and no particle ever crosses GPUs.  The only
exchange is the sum of live-particle counts, all-reduced over RCCL (``nccl`` backend
on ROCm) -- bucketed over several frames because the message is a few bytes and
latency-bound.

``make_system`` is a callable returning an object with the ``ParticleSystem``
interface (``spawn``, ``step``/``update``, ``live_count``); on a GPU it is
``bevy_firework_amd.system.ParticleSystem``.  RNG streams are keyed by the GLOBAL
spawner index (``uid``), so results do not depend on the number of ranks.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

from . import settings as S


def owner_rank(spawner_index: int, world: int) -> int:
    return spawner_index % world


def local_indices(n_spawners: int, rank: int, world: int) -> List[int]:
    return [e for e in range(n_spawners) if owner_rank(e, world) == rank]


class ShardedParticleSystem:
    def __init__(self, make_system: Callable[[], object], spawners: Sequence[Tuple[S.ParticleSpawner, S.Transform]],
                 rank: int = 0, world: int = 1, process_group=None, reduce_every: int = 1):
        self.rank, self.world = rank, world
        self.pg = process_group
        self.reduce_every = max(1, int(reduce_every))
        self.system = make_system()
        self.global_indices = local_indices(len(spawners), rank, world)
        self.handles = [self.system.spawn(spawners[e][0], spawners[e][1], uid=e) for e in self.global_indices]
        self._frame = 0
        self._pending: List[int] = []
        self.global_live_history: List[int] = []

    def update(self, dt: float) -> None:
        self.system.update(dt)
        self._after_frame()

    def step(self, dt: float) -> None:
        self.system.step(dt)
        self._after_frame()

    def _after_frame(self) -> None:
        self._frame += 1
        if self.world > 1 or self.pg is not None:
            self._pending.append(self.system.live_count())
            if len(self._pending) == self.reduce_every:
                self.flush()

    def flush(self) -> None:
        """All-reduce the buffered per-frame live counts (one collective for the whole bucket)."""
        if not self._pending:
            return
        import torch
        import torch.distributed as dist

        t = torch.tensor(self._pending, dtype=torch.int64)
        if self.world > 1:
            dist.all_reduce(t, group=self.pg)
        self.global_live_history += [int(x) for x in t.tolist()]
        self._pending = []

    def local_live_count(self) -> int:
        return self.system.live_count()

    def global_live_count(self) -> int:
        """Sum of live particles over all ranks, now (one small collective)."""
        import torch
        import torch.distributed as dist

        t = torch.tensor([self.system.live_count()], dtype=torch.int64)
        if self.world > 1:
            dist.all_reduce(t, group=self.pg)
        return int(t.item())
