"""Multi-GPU partitioning of spawners.

Spawners are independent units: nothing in the reference's spawn/update path reads
another spawner's state (the reference itself relies on this for ``par_iter_mut``,
src/core.rs:583-585), and Nested emission stays inside one spawner.  So spawner
``e`` lives on rank ``e mod world`` and no particle ever crosses GPUs.  The only
exchange is the sum of live-particle counts:

  * every update kernel leaves its frame's total in a DEVICE ring registered with
    ``fw_ctx_live_count_ring`` (written by the kernel itself: no extra launch, no host
    synchronisation, no D2H copy in the frame);
  * every ``reduce_every`` frames one all-reduce carries the whole bucket of per-frame
    totals -- RCCL over xGMI with the ``nccl`` backend, on device memory, enqueued behind the
    frames that produced it.  The message is a few bytes and latency-bound, so it is sent
    rarely and -- round 6: ``async_op`` -- never sits between two update kernels: the collective
    runs on the backend's own stream behind the frames that wrote the bucket, the frames that
    follow do not wait for it (a few tens of microseconds of latency every ``reduce_every`` frames
    would be ~5 % of a 47 us frame at 8 ranks), and only a reader of the history waits;
  * results are only brought to the host when somebody asks (``global_live_history``).

With the ``gloo`` backend (CPU tests, or a GPU run without RCCL) the same bucket is
copied to the host first -- the one place a synchronisation is unavoidable there.

``make_system`` is a callable returning an object with the ``ParticleSystem``
interface (``spawn``, ``step``/``update``, ``live_count``); on a GPU it is
``bevy_firework_amd.system.ParticleSystem`` (which also has ``live_count_ring``).  A
system without ``live_count_ring`` (the oracle-backed double of the CPU tests) has its
ring slot filled from ``live_count()`` by this class instead; everything downstream --
bucket slicing, the collective, the history -- is the same code.  RNG streams are keyed
by the GLOBAL spawner index (``uid``), so results do not depend on the number of ranks.
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
                 rank: int = 0, world: int = 1, process_group=None, reduce_every: int = 16, torch_stream=None,
                 exchange: Optional[bool] = None):
        """``torch_stream``: the torch.cuda.Stream the system enqueues on (the bucket copy and the collective are
        ordered behind the frames on it).  ``exchange``: force the live-count exchange on/off (default: on when
        world > 1 or a process group is given)."""
        self.rank, self.world = rank, world
        self.pg = process_group
        self.reduce_every = max(1, int(reduce_every))
        self.system = make_system()
        self.global_indices = local_indices(len(spawners), rank, world)
        self.handles = [self.system.spawn(spawners[e][0], spawners[e][1], uid=e) for e in self.global_indices]
        self._frame = 0
        self._sent = 0                 # frames whose totals have been handed to a collective
        self._stream = torch_stream
        self._exchange = (world > 1 or process_group is not None) if exchange is None else bool(exchange)
        self._buckets: list = []       # reduced buckets (tensors, device or host), oldest first
        self._works: list = []         # ... and the handle of each one's collective (None: no collective was needed)
        self._history: List[int] = []  # buckets already brought to the host
        self._ring = None
        self._ring_n = 2 * self.reduce_every
        self._device_ring = False
        if self._exchange:
            import torch

            self._device_ring = hasattr(self.system, "live_count_ring") and torch.cuda.is_available()
            self._ring = torch.zeros(self._ring_n, dtype=torch.int64, device="cuda" if self._device_ring else "cpu")
            if self._device_ring:
                self.system.live_count_ring(self._ring.data_ptr(), self._ring_n)
                if self._stream is None:
                    # The update kernels write the ring on the SYSTEM's stream (a non-blocking stream of its own unless
                    # the caller gave it one): the bucket copy and the collective must be ordered behind them there, not
                    # on torch's current stream, which nothing orders against it.
                    self._stream = torch.cuda.ExternalStream(int(self.system.stream))

    # ---- frames ---------------------------------------------------------------------------------------------
    def update(self, dt: float) -> None:
        self.system.update(dt)
        self._after_frame()

    def step(self, dt: float) -> None:
        self.system.step(dt)
        self._after_frame()

    def _after_frame(self) -> None:
        k = self._frame
        self._frame += 1
        if not self._exchange:
            return
        if not self._device_ring:  # a system that cannot write the ring itself (CPU test double)
            self._ring[k % self._ring_n] = int(self.system.live_count())
        if self._frame - self._sent == self.reduce_every:
            self._reduce(self._sent, self.reduce_every)

    def _reduce(self, first_frame: int, n: int) -> None:
        """one collective for the per-frame totals of frames [first_frame, first_frame + n)"""
        import torch
        import torch.distributed as dist

        backend = dist.get_backend(self.pg) if (dist.is_available() and dist.is_initialized()) else None
        lo = first_frame % self._ring_n

        def go():
            if lo + n <= self._ring_n:
                bucket = self._ring[lo:lo + n].clone()  # stream-ordered behind the frames that wrote it
            else:  # a flush moved the bucket boundary: the window wraps around the ring
                bucket = torch.cat([self._ring[lo:], self._ring[: lo + n - self._ring_n]])
            if backend == "gloo" and bucket.is_cuda:
                bucket = bucket.cpu()  # gloo reduces host memory: the only synchronising fallback
            work = None
            # (a ONE-rank group runs the collective too -- a copy -- so that the tests' way through RCCL is the product's)
            if self.world > 1 or (self.pg is not None and backend is not None):
                work = dist.all_reduce(bucket, group=self.pg, async_op=True)  # RCCL over xGMI with the nccl backend: live counts only
            self._buckets.append(bucket)
            self._works.append(work)

        if self._device_ring and self._stream is not None:
            with torch.cuda.stream(self._stream):
                go()
        else:
            go()
        self._sent = first_frame + n

    def flush(self) -> None:
        """All-reduce the frames that have not been sent yet (a partial bucket)."""
        if self._exchange and self._frame > self._sent:
            self._reduce(self._sent, self._frame - self._sent)

    # ---- results --------------------------------------------------------------------------------------------
    @property
    def global_live_history(self) -> List[int]:
        """all-reduced live count of every frame reduced so far (synchronises: copies the buckets to the host)"""
        # The bucket copy and the collective were enqueued on the SYSTEM's stream (a non-blocking one); `tolist` copies on
        # torch's current stream, which nothing orders behind it: wait for the producer stream first.
        for w in self._works:  # (nccl: the CURRENT stream waits for the collective -- `tolist` below copies on it; gloo: the host waits)
            if w is not None:
                w.wait()
        self._works = []
        if self._buckets and self._device_ring and self._stream is not None:
            self._stream.synchronize()
        for b in self._buckets:
            self._history += [int(x) for x in b.tolist()]
        self._buckets = []
        return self._history

    def local_live_count(self) -> int:
        return self.system.live_count()

    def global_live_count(self) -> int:
        """Sum of live particles over all ranks, now (synchronises; one small collective)."""
        import torch
        import torch.distributed as dist

        dev = "cuda" if self._device_ring else "cpu"
        backend = dist.get_backend(self.pg) if (dist.is_available() and dist.is_initialized()) else None
        t = torch.tensor([self.system.live_count()], dtype=torch.int64, device="cpu" if backend == "gloo" else dev)
        if self.world > 1:
            dist.all_reduce(t, group=self.pg)
        return int(t.item())
