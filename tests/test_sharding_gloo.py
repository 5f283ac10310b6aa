"""The N>1 path on CPU: two gloo ranks, emitters sharded round-robin, live counts all-reduced.
The per-rank engine here is an oracle-backed stand-in with the ParticleSystem interface (the HIP
backend needs a GPU); what is under test is the partitioning + reduction logic of sharding.py and the
property that results do not depend on the number of ranks (RNG streams keyed by global index)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from bevy_firework_amd import sharding, workloads  # noqa: E402

DT = np.float32(1.0 / 60.0)
FRAMES = 43  # five full buckets of 8 frames + a partial one (flush)
N_EMITTERS = 6


class OracleSystem:
    """ParticleSystem look-alike over the CPU oracle (test double)."""

    def __init__(self, seed):
        self.seed, self.sp = seed, []

    def spawn(self, spawner, transform=None, uid=0, **kw):
        o = oracle.OracleSpawner(spawner, seed=self.seed, uid=uid, transform=transform)
        self.sp.append(o)
        return o

    def update(self, dt):
        for o in self.sp:
            o.step(dt)

    step = update

    def live_count(self):
        return sum(sum(o.counts()) for o in self.sp)


def emitters():
    return workloads.many_emitters(N_EMITTERS, live_per_emitter=300)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = sharding.ShardedParticleSystem(lambda: OracleSystem(workloads.SEED), emitters(), rank, world, reduce_every=8)
    for _ in range(FRAMES):
        sh.update(DT)
    sh.flush()
    digest = [(e, h.counts(), float(np.sum(h.particles(0)["position"], dtype=np.float64)))
              for e, h in zip(sh.global_indices, sh.handles)]
    out.put((rank, sh.global_live_history, sh.global_live_count(), digest))
    dist.destroy_process_group()


def test_round_robin_assignment():
    assert sharding.local_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((sharding.local_indices(4096, r, 8) for r in range(8)), [])) == list(range(4096))
    assert all(len(sharding.local_indices(4096, r, 8)) == 512 for r in range(8))


def test_two_rank_sharding_matches_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: all emitters in one system
    single = sharding.ShardedParticleSystem(lambda: OracleSystem(workloads.SEED), emitters(), 0, 1)
    hist = []
    for _ in range(FRAMES):
        single.update(DT)
        hist.append(single.local_live_count())
    want = {e: (h.counts(), float(np.sum(h.particles(0)["position"], dtype=np.float64)))
            for e, h in zip(single.global_indices, single.handles)}
    res.sort()
    (r0, h0, g0, d0), (r1, h1, g1, d1) = res
    assert h0 == h1 == hist  # every frame's all-reduced live count == unsharded count
    assert g0 == g1 == hist[-1]
    got = {e: (c, s) for e, c, s in d0 + d1}
    assert got == want  # each emitter evolves identically wherever it lives
    assert sorted(got) == list(range(N_EMITTERS))
