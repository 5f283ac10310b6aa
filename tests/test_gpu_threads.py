"""Contexts driven from different threads at the same time (include/firework_hip.h: contexts share no state).  The reference
walks its spawners in parallel (update_particles: par_iter_mut, core.rs:583-585); a host of this backend spreads them over a
few contexts, one per worker thread (examples/many_contexts.cpp).  ctypes releases the GIL for the length of a call, so the
two threads below really are inside fw_step / the read-backs together."""
import threading

import numpy as np
import pytest

import oracle  # noqa: F401
from bevy_firework_amd import settings as S
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
from parity import Pair

pytestmark = pytest.mark.gpu
SEED = 20240607
DT = np.float32(1.0 / 60.0)


def _drive(k, frames, results, errors, barrier):
    try:
        with ParticleSystem(device=0, seed=SEED + k) as system:
            pairs = []
            # a few emitters of every update path: compacting, range ring, FIFO ring, a spawner with Nested entries
            for e, (sp, tf) in enumerate(workloads.many_emitters(6, 300 + 50 * k)):
                pairs.append(Pair(system, sp, tf, seed=SEED + k, uid=100 * k + e))
            sp, tf = workloads.stress_test(rate=20000.0 + 3000.0 * k)
            pairs.append(Pair(system, sp, tf, seed=SEED + k, uid=100 * k + 50))
            sp, tf = workloads.nested(spark_rate=400.0, smoke_per_spark=6.0)
            pairs.append(Pair(system, sp, tf, seed=SEED + k, uid=100 * k + 60))
            barrier.wait(timeout=120)
            rng = np.random.default_rng(k)
            for fr in range(frames):
                dt = np.float32(DT if fr % 3 else rng.uniform(0.008, 0.02))
                system.update(dt)
                for p in pairs:
                    p.step_cpu(dt)
                if fr % 20 == 19:
                    for i, p in enumerate(pairs):
                        p.check(what=f"thread {k} frame {fr} spawner {i}")
            results[k] = [p.gpu.counts() for p in pairs]
    except BaseException as e:  # noqa: BLE001 -- handed to the main thread
        errors[k] = e
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


def test_two_contexts_on_two_threads_match_the_oracle():
    n = 2
    results, errors = [None] * n, [None] * n
    barrier = threading.Barrier(n)
    threads = [threading.Thread(target=_drive, args=(k, 100, results, errors, barrier)) for k in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    for e in errors:
        if e is not None:
            raise e
    assert all(r is not None for r in results)
    assert all(sum(sum(c) for c in r) > 20000 for r in results), results


def test_create_errors_are_per_thread():
    """fw_last_error(NULL) after a failed fw_ctx_create: the calling thread's own message"""
    msgs = [None, None]

    def bad(k):
        try:
            ParticleSystem(device=4096 + k, seed=1)
        except Exception as e:  # noqa: BLE001
            msgs[k] = str(e)

    ts = [threading.Thread(target=bad, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert all(m and "device" in m for m in msgs), msgs
