#!/usr/bin/env python3
"""Per-tile timeline of fw_k_update (FW_DEBUG=8): when workgroups start, how long each phase takes."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FW_DEBUG"] = os.environ.get("FW_DEBUG", "8")
from bevy_firework_amd import workloads  # noqa: E402
from bevy_firework_amd.system import ParticleSystem  # noqa: E402

ps = ParticleSystem(seed=workloads.SEED)
sp, tf = workloads.one_million()
ps.spawn(sp, tf, uid=0)
dt = np.float32(1 / 60)
ps.update(dt)
for _ in range(100):
    ps.step(dt)
ps.synchronize()
lib = ps._lib
lib.fw_debug_read_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
n = C.c_uint64()
buf = np.zeros((8192, 8), dtype=np.uint64)
assert lib.fw_debug_read_timestamps(ps._ctx, buf.ctypes.data_as(C.c_void_p), 8192, C.byref(n)) == 0
t = buf[: n.value].astype(np.int64)
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
rel = (t - t0) / 100.0  # s_memrealtime / readcyclecounter: 100 MHz -> us
print("tiles with data", len(t), "kernel span us", rel[:, 3].max())
life = rel[:, 3] - rel[:, 0]
print("lifetime us: mean %.2f p50 %.2f p90 %.2f max %.2f" % (life.mean(), np.median(life), np.percentile(life, 90), life.max()))
print("  prologue (entry -> loads issued)   mean %.2f" % (rel[:, 4] - rel[:, 0]).mean())
print("  Q0/Q3 arrive + parked in LDS      mean %.2f" % (rel[:, 5] - rel[:, 4]).mean())
print("  keys + forecast table             mean %.2f" % (rel[:, 6] - rel[:, 5]).mean())
print("  spawn/count/barrier               mean %.2f" % (rel[:, 1] - rel[:, 6]).mean())
print("phase1 (entry->after count barrier) mean %.2f" % (rel[:, 1] - rel[:, 0]).mean())
print("phase2 (prefix)                      mean %.2f" % (rel[:, 2] - rel[:, 1]).mean())
print("phase3 (integrate+store)             mean %.2f" % (rel[:, 3] - rel[:, 2]).mean())
starts = np.sort(rel[:, 0])
print("start times us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(starts, [10, 50, 90, 100])))
ends = np.sort(rel[:, 3])
print("end times us:   p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(ends, [10, 50, 90, 100])))
order = np.argsort(rel[:, 0])
for i in list(order[:3]) + list(order[len(order) // 2: len(order) // 2 + 3]) + list(order[-3:]):
    print("tile", i, "start %.2f  +p1 %.2f  +p2 %.2f  +p3 %.2f" % (rel[i, 0], rel[i, 1] - rel[i, 0], rel[i, 2] - rel[i, 1], rel[i, 3] - rel[i, 2]))
late = np.where(rel[:, 0] > 5.0)[0]
print("late starters:", len(late))
for i in late[:6]:
    print("  tile", i, "start %.2f  +p1 %.2f  +p2 %.2f  +p3 %.2f" % (rel[i, 0], rel[i, 1] - rel[i, 0], rel[i, 2] - rel[i, 1], rel[i, 3] - rel[i, 2]))
