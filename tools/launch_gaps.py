#!/usr/bin/env python3
"""GPU-side gaps between consecutive update launches (FW_DEBUG=8): last workgroup end of launch i -> first workgroup
start of launch i+1, from the per-launch {min start, max end} ring the kernels keep.  Run on the GPU box."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import ctypes as C
import os
import sys
import time

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
for _ in range(200):
    ps.step(dt)
ps.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    ps.step(dt)
t1 = time.perf_counter()
ps.synchronize()
t2 = time.perf_counter()
print("host submit %.2f us/step, wall %.2f us/step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
lib = ps._lib
lib.fw_debug_read_timestamps2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
n = C.c_uint64()
cur = np.zeros((8192, 8), dtype=np.uint64)
prev = np.zeros((8192, 8), dtype=np.uint64)
assert lib.fw_debug_read_timestamps2(ps._ctx, cur.ctypes.data_as(C.c_void_p), prev.ctypes.data_as(C.c_void_p), 8192, C.byref(n)) == 0
c = cur[: n.value].astype(np.int64)
q = prev[: n.value].astype(np.int64)
c = c[c[:, 3] > 0]
q = q[q[:, 3] > 0]
c = c[c[:, 0] > c[:, 0].max() - 5000]  # tiles that ran in that launch (idle tiles keep stale stamps)
q = q[q[:, 0] > q[:, 0].max() - 5000]
us = lambda x: x / 100.0  # s_memrealtime ticks at 100 MHz
print("last launch : tiles %d span %.2f us" % (len(c), us(c[:, 3].max() - c[:, 0].min())))
print("one before  : tiles %d span %.2f us" % (len(q), us(q[:, 3].max() - q[:, 0].min())))
print("gap (last workgroup end of the one before -> first workgroup start of the last): %.2f us" % us(c[:, 0].min() - q[:, 3].max()))
print("period (first start -> first start): %.2f us" % us(c[:, 0].min() - q[:, 0].min()))

lib.fw_debug_read_launches.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
ring = np.zeros((256, 2, 64), dtype=np.uint64)
ep = C.c_uint32()
assert lib.fw_debug_read_launches(ps._ctx, ring.ctypes.data_as(C.c_void_p), C.byref(ep)) == 0
e = ep.value
idx = [(e - k) & 255 for k in range(100, -1, -1)]  # the last 101 launches, oldest first (slots half a lap ahead are being recycled)
start = (~ring[idx, 0, :].max(axis=1)).astype(np.int64)
end = ring[idx, 1, :].max(axis=1).astype(np.int64)
period = (start[1:] - start[:-1]) / 100.0
span = (end - start) / 100.0
gap = (start[1:] - end[:-1]) / 100.0
print("ring: period mean %.2f p50 %.2f p10 %.2f p90 %.2f" % (period.mean(), np.median(period), np.percentile(period, 10), np.percentile(period, 90)))
print("ring: kernel span (first start -> last end) mean %.2f p50 %.2f max %.2f" % (span.mean(), np.median(span), span.max()))
print("ring: gap (last end -> next first start)    mean %.2f p50 %.2f max %.2f" % (gap.mean(), np.median(gap), gap.max()))
print("spans:", np.round(span[:40], 1))
print("gaps :", np.round(gap[:40], 1))
