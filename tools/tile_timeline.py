#!/usr/bin/env python3
"""Per-tile timeline of fw_k_update (FW_DEBUG=8): when workgroups start, how long each phase takes."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FW_DEBUG"] = os.environ.get("FW_DEBUG", "8")
from bevy_firework_amd import workloads  # noqa: E402
from bevy_firework_amd.system import ParticleSystem  # noqa: E402

ps = ParticleSystem(seed=workloads.SEED)
if os.environ.get("FW_TL_EMITTERS"):  # e.g. 2048x200: many small emitters instead of the one big one
    n_em, per = (int(v) for v in os.environ["FW_TL_EMITTERS"].split("x"))
    for e, (sp, tf) in enumerate(workloads.many_emitters(n_em, per)):
        ps.spawn(sp, tf, uid=e)
else:
    sp, tf = workloads.one_million()
    ps.spawn(sp, tf, uid=0)
dt = np.float32(1 / 60)
ps.update(dt)
jitter = os.environ.get("FW_TL_JITTER") == "1"  # a dt that changes every frame (the look-back / death-threshold schedule)
for k in range(100):
    ps.step(np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) if jitter else dt)
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
if (t[:, 7] > 0).any():
    ok = t[:, 7] > 0
    print("  of which round 0                  mean %.2f" % (rel[ok, 7] - rel[ok, 2]).mean())
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
cur = rel[:, 0] > (rel[:, 0].max() - 5.0)  # tiles of the last launch only
r2 = rel[cur] - rel[cur][:, 0].min()
idxs = np.where(cur)[0]
end = r2[:, 3]
print("last launch: tiles", cur.sum(), "end p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(end, [50, 90, 99, 100])))
slow = np.argsort(end)[-8:]
for j in slow:
    print("  slow tile", idxs[j], "start %.2f p1 %.2f p2 %.2f p3 %.2f (round0 %.2f) end %.2f" % (r2[j, 0], r2[j, 1] - r2[j, 0], r2[j, 2] - r2[j, 1], r2[j, 3] - r2[j, 2], r2[j, 7] - r2[j, 2] if t[idxs[j], 7] > 0 else -1, r2[j, 3]))
hist, edges = np.histogram(end, bins=12)
print("end-time histogram:", list(zip(np.round(edges[:-1], 1), hist)))
hw = t[cur][:, 6]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
print("mean end time per XCD:", [round(float(end[xcc == x].mean()), 2) if (xcc == x).any() else None for x in range(8)])
print("tiles per XCD:", [int((xcc == x).sum()) for x in range(8)])
key = (xcc * 1000 + se * 100 + sh * 16 + cu)
import collections
cnt = collections.Counter(key.tolist())
print("distinct CUs used:", len(cnt), "tiles per CU min/max:", min(cnt.values()), max(cnt.values()))
per_cu_end = {k: end[key == k].max() for k in cnt}
many = [k for k, v in cnt.items() if v == max(cnt.values())]
few = [k for k, v in cnt.items() if v == min(cnt.values())]
print("mean last-end on CUs with max tiles %.2f, with min tiles %.2f" % (np.mean([per_cu_end[k] for k in many]), np.mean([per_cu_end[k] for k in few])))
print("corr(tile index, end) = %.2f" % np.corrcoef(idxs, end)[0, 1])
