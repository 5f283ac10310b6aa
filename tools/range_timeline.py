#!/usr/bin/env python3
"""Per-workgroup timeline of fw_k_update_range (FW_DEBUG=8): when the workgroups of each role start and end, how many run
at a time, how long the tail is.  FW_TL_EMITTERS=512x8192 (default: configs[4]'s share) / 256x65536.  Needs the instrumented
build of the library (`make -C bevy_firework_amd/csrc timeline`; the product build carries no timestamps -- they cost 6 %
of configs[4]'s share even unused), loaded through FW_LIB_PATH.  Run on the GPU box."""
import os as _os; _os.environ.setdefault("FW_ENABLE_KNOBS", "1")  # the A/B switches are honoured only with this set
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FW_DEBUG"] = os.environ.get("FW_DEBUG", "8")
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FW_LIB_PATH", os.path.join(_root, "bevy_firework_amd", "csrc", "libfirework_hip_timeline.so"))
assert os.path.exists(os.environ["FW_LIB_PATH"]), "build it first: make -C bevy_firework_amd/csrc timeline"
from bevy_firework_amd import workloads  # noqa: E402
from bevy_firework_amd.system import ParticleSystem  # noqa: E402

n_em, per = (int(v) for v in os.environ.get("FW_TL_EMITTERS", "512x8192").split("x"))
ps = ParticleSystem(seed=workloads.SEED)
for e, (sp, tf) in enumerate(workloads.many_emitters(n_em, per)):
    ps.spawn(sp, tf, uid=e)
dt = np.float32(1 / 60)
ps.update(dt)
for k in range(140):
    ps.step(dt)
ps.synchronize()
lib = ps._lib
n = C.c_uint64()
cap = 1 << 17
buf = np.zeros((cap, 8), dtype=np.uint64)
assert lib.fw_debug_read_range_timestamps(ps._ctx, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n)) == 0
t = buf[: n.value].astype(np.int64)
print("workgroups in the launch", len(t), "live", ps.live_count())
ran = t[:, 3] > 0
role = (t[:, 4] >> 30) & 3
t0 = t[ran, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 3] - t0) / 100.0  # s_memrealtime: 100 MHz -> us
print("span of the launch (first start -> last end of a wave 0): %.1f us" % en[ran].max())
for r, name in ((0, "OLD"), (1, "NEW"), (2, "YOUNG")):
    m = ran & (role == r)
    if not m.any():
        continue
    life = en[m] - st[m]
    print("%-5s %5d workgroups  lifetime us mean %.2f p50 %.2f p90 %.2f max %.2f   start p50 %.1f max %.1f   end p50 %.1f p99 %.1f max %.1f"
          % (name, m.sum(), life.mean(), np.median(life), np.percentile(life, 90), life.max(), np.median(st[m]), st[m].max(),
             np.median(en[m]), np.percentile(en[m], 99), en[m].max()))
m = ran & (role == 2) & (t[:, 1] >= t[:, 0]) & (t[:, 2] >= t[:, 1]) & (t[:, 6] >= t[:, 2]) & (t[:, 3] >= t[:, 6])
if m.any():
    print("YOUNG phases, mean us: descriptor + pinned record + segment record %.2f | type record + keys, barrier %.2f | first round arrives %.2f | rounds + stores %.2f"
          % (((t[m, 1] - t[m, 0]) / 100.0).mean(), ((t[m, 2] - t[m, 1]) / 100.0).mean(), ((t[m, 6] - t[m, 2]) / 100.0).mean(),
             ((t[m, 3] - t[m, 6]) / 100.0).mean()))
# how many workgroups are resident over time
grid = np.arange(0.0, en[ran].max() + 1.0, 2.0)
print("resident workgroups at t (us):", " ".join("%d:%d" % (g, ((st[ran] <= g) & (en[ran] > g)).sum()) for g in grid))
last = np.argsort(en * ran)[-8:]
for i in last:
    print("  late finisher: workgroup %d role %d k %d seg %d start %.1f end %.1f" % (i, role[i], t[i, 4] & 0x3FFFFFFF, t[i, 5], st[i], en[i]))
ps.close()
