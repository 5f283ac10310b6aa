"""10 000 frames of ONE context at product defaults (no knob set) in which spawners come and go the whole time: the segment count
crosses fw_ctx::range_few (192) in both directions again and again, now and then fw_ctx::small_min (352 small types: the wave-per-type
kernel takes them over, and hands them back below 264), FIFO rings arrive and leave, a ninth large one-lifetime type
arrives (the FIFO rings become range rings where they stand) and the converted rings drain away, a Nested spawner's entry runs inside
its FIFO launch; small types are updated by one wave each.  The state of a random subset of the spawners against the ORACLE every
250 frames and of all of them at the end; any disagreement between the host's bookkeeping and the particles raises an internal error
inside an update kernel and surfaces as FW_EHIP at the next call.   python tools/soak_r05.py   (GPU box; FW_SOAK_FRAMES=n)"""
import os, sys, time
for k in list(os.environ):
    if k.startswith("FW_") and k not in ("FW_SOAK_FRAMES", "FW_LIB_PATH"):
        del os.environ[k]
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from bevy_firework_amd.system import ParticleSystem
import test_gpu_lifecycle as L

frames = int(os.environ.get("FW_SOAK_FRAMES", "10000"))
rng = np.random.default_rng(2025)
events = {"created": 0, "despawned": 0, "rebuilt": 0, "checks": 0, "crossed range_few upwards": 0, "small mode on": 0, "ninth one-lifetime type": 0}
seen, seen_now, dust_was_small = set(), set(), False
t0 = time.perf_counter()
with ParticleSystem(device=0, seed=L.SEED) as system:
    w = L.World(system, rng, 0)
    phase, target, cycle = "grow", 180, -1
    while w.frames < frames:
        big = sum(k in ("fifo", "nested_big") for k in w.kinds)
        # ---- one lifecycle action, then a stretch of frames
        r = rng.random()
        segs = w.segments()
        if phase == "grow":
            kind = str(rng.choice(["tiny"] * 3 + ["dust"] * 3 + ["two", "mid", "nested_small"])) if target < 300 else "dust"
            if r < 0.04 and big < 9 and segs < 60:
                kind = "fifo"
            if r > 0.98 and "nested_big" not in w.kinds:
                kind = "nested_big"
            before = segs
            w.add(kind); events["created"] += 1
            if before <= L.RANGE_FEW < w.segments(): events["crossed range_few upwards"] += 1
            dust_small = any(row == ("small",) for row, k2 in zip(w.paths(), w.kinds) if k2 == "dust")
            if dust_small and not dust_was_small: events["small mode on"] += 1
            dust_was_small = dust_small
            if sum(k == "fifo" for k in w.kinds) == 9 and kind == "fifo": events["ninth one-lifetime type"] += 1
            if w.segments() >= target: phase, target = "shrink", int(rng.integers(6, 30))
        else:
            if len(w.pairs) > 1:
                w.remove(int(rng.integers(0, len(w.pairs)))); events["despawned"] += 1
            if w.segments() <= target:
                cycle += 1
                phase, target = "grow", (560, 120, 200, 60)[cycle % 4]  # (every fourth growth passes fw_ctx::small_min)
        if r < 0.03 and w.pairs:
            w.rebuild(int(rng.integers(0, len(w.pairs)))); events["rebuilt"] += 1
        # (a burst of large one-lifetime types now and then: the spill rule)
        if rng.random() < 0.004 and w.segments() < 40:
            while sum(k == "fifo" for k in w.kinds) < 9:
                w.add("fifo"); events["created"] += 1
            events["ninth one-lifetime type"] += 1
        w.step(int(rng.integers(3, 25)) if w.segments() < 250 else int(rng.integers(1, 4)))
        seen_now = {p for row in w.paths() for p in row}
        seen |= seen_now
        if w.frames // 250 > events["checks"]:
            events["checks"] = w.frames // 250
            w.check("soak", limit=8)
            print(w.frames, "frames ok;", len(w.pairs), "spawners,", w.segments(), "segments, live", sum(sum(p.gpu.counts()) for p in w.pairs),
                  "| nested frames inside the FIFO launch / separate:", system.nest_frames(), flush=True)
    w.check("end of the soak", limit=10 ** 6)
    print("events", events, "paths seen", sorted(seen), "%.1f s" % (time.perf_counter() - t0))
print("SOAK-R05-OK")
