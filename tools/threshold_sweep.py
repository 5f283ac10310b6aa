#!/usr/bin/env python3
"""Round 6: do the path thresholds of csrc/fw_engine.h -- range_few 192, small_min 352, wide_min 768,
wide_mid 1400, wave_all_min 896 (added because this sweep found the miss), range_min 8192 -- pick a path within 10 % of the BEST forced path on THIS box?

Six (emitters x particles per emitter) points on either side of the thresholds; at each, us per frame (pipelined, best of 3 x 200 frames)
of the product's own choice and of every forced path:
    range      every type on an in-place range ring whatever its size           FW_RANGE_MIN=0 FW_RANGE_FEW=100000 FW_SMALL=0
    compacting one workgroup chain per type on the compacting kernels            FW_RANGE_MIN=4000000000 FW_RANGE_FEW=0 FW_SMALL=0
    wave       one WAVE per type (fw_k_update_small, narrow role)                FW_RANGE_FEW=0 FW_SMALL_MIN=0 FW_SMALL_MAX=2000000000 FW_WIDE_MAX=0
    workgroup  one WORKGROUP per type (fw_k_update_small, wide role)             FW_RANGE_FEW=0 FW_SMALL_MIN=0 FW_WIDE_MIN=0 FW_WIDE_MAX=2000000000 FW_SMALL_MAX=0 FW_WAVE_ALL_MIN=0
The last line says PASS when the product is within TOL (default 1.10) of the best at every point.  The GPU test
tests/test_gpu_thresholds.py runs the same code at three of the points.      python tools/threshold_sweep.py      (GPU box)"""
import os
import sys
import time

os.environ["FW_ENABLE_KNOBS"] = "1"
import numpy as np  # noqa: E402

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevy_firework_amd import workloads  # noqa: E402
from bevy_firework_amd.system import ParticleSystem  # noqa: E402

DT = np.float32(1 / 60)
KNOBS = ("FW_RANGE", "FW_RANGE_MIN", "FW_RANGE_FEW", "FW_SMALL", "FW_SMALL_MIN", "FW_SMALL_MAX", "FW_WIDE_MIN", "FW_WIDE_MAX", "FW_FIFO", "FW_WAVE_ALL_MIN")
FORCED = {
    "range": {"FW_RANGE_MIN": "0", "FW_RANGE_FEW": "100000", "FW_SMALL": "0"},
    "compacting": {"FW_RANGE_MIN": "4000000000", "FW_RANGE_FEW": "0", "FW_SMALL": "0"},
    "wave": {"FW_RANGE_FEW": "0", "FW_RANGE_MIN": "4000000000", "FW_SMALL_MIN": "0", "FW_SMALL_MAX": "2000000000", "FW_WIDE_MAX": "0"},
    "workgroup": {"FW_RANGE_FEW": "0", "FW_RANGE_MIN": "4000000000", "FW_SMALL_MIN": "0", "FW_WIDE_MIN": "0", "FW_WIDE_MAX": "2000000000", "FW_SMALL_MAX": "0",
                  "FW_WAVE_ALL_MIN": "0"},
}
# (emitters, particles per emitter): below / above range_few, around small_min, around wide_min / wide_mid, and the ring threshold
POINTS = [(64, 700), (256, 300), (512, 300), (512, 1000), (1024, 1000), (256, 9000)]


def measure(n_em, per, env, frames=200, reps=3):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        with ParticleSystem(seed=workloads.SEED) as ps:
            ems = workloads.many_emitters(n_em, per)
            hs = [ps.spawn(ems[e][0], ems[e][1], uid=e) for e in range(n_em)]
            ps.update(DT)
            for _ in range(90):
                ps.step(DT)
            best = 1e9
            for _ in range(reps):
                ps.synchronize()
                t0 = time.perf_counter()
                for _ in range(frames):
                    ps.step(DT)
                ps.synchronize()
                best = min(best, (time.perf_counter() - t0) / frames * 1e6)
            mode = hs[0].update_mode(0)
            path = {0: "compacting", 1: "fifo", 2: "range", 3: "wave", 4: "workgroup"}[mode]
    finally:
        for k in KNOBS:
            os.environ.pop(k, None)
    return best, path


def sweep(points=POINTS, tol=1.10, out=sys.stdout):
    ok = True
    rows = []
    for n_em, per in points:
        us, path = measure(n_em, per, {})
        forced = {}
        for name, env in FORCED.items():
            f_us, f_path = measure(n_em, per, env)
            if f_path == name:  # (a path that cannot take this size -- e.g. a wave for 9000 particles -- is not a candidate)
                forced[name] = f_us
        best_name = min(forced, key=forced.get)
        ratio = us / forced[best_name]
        good = ratio <= tol
        ok &= good
        rows.append((n_em, per, path, us, best_name, forced[best_name], ratio, good))
        print(f"{n_em:5d} x {per:5d}: product -> {path:10s} {us:7.1f} us | " + "  ".join(f"{k} {v:7.1f}" for k, v in forced.items()) +
              f" | best {best_name} -> product / best = {ratio:.3f} {'ok' if good else 'OUTSIDE ' + str(tol)}", file=out, flush=True)
    print(("PASS" if ok else "FAIL") + f": the product's choice is within {tol:.2f}x of the best forced path at {sum(r[-1] for r in rows)} of {len(rows)} points",
          file=out, flush=True)
    return ok, rows


if __name__ == "__main__":
    tol = float(os.environ.get("TOL", "1.10"))
    raise SystemExit(0 if sweep(tol=tol)[0] else 1)
