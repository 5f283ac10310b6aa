"""The path thresholds of csrc/fw_engine.h (range_few, small_min, wide_min / wide_mid, range_min) were fitted to measurements of the boxes
of rounds 4-5, whose kernel times differ by 17 % from box to box.  tools/threshold_sweep.py measures, on the box
it runs on, the product's own choice against every forced path at six (emitters x particles) points and says whether it is within 10 %
of the best (profiles/r06/threshold_check*.txt: two leases).  This test runs the same code at three of the points as a guard against a
threshold that has become WRONG on this box -- 30 % slower than the best path, three times in a row -- not as a timing benchmark (a shared or throttled GPU
moves single measurements by more than 10 %)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_the_product_picks_a_path_close_to_the_best_forced_one(monkeypatch):
    for k in list(os.environ):
        if k.startswith("FW_") and k not in ("FW_LIB_PATH",):
            monkeypatch.delenv(k, raising=False)
    import threshold_sweep as T

    monkeypatch.setenv("FW_ENABLE_KNOBS", "1")
    points = [(64, 700), (512, 300), (1024, 1000)]
    ok, rows = T.sweep(points, tol=1.3, out=sys.stderr)
    for _ in range(2):  # two more looks before failing: timing on a box somebody else may be using (round 6: one lease measured the SAME
        if ok:          # path 24 % apart in two consecutive runs, profiles/r06 threshold sweeps)
            break
        ok, rows = T.sweep(points, tol=1.3, out=sys.stderr)
    assert ok, rows
    # the product's choices at these points, as the thresholds promise: few small emitters on range rings, hundreds of small ones on a
    # workgroup each (their bound passes what a wave is given)
    # (... a thousand mid-size ones on a WAVE each: what the first run of this sweep found, fw_ctx::wave_all_min)
    assert [r[2] for r in rows] == ["range", "workgroup", "wave"], rows
