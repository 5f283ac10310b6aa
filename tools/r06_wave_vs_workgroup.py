"""Round 6: tools/threshold_sweep.py found the product 27-44 % behind the best forced path at 1024 emitters x 1000 particles -- a WAVE
per type beats a WORKGROUP per type there, although a type of 1000 particles is a "wide" one (fw_ctx::small_max 768).  Where is the
crossover?  us per frame of the two roles of fw_k_update_small over (types x particles per type).   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import threshold_sweep as T
pts = [(n, p) for n in (512, 768, 1024, 1536, 2048) for p in (300, 600, 1000, 1500, 2000)]
if len(sys.argv) > 1:
    pts = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for n, p in pts:
    w, wp = T.measure(n, p, T.FORCED["wave"], frames=150, reps=2)
    g, gp = T.measure(n, p, T.FORCED["workgroup"], frames=150, reps=2)
    d, dp = T.measure(n, p, {}, frames=150, reps=2)
    print(f"{n:5d} x {p:5d}: wave {w:7.1f} ({wp})  workgroup {g:7.1f} ({gp})  product {d:7.1f} ({dp})  wave / workgroup {w / g:.2f}", flush=True)
