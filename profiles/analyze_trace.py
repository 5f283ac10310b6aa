#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: steady-state per-kernel durations and inter-kernel gaps."""
import collections
import csv
import sys


def main(path, tail=600):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-tail:]
    dur = collections.defaultdict(list)
    gaps = []
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur[r["Kernel_Name"].split("(")[0][-40:]].append(e - s)
        if prev_end is not None:
            gaps.append(s - prev_end)
        prev_end = e
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"last {len(rows)} dispatches, span {span/1e3:.1f} us")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"  {k:42s} n={len(v):5d} avg={sum(v)/len(v)/1e3:8.2f} us  p50={v2[len(v2)//2]/1e3:8.2f}  min={v2[0]/1e3:7.2f}  max={v2[-1]/1e3:7.2f}  total={sum(v)/span*100:5.1f}% of span")
    g = sorted(gaps)
    print(f"  gaps between consecutive dispatches: avg={sum(g)/len(g)/1e3:.2f} us p50={g[len(g)//2]/1e3:.2f} max={g[-1]/1e3:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 600)
