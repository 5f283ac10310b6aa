#!/usr/bin/env python3
"""Side-by-side counters of the update kernel and its structural twin (tools/pmc_compare.sh output directory)."""
import collections
import csv
import glob
import os
import sys


def load(path, want):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if want in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[-60:]) / len(v[-60:]) for k, v in per.items() if v}


def main(d):
    for real in sorted(glob.glob(d + "/**/real_*counter_collection.csv", recursive=True)):
        twin = real.replace("real_", "twin_")
        if not os.path.exists(twin):
            continue
        a, b = load(real, "fw_k_update"), load(twin, "k_twin")
        print(os.path.basename(real).replace("real_", "").replace("_counter_collection.csv", ""))
        for k in sorted(a):
            tv = b.get(k)
            print(f"  {k:48s} real {a[k]:14.1f}   twin {tv if tv is None else format(tv, '14.1f')}   ratio {a[k] / tv if tv else float('nan'):.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
