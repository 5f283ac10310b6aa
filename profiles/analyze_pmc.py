#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel over the last dispatches of each *_counter_collection.csv in a directory."""
import collections
import csv
import glob
import sys


def main(d, tail=60):
    for path in sorted(glob.glob(d + "/*counter_collection.csv")):
        rows = list(csv.DictReader(open(path)))
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(path.split("/")[-1])
        for k, cs in per.items():
            if "fw_k" not in k:
                continue
            print("  " + k + "  " + "  ".join(f"{c}={sum(v[-tail:]) / len(v[-tail:]):.4g}" for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
