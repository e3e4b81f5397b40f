#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output directories (one counter group per run): sum per kernel and counter."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
want = sys.argv[2:] or ["init", "refine"]
for d in sorted(glob.glob(root + "/pmc*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(d)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if any(w in k for w in want):
            for a, b in sorted(v.items()):
                print(f"{k:45s} {a:32s} {b:.6g}")
