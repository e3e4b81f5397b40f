#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV of an in-flight bench run -> how the pairs' kernels overlap on the GPU:
wall time, time with 0/1/2/3+ kernels resident, and per-kernel-class duration stretch against the solo durations.
usage: overlap_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
cls = lambda n: ("refine" if "k_refine" in n else "sweep" if "k_spatial_sweep" in n else "view_eval" if "k_view_eval" in n else
                 "init" if "k_init" in n else "other")
dur = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    c = cls(r["Kernel_Name"])
    dur[c].append((e - s) / 1e6)
    if c != "other":
        ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
active = collections.Counter()
hist = collections.Counter()
combo = collections.Counter()
last = t0
for t, d, c in ev:
    n = sum(active.values())
    hist[min(n, 3)] += t - last
    combo[tuple(sorted(k for k, v in active.items() if v > 0 for _ in range(v)))] += t - last
    last = t
    active[c] += d
wall = (t1 - t0) / 1e6
print(f"wall {wall:.1f} ms; big kernels resident: " + ", ".join(f"{k}: {v / 1e6:.1f} ms ({100 * v / (t1 - t0):.0f} %)" for k, v in sorted(hist.items())))
for k, v in combo.most_common(8):
    print(f"  {'+'.join(k) or 'none':40s} {v / 1e6:8.1f} ms")
for c, v in dur.items():
    if c != "other":
        print(f"{c:10s} n={len(v):3d} mean {sum(v) / len(v):7.2f} ms  min {min(v):7.2f}  max {max(v):7.2f}")
