"""debug: what the GPU box's host really offers (visible CPUs vs affinity vs cgroup quota) and how the oracle's OpenMP legs scale"""
import os, sys, time
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
from crossscalepatchmatch_amd import synth
from oracle import pyoracle as po
l, r, _, _ = synth.make_pair(200, 120, 32, 4, 5)
pc = po.PlaneCost(l, r, 32, 35, 5, 0.3)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    pm = po.PatchMatch(l, r, 32, 2)
    t = time.time(); pm.run(1, pc, False, seed=3, schedule=0, sum_order=po.SUM_SERIAL, threads=th); print("threads", th, f"{time.time()-t:.2f} s")
