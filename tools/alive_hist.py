"""debug (needs the -DCSPM_COUNT_ALIVE build: make -C crossscalepatchmatch_amd/csrc ../libcspm_alive.so): how many lanes of a wave still
carry a live candidate when a level pass of k_refine STARTS -- the reserve of a lane exchange (round-4 review, item 4): a wave leaves
a level pass only when all 64 candidates are rejected, so a pass with few live lanes wastes the others."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSPM_LIB"] = os.path.join(ROOT, "crossscalepatchmatch_amd", "libcspm_alive.so")
import numpy as np
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg, l, r, _, _ = synth.make_config(name)
ctx = cs.StereoContext(0)
L = cs.load_library()
L.cspm_debug_alive_hist.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def hist():
    ctx.synchronize()
    a = (C.c_ulonglong * 120)()
    L.cspm_debug_alive_hist(a, 1)
    return np.array(list(a), dtype=np.float64).reshape(3, 8, 5)
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=12345)
hist()
tot = np.zeros((3, 8, 5))
# relative cost of a level pass (in-image taps of a window at that level of C3: the deeper levels have clipped windows)
print(f"{name}: level passes of k_refine by live lanes at the start of the pass (3 iterations; buckets 1-8 / 9-16 / 17-32 / 33-48 / 49-64 lanes)")
for it in range(3):
    ctx.pm_spatial(it, seed=12345); ctx.pm_view(it, seed=12345); hist()
    ctx.pm_refine(it, seed=12345)
    h = hist()
    tot += h
    for g, gname in ((0, "steps 0-3"), (1, "steps 4-9")):
        for s in range(cfg["scale_num"]):
            n = h[g, s].sum()
            if n:
                print(f"  iteration {it} {gname} level {s}: passes {int(n):8d}  " + "  ".join(f"{100 * x / n:5.1f} %" for x in h[g, s]))
print("\nall three iterations:")
for g, gname in ((0, "steps 0-3"), (1, "steps 4-9")):
    for s in range(cfg["scale_num"]):
        n = tot[g, s].sum()
        print(f"  {gname} level {s}: passes {int(n):9d}  " + "  ".join(f"{100 * x / n:5.1f} %" for x in tot[g, s]) + f"   <= 16 live lanes: {100 * tot[g, s, :2].sum() / n:5.1f} %")
    n = tot[g].sum()
    print(f"  {gname} all levels: <= 16 live lanes in {100 * tot[g, :, :2].sum() / n:5.1f} % of the level passes, <= 32 in {100 * tot[g, :, :3].sum() / n:5.1f} %")
n = tot[:2].sum()
print(f"k_refine, all steps and levels: {int(n)} level passes; <= 8 live lanes {100 * tot[:2, :, 0].sum() / n:.1f} %, <= 16: {100 * tot[:2, :, :2].sum() / n:.1f} %, <= 32: {100 * tot[:2, :, :3].sum() / n:.1f} %")
# the bound of a perfect exchange: executed lane-passes against live lane-passes (bucket midpoints)
mid = np.array([4.5, 12.5, 24.5, 40.5, 56.5])
live = (tot[:2] * mid).sum()
print(f"live lanes x passes / (64 x passes) = {live / (64 * n):.3f}  (what a perfect compaction of the live candidates would leave of the executed level passes)")
