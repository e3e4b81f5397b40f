"""debug (needs `make -C crossscalepatchmatch_amd/csrc ../libcspm_rowstats.so`): per phase and pyramid level, how many window rows of
the row engine have all 64 lanes on the interpolation branch, and how many integer disparities a wave touches on such a row."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSPM_LIB"] = os.path.join(ROOT, "crossscalepatchmatch_amd", "libcspm_rowstats.so")
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
if len(sys.argv) > 1 and sys.argv[1] == "real":  # the Middlebury motorcycle pair (scikit-image's copy, or the committed crop)
    from crossscalepatchmatch_amd import realdata
    cfg, l, r, _ = realdata.load_full() or realdata.load_crop()
else:
    cfg, l, r, _, _ = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
print("view propagation in", "TARGET order (CSPM_OPT_VIEW_SORT = 1)" if os.environ.get("CSPM_VIEW_SORT", "1") != "0" else "source order (CSPM_VIEW_SORT=0)")
ctx = cs.StereoContext(0)
L = cs.load_library()
L.cspm_debug_rowstats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
NAMES = ["nd<=4", "nd<=8", "nd<=16", "nd<=32", "nd>32", "some lane invalid", "unstaged"]
def stats(tag):
    ctx.synchronize()
    a = (C.c_ulonglong * 1024)()
    L.cspm_debug_rowstats(a, 1)
    for slot in range(16):
        for s in range(5):
            row = [a[(slot * 8 + s) * 8 + b] for b in range(7)]
            tot = sum(row)
            if tot:
                name = "init" if slot == 0 else "view" if slot == 1 else f"refine step {slot - 2}"
                print(f"{tag:10s} {name:15s} level {s}: rows {tot:9d}  " + "  ".join(f"{NAMES[b]} {row[b] / tot:6.3f}" for b in range(7)))
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=12345)
stats("init")
for it in range(3):
    ctx.pm_spatial(it, seed=12345); ctx.pm_view(it, seed=12345); stats(f"view {it}")
    ctx.pm_refine(it, seed=12345); stats(f"refine {it}")
