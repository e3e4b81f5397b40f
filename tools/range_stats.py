"""debug (needs `make -C crossscalepatchmatch_amd/csrc ../libcspm_rowstats.so`): per phase and pyramid level, how the level passes of the
row engine's waves split between full cell mode, range-restricted cell mode (with / without the weight table) and the general taps."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSPM_LIB"] = os.path.join(ROOT, "crossscalepatchmatch_amd", "libcspm_rowstats.so")
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
cfg, l, r, _, _ = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
ctx = cs.StereoContext(0)
L = cs.load_library()
L.cspm_debug_rangestats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def stats(tag):
    ctx.synchronize()
    a = (C.c_ulonglong * 1024)()
    L.cspm_debug_rangestats(a, 1)
    for slot in range(16):
        for s in range(5):
            g = [a[(slot * 8 + s) * 8 + b] for b in range(8)]
            tot = g[0] + g[7]
            if tot:
                name = "init" if slot == 0 else "view" if slot == 1 else f"refine step {slot - 2}"
                took = g[2] + g[3]
                print(f"{tag:10s} {name:15s} level {s}: passes {tot:8d}  full {g[7] / tot:5.3f}  range+wtab {g[2] / tot:5.3f}  range {g[3] / tot:5.3f} (mean ND {g[5] / max(took, 1):5.1f})  "
                      f"not all interpolating {g[1] / tot:5.3f}  too many disparities {g[4] / tot:5.3f} (mean ND {g[6] / max(g[4], 1):5.1f})")
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=12345)
stats("init")
for it in range(3):
    ctx.pm_spatial(it, seed=12345); ctx.pm_view(it, seed=12345); stats(f"view {it}")
    ctx.pm_refine(it, seed=12345); stats(f"refine {it}")
# wave cycles per row by level (s_memtime, 100 MHz constant clock on gfx950: 1 tick = 10 ns), summed over the whole run above
L.cspm_debug_rowtime.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
t = (C.c_ulonglong * 64)()
L.cspm_debug_rowtime(t, 1)
for s in range(5):
    g = [t[s * 8 + k] for k in range(8)]
    if g[3]:
        print(f"level {s}: table rows {g[3]:9d}: {g[0] / g[3]:7.1f} ticks per row, waiting for the strips {g[1] / g[3]:6.1f}, table build {g[2] / g[3]:6.1f}")
    if g[7]:
        print(f"level {s}: general DMA rows {g[7]:9d}: {g[4] / g[7]:7.1f} ticks per row, waiting for the strips {g[5] / g[7]:6.1f}")
L.cspm_debug_unionstat.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
u = (C.c_ulonglong * 64)()
L.cspm_debug_unionstat(u, 1)
for s in range(5):
    g = [u[s * 8 + k] for k in range(5)]
    if sum(g):
        print(f"level {s}: passes with too wide a SPAN of disparities {sum(g):8d}; their UNION of disparities: <= 8: {g[0] / sum(g):5.3f}  <= 11: {g[1] / sum(g):5.3f}  "
              f"<= 16: {g[2] / sum(g):5.3f}  <= 24: {g[3] / sum(g):5.3f}  more: {g[4] / sum(g):5.3f}")
