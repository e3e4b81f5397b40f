"""debug (needs a -DCSPM_COUNT_ALIVE build): fraction of refinement / view candidates still alive after each pyramid level"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSPM_LIB"] = os.path.join(ROOT, "crossscalepatchmatch_amd", "libcspm_alive.so")
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
cfg, l, r, _, _ = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
ctx = cs.StereoContext(0)
L = cs.load_library()
L.cspm_debug_alive.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def stats(tag):
    ctx.synchronize()
    a = (C.c_ulonglong * 16)()
    L.cspm_debug_alive(a, 1)
    ev = [a[8 + s] for s in range(5)]
    al = [a[s] for s in range(5)]
    print(tag, "evaluated lanes per level", ev, "alive after level / evaluated at level 0:", [round(x / max(1, ev[0]), 4) for x in al])
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=12345)
stats("init")
for it in range(3):
    ctx.pm_spatial(it, seed=12345); ctx.pm_view(it, seed=12345); stats(f"view {it}")
    ctx.pm_refine(it, seed=12345); stats(f"refine {it}")
