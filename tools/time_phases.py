#!/usr/bin/env python
"""Per-phase kernel times of ONE pair (hipEvents around every launch): init, iteration 0 (sweep, view, refine), and the
planes' checksum, so that variants of the library (CSPM_LIB=...) can be compared quickly and their results checked for
equality.  Usage: time_phases.py [config] [iters]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs  # noqa: E402
from crossscalepatchmatch_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg, l, r, _, _ = synth.make_config(name)
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
kind = os.environ.get("CSPM_TP_COST", "grd")  # grd | cen | img | grdvol
if kind == "cen":
    ctx.build_cost_cen(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
elif kind == "img":
    ctx.build_cost_img(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
else:
    ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=kind == "grdvol", sweep_pairs=os.environ.get("CSPM_TP_PAIRS", "0") != "0")
ctx.patchmatch(1, seed=12345)  # warm-up
ctx.synchronize()
ctx.enable_timing(True)
ctx.reset_timing()
t0 = time.perf_counter()
ctx.patchmatch(iters, seed=12345)
ctx.synchronize()
wall = (time.perf_counter() - t0) * 1e3
t = ctx.timing()
h = hashlib.sha256()
for v in (0, 1):
    npar, cost = ctx.get_planes(v)
    h.update(npar.tobytes())
    h.update(cost.tobytes())
print(os.environ.get("CSPM_LIB", "default"), name, f"iters={iters} wall={wall:.1f} ms ",
      " ".join(f"{k}={v['ms']:.2f}/{v['launches']}" for k, v in t.items()), "sha", h.hexdigest()[:12])
ctx.close()
