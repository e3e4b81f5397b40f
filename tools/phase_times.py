#!/usr/bin/env python
"""Kernel time of every phase of every iteration of ONE pair (hipEvents; phases run through the single-phase entries).
Usage: phase_times.py [config] [iters]     env CSPM_TP_PAIRS=0: raster sweep on the fused cells"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs  # noqa: E402
from crossscalepatchmatch_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg, l, r, _, _ = synth.make_config(name)
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], sweep_pairs=os.environ.get("CSPM_TP_PAIRS", "0") != "0")
ctx.patchmatch(1, seed=12345)
ctx.synchronize()
ctx.enable_timing(True)


def timed(label, fn):
    ctx.reset_timing()
    fn()
    ctx.synchronize()
    t = ctx.timing()
    return f"{label}={sum(v['ms'] for v in t.values()):.2f}"


out = [timed("init", lambda: ctx.pm_init(seed=12345))]
for it in range(iters):
    out.append(timed(f"sweep{it}", lambda: ctx.pm_spatial(it, seed=12345)))
    out.append(timed(f"view{it}", lambda: ctx.pm_view(it, seed=12345)))
    out.append(timed(f"refine{it}", lambda: ctx.pm_refine(it, seed=12345)))
print(os.environ.get("CSPM_LIB", "default"), name, "pairs" if os.environ.get("CSPM_TP_PAIRS", "0") != "0" else "fused", " ".join(out))
ctx.close()
