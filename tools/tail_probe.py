"""How much of a row-kernel launch is ramp and tail: the C3 texture at 1x, 2x, 3x the image height (time = fixed + per-row part).
GPU box: python tools/tail_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
for h in (375, 750, 1125):
    cfg = dict(synth.CONFIGS["C3"]); cfg["h"] = h
    l, r, _, _ = synth.make_pair(cfg["w"], cfg["h"], cfg["max_dis"], cfg["regions"] * h // 375, cfg["seed"])
    ctx = cs.StereoContext(0)
    ctx.set_images(l, r)
    ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    ctx.patchmatch(1, seed=12345); ctx.synchronize()
    ctx.enable_timing(True); ctx.reset_timing()
    ctx.patchmatch(3, seed=12345); ctx.synchronize()
    t = ctx.timing()
    print(h, {k: round(v['ms'], 2) for k, v in t.items()}, flush=True)
    del ctx
