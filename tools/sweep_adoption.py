"""debug: how many pixels adopt a neighbour's plane in each raster sweep (C3), and how often the two candidates of a pixel are
bitwise identical / identical to the pixel's own plane (the evaluations the sweep already skips)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
cfg, l, r, _, _ = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=12345)
for it in range(3):
    before = [ctx.get_planes(v)[0] for v in (0, 1)]
    ctx.pm_spatial(it, seed=12345)
    after = [ctx.get_planes(v)[0] for v in (0, 1)]
    for v in (0, 1):
        ch = np.any(before[v] != after[v], axis=-1)
        inc = 1 if it % 2 == 0 else -1
        a = after[v]
        # candidates as the sweep sees them: final plane of the x- and y-predecessor
        px = np.roll(a, inc, axis=1); py = np.roll(a, inc, axis=0)
        same01 = np.all(px == py, axis=-1)
        own0 = np.all(px == before[v], axis=-1); own1 = np.all(py == before[v], axis=-1)
        evals = (~own0).astype(int) + (~own1 & ~same01).astype(int)
        print(f"sweep {it} view {v}: adopted {ch.mean():.3f}; candidates identical {same01.mean():.3f}; x-pred == own {own0.mean():.3f}, y-pred == own {own1.mean():.3f}; "
              f"evaluations per pixel {evals.mean():.3f} (0: {np.mean(evals == 0):.3f}, 1: {np.mean(evals == 1):.3f}, 2: {np.mean(evals == 2):.3f})")
    ctx.pm_view(it, seed=12345); ctx.pm_refine(it, seed=12345)
