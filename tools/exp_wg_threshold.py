#!/usr/bin/env python
"""sweep time alone by image size and CSPM_OPT_SWEEP_WG (2, 3; 0 = the library's choice): where does a third workgroup per CU start to pay?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import capi, synth
for w, h in ((1242, 375), (1242, 600), (1500, 800), (1600, 1000), (2000, 1200), (3000, 2000)):
    l, r, _, _ = synth.make_pair(w, h, 128, regions=12, seed=5)
    ctx = cs.StereoContext(0)
    ctx.set_images(l, r)
    ctx.build_cost_grd(128, 35, 5, 0.3)
    ctx.pm_init(seed=1)
    ctx.pm_spatial(0, seed=1)
    st = [ctx.get_planes(v) for v in (0, 1)]
    res = []
    for wg in (2, 3, 0):
        ctx.set_option(capi.OPT_SWEEP_WG, wg)
        ts = []
        for rep in range(3):
            for v in (0, 1):
                ctx.set_planes(v, *st[v])
            ctx.synchronize()
            t = time.perf_counter(); ctx.pm_spatial(1, seed=1); ts.append((time.perf_counter() - t) * 1e3)
        res.append(min(ts))
    print(f"{w}x{h}: 2*min(w,h) = {2 * min(w, h)}: sweep {res[0]:.1f} ms at 2 per CU, {res[1]:.1f} at 3, library's choice {res[2]:.1f}")
    ctx.close()
