#!/usr/bin/env python
"""Small fixed workload for rocprofv3 --pmc passes: C3-size pair, plane-cost construction, random init and one
PlaneRefinement (10 launches of the dominant kernel).  Usage: prof_workload.py [config] [volumes] [noexit]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs  # noqa: E402
from crossscalepatchmatch_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
volumes = "volumes" in sys.argv[2:]
early = 0 if "noexit" in sys.argv[2:] else 1
cfg, l, r, _, _ = synth.make_config(name)
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=volumes)
ctx.pm_init(seed=12345)
ctx.pm_refine(0, seed=12345, early_exit=early)
ctx.synchronize()
print("taps per launch", 2 * ctx.taps_per_view_pass())
ctx.close()
