"""would a per-pixel LOWER BOUND on the remaining pyramid levels make the early exit of the row engine bite?

LB_s(p) = sum over the window of w(p,q) * min_d cell_s(q,d)  (valid: every interpolated cell >= the minimum over d and all terms
are >= 0).  After level s a candidate is proven rejected when  cost_so_far + sum_{s'>s} wgt_s' * LB_s'(p) >= thresh  (today: the same
test with LB = 0).  The script runs PatchMatch on the GPU (C3 by default), takes the plane field before the last refinement,
draws refinement candidates exactly as PlaneRefinement does (cs_patchmatch.cc:292-345) for runs of 64 adjacent pixels (= one
wavefront of the row engine), evaluates their per-level costs with the oracle, and reports per halving step the fraction of
lanes -- and of whole wavefronts -- whose rejection is proven after each level, with LB = 0 and with the bound.
Usage (GPU box): python tests/studies/lb_exit_study.py [C3|C2] [n_runs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
from oracle import pyoracle as po

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg, l, r, _, _ = synth.make_config(name)
W, H, D = cfg["w"], cfg["h"], cfg["max_dis"]
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
ctx.build_cost_grd(D, 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=12345)
for it in range(3):
    ctx.pm_spatial(it, seed=12345); ctx.pm_view(it, seed=12345)
    if it < 2:
        ctx.pm_refine(it, seed=12345)
npar, mincost = ctx.get_planes(0)  # the state PlaneRefinement of the last iteration starts from
pc = po.PlaneCost(l, r, D, 35, cfg["scale_num"], cfg["reg_lambda"])
wgt = pc.scale_wgt()
S = pc.levels
lut = np.exp(-np.arange(1000) / 10.0)
# min over d of the cells, per level
minc = [pc.volume(0, s).min(axis=0) for s in range(S)]
imgs = [pc.image(0, s).astype(np.int64) for s in range(S)]


def lower_bound(s, cx, cy):
    w_, h_, _ = pc.dims(s)
    y0, y1, x0, x1 = max(0, cy - 17), min(h_ - 1, cy + 17), max(0, cx - 17), min(w_ - 1, cx + 17)
    sad = np.abs(imgs[s][y0:y1 + 1, x0:x1 + 1] - imgs[s][cy, cx]).sum(-1)
    return float((lut[sad] * minc[s][y0:y1 + 1, x0:x1 + 1]).sum())


rng = np.random.default_rng(1)
steps = po.lib().csor_refine_steps(D)
dead0 = np.zeros((steps, S)); deadlb = np.zeros((steps, S)); lanes = np.zeros(steps)
wave0 = np.zeros((steps, S)); wavelb = np.zeros((steps, S)); rejected = np.zeros(steps)
tight = []
for run in range(n_runs):
    y = int(rng.integers(20, H - 20)); x0 = int(rng.integers(0, (W - 64) // 64 + 1)) * 64
    z, nn = D / 2.0, 1.0
    lbs = {}
    for x in range(x0, x0 + 64):
        lb = np.zeros(S)
        cx, cy = x, y
        for s in range(S):
            lb[s] = lower_bound(s, cx, cy)
            cx //= 2; cy //= 2
        lbs[x] = lb
        inc = pc.level_costs(x, y, npar[y, x, :3], npar[y, x, 3:], 0)
        tight.append(lb / np.maximum(inc, 1e-300))
    for st in range(steps):
        d0 = np.zeros((64, S), bool); dl = np.zeros((64, S), bool)
        for k, x in enumerate(range(x0, x0 + 64)):
            n0, prm0 = npar[y, x, :3], npar[y, x, 3:]
            pz = prm0[0] * x + prm0[1] * y + prm0[2] + rng.uniform(-z, z)
            n = n0 + rng.uniform(-nn, nn, 3)
            n /= max(np.linalg.norm(n), 1e-8)
            prm = po.plane_param(n, [x, y, pz])
            lv = pc.level_costs(x, y, n, prm, 0)
            partial = np.cumsum(lv * wgt)
            rest = np.array([np.sum(wgt[s + 1:] * lbs[x][s + 1:]) for s in range(S)])
            d0[k] = partial >= mincost[y, x]
            dl[k] = partial + rest >= mincost[y, x]
            rejected[st] += partial[-1] >= mincost[y, x]
        dead0[st] += d0.sum(0); deadlb[st] += dl.sum(0); lanes[st] += 64
        wave0[st] += d0.all(0); wavelb[st] += dl.all(0)
        z /= 2.0; nn /= 2.0
tight = np.array(tight)
print(f"{name}: {n_runs} wavefront runs of 64 pixels, view 0, plane field before the 3rd refinement; scale weights {np.round(wgt, 4)}")
print("tightness LB_s / level cost of the incumbent plane, median per level:", np.round(np.median(tight, 0), 3))
print("step  rejected  | lanes proven rejected after level 0..%d, LB = 0      | with the lower bound              | whole waves, LB = 0 | whole waves, bound" % (S - 1))
for st in range(steps):
    f = lambda a: " ".join(f"{v:5.3f}" for v in a)
    print(f"{st:4d}  {rejected[st] / lanes[st]:7.3f}  | {f(dead0[st] / lanes[st])} | {f(deadlb[st] / lanes[st])} | {f(wave0[st] / n_runs)} | {f(wavelb[st] / n_runs)}")
# taps saved: a lane (wave) stops after the first level at which it is proven rejected; level tap shares ~ equal
share = np.array([pc.taps(W // 2, H // 2)] * 1, float)
print("executed level-evaluations per candidate (of %d), lanes compacted perfectly: LB=0 %.2f, bound %.2f;  whole waves only: LB=0 %.2f, bound %.2f"
      % (S, np.mean([S - (dead0[st] / lanes[st])[:-1].sum() for st in range(steps)]), np.mean([S - (deadlb[st] / lanes[st])[:-1].sum() for st in range(steps)]),
         np.mean([S - (wave0[st] / n_runs)[:-1].sum() for st in range(steps)]), np.mean([S - (wavelb[st] / n_runs)[:-1].sum() for st in range(steps)])))
