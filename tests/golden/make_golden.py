#!/usr/bin/env python
"""Generates tests/golden/*.npz.

The reference ships no test vectors and cannot be built in this image (needs OpenCV 2.4 + gflags), so these
fixtures are NOT reference output: they are produced by the oracle (oracle/cspm_oracle.c), whose restatement
is cross-checked by tests/test_oracle_vs_independent.py.  They pin (a) the oracle against accidental change
and (b) the HIP path on the GPU box against a committed answer.  PARITY UNPINNED with respect to the reference.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from crossscalepatchmatch_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from conftest import random_planes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cost_fixture():
    """T1 + T2: 64x48, max_dis 16: volumes (hashes + samples), max_cost, weights, 2000 GetPlaneCost records."""
    w, h, D = 64, 48, 16
    l, r, _, _ = synth.make_pair(w, h, D, regions=3, seed=11)
    out = dict(l=l, r=r, max_dis=D)
    for name, sn, lam in (("ss", 0, 0.0), ("cs", 5, 0.3)):
        pc = po.PlaneCost(l, r, D, 35, sn, lam)
        out[f"{name}_wgt"] = pc.scale_wgt()
        out[f"{name}_maxc"] = np.array([[pc.max_cost(v, s) for s in range(pc.levels)] for v in (0, 1)])
        out[f"{name}_vol_sha"] = np.array([[sha(pc.volume(v, s)) for s in range(pc.levels)] for v in (0, 1)])
        out[f"{name}_img_sha"] = np.array([[sha(pc.image(v, s)) for s in range(pc.levels)] for v in (0, 1)])
        out[f"{name}_vol0_d5"] = np.stack([pc.volume(v, 0)[5].copy() for v in (0, 1)])
        # the cells / max_cost of the device order (GRD cells with the contracted last step)
        out[f"{name}_maxc_dev"] = np.array([[pc.max_cost_dev(v, s) for s in range(pc.levels)] for v in (0, 1)])
        out[f"{name}_vol_dev_sha"] = np.array([[sha(pc.volume_dev(v, s)) for s in range(pc.levels)] for v in (0, 1)])
        out[f"{name}_vol0_d5_dev"] = np.stack([pc.volume_dev(v, 0)[5].copy() for v in (0, 1)])
        rng = np.random.default_rng(2024)
        n = 1000
        for v in (0, 1):
            xy, norm, point, param = random_planes(rng, n, w, h, D)
            out[f"{name}_v{v}_xy"] = xy
            out[f"{name}_v{v}_np"] = np.concatenate([norm, param], 1)
            out[f"{name}_v{v}_serial"] = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], v, po.SUM_SERIAL) for i in range(n)])
            out[f"{name}_v{v}_device"] = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], v, po.SUM_DEVICE) for i in range(n)])
    np.savez_compressed(os.path.join(OUT, "cost_64x48_d16.npz"), **out)


def pipeline_fixture():
    """T3/T4: 96x64, max_dis 16, 3 iterations, dis_scale 4: final 8-bit maps (raw and post-processed), plane hashes."""
    w, h, D = 96, 64, 16
    l, r, _, _ = synth.make_pair(w, h, D, regions=3, seed=12)
    out = dict(l=l, r=r, max_dis=D, dis_scale=4, seed=4242)
    for name, sn, lam in (("ss", 0, 0.0), ("cs", 5, 0.3)):
        pc = po.PlaneCost(l, r, D, 35, sn, lam)
        for sname, sched, order in (("raster_serial", po.SCHED_RASTER, po.SUM_SERIAL), ("raster_device", po.SCHED_RASTER, po.SUM_DEVICE),
                                    ("redblack_device", po.SCHED_REDBLACK, po.SUM_DEVICE)):
            pm = po.PatchMatch(l, r, D, 4)
            pm.run(3, pc, False, seed=4242, schedule=sched, sum_order=order, rb_rounds=1, rb_neighbours=4)
            k = f"{name}_{sname}"
            out[k + "_dis"] = np.stack([pm.dis(v) for v in (0, 1)])
            out[k + "_plane_sha"] = np.array([sha(np.concatenate([pm.planes(v)[..., 0:3], pm.planes(v)[..., 6:9]], -1)) for v in (0, 1)])
            out[k + "_cost_sum"] = np.array([pm.min_cost(v).sum() for v in (0, 1)])
            pm.postprocess()
            out[k + "_pp"] = np.stack([pm.dis(v) for v in (0, 1)])
    np.savez_compressed(os.path.join(OUT, "pipeline_96x64_d16.npz"), **out)


def img_fixture():
    """GrdPC / CSPC (plane_cost/grd_pc.cc, cspc.cc): 64x48, max_dis 16 -- 600 GetPlaneCost records per view and the final maps of a
    2-iteration raster run with post-processing."""
    w, h, D = 64, 48, 16
    l, r, _, _ = synth.make_pair(w, h, D, regions=3, seed=11)
    out = dict(l=l, r=r, max_dis=D, dis_scale=4, seed=777)
    for name, sn, lam in (("grdpc", 0, 0.0), ("cspc", 5, 0.3)):
        pc = po.PlaneCost(l, r, D, 35, sn, lam, cc="IMG")
        rng = np.random.default_rng(2025)
        n = 600
        for v in (0, 1):
            xy, norm, point, param = random_planes(rng, n, w, h, D)
            out[f"{name}_v{v}_xy"] = xy
            out[f"{name}_v{v}_np"] = np.concatenate([norm, param], 1)
            out[f"{name}_v{v}_serial"] = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], v, po.SUM_SERIAL) for i in range(n)])
            out[f"{name}_v{v}_device"] = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], v, po.SUM_DEVICE) for i in range(n)])
        pm = po.PatchMatch(l, r, D, 4)
        pm.run(2, pc, False, seed=777, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
        out[f"{name}_dis"] = np.stack([pm.dis(v) for v in (0, 1)])
        out[f"{name}_plane_sha"] = np.array([sha(np.concatenate([pm.planes(v)[..., 0:3], pm.planes(v)[..., 6:9]], -1)) for v in (0, 1)])
        pm.postprocess()
        out[f"{name}_pp"] = np.stack([pm.dis(v) for v in (0, 1)])
    np.savez_compressed(os.path.join(OUT, "imgcost_64x48_d16.npz"), **out)


if __name__ == "__main__":
    cost_fixture()
    pipeline_fixture()
    img_fixture()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
