#!/usr/bin/env python
"""Who loses when a raster sweep and a refinement share the GPU (round-5 review, Weak 4: sweep || refine takes 45 ms where
max(20, 30) is the goal)?  Two contexts on two HIP streams, the sweep of pair A and the refinement of pair B started together from
two host threads; EACH kernel's own elapsed time is recorded beside the pair's, for a few start offsets (which kernel gets the CUs
first decides who is resident).  Rocprofv3's counter collection serialises dispatches, so per-kernel counters of a co-run cannot be
had; this probe plus build / option toggles (CSPM_LIB=<variant>, CSPM_TABLE_VOLUMES=0, CSPM_SWEEP_WG=1) names the resource instead.

    [CSPM_LIB=...] python tools/corun_probe.py [C3] [repeats]
"""
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs  # noqa: E402
from crossscalepatchmatch_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctxs = []
for k in range(2):
    cfg, l, r, _, _ = synth.make_config(name, index=k)
    c = cs.StereoContext(0)
    c.set_images(l, r)
    c.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    c.patchmatch(1, seed=12345)
    c.synchronize()
    ctxs.append(c)
A, B = ctxs
state = [[c.get_planes(v) for v in (0, 1)] for c in ctxs]


def reset():
    for c, st in zip(ctxs, state):
        for v in (0, 1):
            c.set_planes(v, *st[v])
        c.synchronize()


def timed(fn, delay_ms, out, key):
    def body():
        if delay_ms > 0:
            time.sleep(delay_ms / 1e3)
        t = time.perf_counter()
        fn()
        out[key] = (time.perf_counter() - t) * 1e3
    return body


def sweep():
    A.pm_spatial(1, seed=1)  # synchronising entry


def refine():
    B.pm_refine(1, seed=2)
    B.synchronize()


def corun(d_sweep, d_refine):
    res = []
    for _ in range(reps):
        reset()
        out = {}
        th = [threading.Thread(target=timed(sweep, d_sweep, out, "sweep")), threading.Thread(target=timed(refine, d_refine, out, "refine"))]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        out["both"] = (time.perf_counter() - t0) * 1e3
        res.append(out)
    return {k: statistics.median(r[k] for r in res) for k in ("sweep", "refine", "both")}


def alone(fn):
    v = []
    for _ in range(reps):
        reset()
        t = time.perf_counter()
        fn()
        v.append((time.perf_counter() - t) * 1e3)
    return statistics.median(v)


print(f"lib={os.environ.get('CSPM_LIB', 'default')} TABLE_VOLUMES={os.environ.get('CSPM_TABLE_VOLUMES', '1')} SWEEP_WG={os.environ.get('CSPM_SWEEP_WG', 'default')}")
print(f"  alone: sweep {alone(sweep):.1f} ms, refine {alone(refine):.1f} ms")
for ds, dr, tag in ((0, 0, "together"), (0, 3, "sweep 3 ms first"), (3, 0, "refine 3 ms first")):
    m = corun(ds, dr)
    print(f"  co-run, {tag:18s}: sweep {m['sweep']:.1f} ms, refine {m['refine']:.1f} ms, both done after {m['both']:.1f} ms")
for c in ctxs:
    c.close()
