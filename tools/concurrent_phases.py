#!/usr/bin/env python
"""How two pairs' phases overlap on one GPU: wall time of a phase run on two contexts (two HIP streams) AT THE SAME TIME against the
same phase run on one context -- sweep || sweep, refine || refine, sweep || refine.  Host threads issue the (synchronising)
single-phase entries concurrently; ctypes releases the GIL."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs  # noqa: E402
from crossscalepatchmatch_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctxs = []
for k in range(2):
    cfg, l, r, _, _ = synth.make_config(name, index=k)
    c = cs.StereoContext(0)
    c.set_images(l, r)
    c.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    c.patchmatch(1, seed=12345)
    c.synchronize()
    ctxs.append(c)


def run(fns):
    th = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return (time.perf_counter() - t0) * 1e3


A, B = ctxs
for it in (1, 2):
    one = run([lambda: A.pm_spatial(it, seed=1)])
    two = run([lambda: A.pm_spatial(it, seed=1), lambda: B.pm_spatial(it, seed=1)])
    print(f"iteration {it}: sweep alone {one:.1f} ms, sweep || sweep {two:.1f} ms")
    one = run([lambda: (A.pm_refine(it, seed=1), A.synchronize())])
    two = run([lambda: (A.pm_refine(it, seed=1), A.synchronize()), lambda: (B.pm_refine(it, seed=1), B.synchronize())])
    print(f"iteration {it}: refine alone {one:.1f} ms, refine || refine {two:.1f} ms")
    mix = run([lambda: A.pm_spatial(it + 1, seed=1), lambda: (B.pm_refine(it, seed=2), B.synchronize())])
    print(f"iteration {it}: sweep || refine {mix:.1f} ms")
    v1 = run([lambda: (A.pm_view(it, seed=1), A.synchronize())])
    v2 = run([lambda: (A.pm_view(it, seed=1), A.synchronize()), lambda: (B.pm_view(it, seed=1), B.synchronize())])
    print(f"iteration {it}: view alone {v1:.1f} ms, view || view {v2:.1f} ms")
for c in ctxs:
    c.close()
