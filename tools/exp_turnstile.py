#!/usr/bin/env python
"""Experiment: K pairs in flight, each driven phase by phase from its own host thread (the single-phase C-ABI entries), with and without
a TURNSTILE that lets only one context's PlaneRefinement run at a time -- the other contexts' sweeps and view propagations run beside it.
refine || refine gains nothing (55 ms for two against 2 x 30), sweep || refine does: does forbidding the former buy throughput?

    python tools/exp_turnstile.py [pairs per context] [contexts] [mode: free | refine_lock | sweep_lock | both]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crossscalepatchmatch_amd as cs  # noqa: E402
from crossscalepatchmatch_amd import capi, synth  # noqa: E402

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else "free"
pairs = [synth.make_config("C3", index=k) for k in range(4)]
cfg = pairs[0][0]
refine_lock = threading.Lock() if mode in ("refine_lock", "both") else None
sweep_lock = threading.Lock() if mode in ("sweep_lock", "both") else None


class Null:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def worker(k, ctx, out):
    for i in range(npairs):
        _, l, r, _, _ = pairs[(k + i) % len(pairs)]
        ctx.set_images(l, r)
        ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
        ctx.pm_init(seed=12345)
        for it in range(3):
            with (sweep_lock or Null()):
                ctx.pm_spatial(it, seed=12345)   # synchronising entry
            ctx.pm_view(it, seed=12345)
            with (refine_lock or Null()):
                ctx.pm_refine(it, seed=12345)
                ctx.synchronize()
        ctx.disparity_u8(0, 1)
    out[k] = True


ctxs = [cs.StereoContext(0) for _ in range(nctx)]
for c in ctxs:
    if nctx >= 2 and not os.environ.get("CSPM_SWEEP_FOLD"):
        c.set_option(capi.OPT_SWEEP_FOLD, 1)
# warm-up
w = {}
th = [threading.Thread(target=worker, args=(k, c, w)) for k, c in enumerate(ctxs)]
npairs_keep, npairs = npairs, 1
[t.start() for t in th]; [t.join() for t in th]
npairs = npairs_keep
out = {}
th = [threading.Thread(target=worker, args=(k, c, out)) for k, c in enumerate(ctxs)]
t0 = time.perf_counter()
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
print(f"mode={mode} contexts={nctx} pairs={npairs * nctx}: {dt / (npairs * nctx) * 1e3:.1f} ms per pair")
for c in ctxs:
    c.close()
