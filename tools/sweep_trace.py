"""debug: per-item stage latencies of the persistent sweep (needs a -DCSPM_SWEEP_TRACE build of the library)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth, capi
capi._SO = os.environ["CSPM_TRACE_LIB"]
cfg, l, r, _, _ = synth.make_config("C3")
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.pm_init(seed=1)
ctx.pm_spatial(0, seed=1, schedule=0)
ctx.pm_spatial(2, seed=1, schedule=0)   # dumps the first sweep
t = np.fromfile(os.environ["CSPM_SWEEP_TRACE_FILE"], dtype=np.int64).reshape(-1, 8).astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
print("sweep span us", t[:, 7].max() - t0, "items", len(t))
d = np.diff(t, axis=1)
names = ["claim->decoded", "decoded->flags ok", "flags->planes loaded", "planes->eval done", "eval->sync", "sync->decided+stored", "stored->flag"]
mid = slice(len(t) // 3, 2 * len(t) // 3)
for i, n in enumerate(names):
    print(f"{n:28s} median {np.median(d[mid, i]):7.2f} us  p90 {np.percentile(d[mid, i], 90):7.2f}")
print("item total median", np.median(t[mid, 7] - t[mid, 0]), "flags_ok->flag median", np.median(t[mid, 7] - t[mid, 2]))
