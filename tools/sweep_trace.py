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
mid = slice(len(t) // 3, 2 * len(t) // 3)
# slots: 0 claimed, 1 decoded, 2 flags ok, 3 planes loaded, 4 level-0 wave: tables filled, 5 level-0 wave: chain passes done (both 0 when
# the pixel evaluated fewer than two candidates), 6 decided + stored, 7 flag raised
names = [("claim->decoded", 0, 1), ("decoded->flags ok", 1, 2), ("flags->planes loaded", 2, 3), ("planes->tables filled (level 0)", 3, 4),
         ("tables->chain passes done (level 0)", 4, 5), ("passes done->decided+stored", 5, 6), ("planes->decided+stored", 3, 6), ("stored->flag", 6, 7)]
two = (t[:, 4] > 0) & (t[:, 5] > 0)
sel = np.zeros(len(t), bool); sel[mid] = True
for n, a, b in names:
    m = sel & two if 4 in (a, b) or 5 in (a, b) else sel
    x = t[m, b] - t[m, a]
    print(f"{n:38s} median {np.median(x):7.2f} us  p90 {np.percentile(x, 90):7.2f}  (n={m.sum()})")
print("item total median", np.median(t[mid, 7] - t[mid, 0]), "flags_ok->flag median", np.median(t[mid, 7] - t[mid, 2]))

# contention check: the evaluation time of items early in the sweep (few pixels in flight) against the middle (all CUs busy)
order = np.argsort(t[:, 0])
for name, idx in (("first 2000 items", order[:2000]), ("items 20000-22000", order[20000:22000]), ("middle 2000", order[len(t) // 2: len(t) // 2 + 2000]), ("last 2000", order[-2000:])):
    x = t[idx, 6] - t[idx, 3]
    w = t[idx, 2] - t[idx, 1]
    print(f"{name:20s} planes->decided median {np.median(x):6.2f} us p10 {np.percentile(x, 10):6.2f} p90 {np.percentile(x, 90):6.2f};  flag wait median {np.median(w):6.2f}")
