"""debug: per-item stage latencies and a per-STEP cycle budget of the persistent raster sweep (round-4 review, item 3).
Needs the -DCSPM_SWEEP_TRACE build:  make -C crossscalepatchmatch_amd/csrc ../libcspm_sweeptrace.so
  CSPM_TRACE_LIB=crossscalepatchmatch_amd/libcspm_sweeptrace.so CSPM_SWEEP_TRACE_FILE=/tmp/t.bin python tools/sweep_trace.py [sweep 1|2|3]
Per item (= pixel) 16 stamps: 0-7 wall clock (100 MHz) at the stages of the item; 8-15 shader clock (s_memtime) inside ONE chain step of
the level-0 wave (second pass, middle step): step start, own element arrived, guide weight read, other-view elements arrived, both cells
computed (colour-table round trip included), accumulated, and two stamps back to back (the cost of a stamp)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["CSPM_SWEEP_TRACE_SWEEP"] = str(which)
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth, capi
capi._SO = os.environ["CSPM_TRACE_LIB"]
cfg, l, r, _, _ = synth.make_config("C3")
ctx = cs.StereoContext(0)
ctx.set_images(l, r)
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
print("packed pixels:", ctx.get_option(capi.OPT_SWEEP_PACKED_ACTIVE), " sweep", which, "of a 3-iteration run")
ctx.patchmatch(3, seed=12345, schedule=0)
ctx.pm_spatial(3, seed=12345, schedule=0)  # the call that dumps sweep 3 (the dumps of sweeps 1 and 2 happened inside the run)
ctx.synchronize()
raw = np.fromfile(os.environ["CSPM_SWEEP_TRACE_FILE"], dtype=np.int64).reshape(-1, 16)
t = raw[:, :8].astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
print(f"sweep span {t[:, 7].max() - t0:.0f} us, {len(t)} items")
order = np.argsort(t[:, 0])
mid = np.zeros(len(t), bool); mid[order[len(t) // 3: 2 * len(t) // 3]] = True
two = (t[:, 4] > 0) & (t[:, 5] > 0)
names = [("claim -> decoded", 0, 1), ("decoded -> predecessors' planes here", 1, 2), ("planes -> candidates compared (shortcuts)", 2, 3),
         ("-> chain passes begin (level 0)", 3, 4), ("chain passes (level 0, 2 candidates)", 4, 5), ("passes done -> decided", 5, 6),
         ("candidates compared -> decided (any number of evaluations)", 3, 6), ("decided -> published", 6, 7), ("whole item", 0, 7)]
print("\nper item, middle third of the sweep (us):")
for n, a, b in names:
    m = mid & two if 4 in (a, b) or 5 in (a, b) else mid
    x = t[m, b] - t[m, a]
    print(f"  {n:58s} median {np.median(x):7.2f}  p10 {np.percentile(x, 10):7.2f}  p90 {np.percentile(x, 90):7.2f}  (n={int(m.sum())})")
ev = t[:, 6] - t[:, 3]
print(f"  items that evaluated nothing (both candidates are the pixel's own plane): {np.mean(ev[mid] < 0.3) * 100:.1f} %")
for name, idx in (("first 2000 items", order[:2000]), ("middle 2000", order[len(t) // 2: len(t) // 2 + 2000]), ("last 2000", order[-2000:])):
    x = t[idx, 6] - t[idx, 3]; w = t[idx, 2] - t[idx, 1]
    print(f"  {name:18s} compared->decided median {np.median(x):6.2f} us (p10 {np.percentile(x, 10):5.2f}, p90 {np.percentile(x, 90):5.2f}); wait for predecessors median {np.median(w):5.2f}")
# the diagonal period: consecutive anti-diagonals finish this far apart
W, H = cfg["w"], cfg["h"]
print(f"  sweep span / diagonals = {(t[:, 7].max() - t0) / (W + H - 1):.2f} us per anti-diagonal")
# per-step budget (shader cycles)
s = raw[:, 8:16].astype(np.float64)
have = (s[:, 0] > 0) & (s[:, 5] > 0)
for label, sel in (("middle third", have & mid), ("first 2000 items (GPU nearly empty)", have & np.isin(np.arange(len(t)), order[:6000]))):
    if sel.sum() < 10:
        continue
    d = np.diff(s[sel], axis=1)
    stamp = np.median(d[:, 6])
    print(f"\none chain step of the level-0 wave, {label}, shader cycles (n={int(sel.sum())}; a stamp itself costs {stamp:.0f}):")
    for k, n in enumerate(["step start -> own element in registers (address arithmetic, gather round trip)", "-> guide weight (v_sad_u8, exp-table LDS round trip)",
                           "-> other view's elements in registers (their gathers were issued with the own one)", "-> both cells (2 x sad, 2 x colour-table LDS round trip, sub/min/fma)",
                           "-> accumulated (interpolation, select, fma)"]):
        x = d[:, k]
        print(f"  {n:88s} median {np.median(x):7.0f}  p10 {np.percentile(x, 10):7.0f}  p90 {np.percentile(x, 90):7.0f}   minus stamp: {np.median(x) - stamp:7.0f}")
    tot = s[sel, 5] - s[sel, 0]
    print(f"  {'whole step':88s} median {np.median(tot):7.0f}  p10 {np.percentile(tot, 10):7.0f}  p90 {np.percentile(tot, 90):7.0f}   minus 5 stamps: {np.median(tot) - 5 * stamp:7.0f}")
