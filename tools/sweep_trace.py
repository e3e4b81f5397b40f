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
ctx_flow = bool(ctx.get_option(capi.OPT_SWEEP_FLOW))
print("packed pixels:", ctx.get_option(capi.OPT_SWEEP_PACKED_ACTIVE), " dataflow scheduling:", ctx_flow, " sweep", which, "of a 3-iteration run")
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
# how many items are in progress at once (= resident workgroups that hold an item), sampled over the middle of the sweep
lo, hi = np.percentile(t[:, 0], 35), np.percentile(t[:, 0], 65)
starts, ends = np.sort(t[:, 0]), np.sort(t[:, 7])
samples = np.linspace(lo, hi, 200)
inflight = np.searchsorted(starts, samples, side="right") - np.searchsorted(ends, samples, side="right")
print(f"  items in progress at once, middle of the sweep: median {np.median(inflight):.0f}, max {inflight.max()}  (CSPM_SWEEP_WG={os.environ.get('CSPM_SWEEP_WG', '2')} x 256 CUs requested)")
# placement of a workgroup's waves on the CU's SIMDs (HW_ID: bits 5:4 = SIMD, 11:8 = CU), and how many workgroups share a CU
hw = raw[:, 14]
if (hw != 0).any():
    import collections
    simd = np.stack([(hw >> (12 * k + 4)) & 3 for k in range(5)], 1)
    pat = collections.Counter(tuple(np.bincount(row, minlength=4)) for row in simd[order[len(t) // 2: len(t) // 2 + 20000]])
    print("  waves of a workgroup per SIMD (SIMD0..3), middle of the sweep:", ", ".join(f"{k}: {v}" for k, v in pat.most_common(6)))
    seq = collections.Counter(tuple(row) for row in simd[order[len(t) // 2: len(t) // 2 + 20000]])
    print("  SIMD of waves 0..4 (levels 0..4):", ", ".join(f"{k}: {v}" for k, v in seq.most_common(6)))
    cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | ((raw[:, 15] & 7) << 5) | (((raw[:, 15] >> 4) & 15) << 8)   # cu, sh, se, xcc
    mids = order[len(t) // 2: len(t) // 2 + 478]
    per_cu = collections.Counter(cu[mids])
    print(f"  distinct CUs among 478 consecutive claims: {len(per_cu)}; items per CU: {collections.Counter(per_cu.values())}")
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

# ---- the critical path through the dependency lattice, walked back from the last pixel of one view --------------------------------
# item index = claim order (one band): diagonal k, then view, then sweep row.  Sweeps 1 and 3 run top-left -> bottom-right (inc > 0),
# sweep 2 the other way; the lattice is the same in sweep coordinates (xs, ys).
W, H = cfg["w"], cfg["h"]
idx = np.full((2, H, W), -1, np.int64)
if ctx_flow:  # k_spatial_flow stamps by pixel: view * W * H + y * W + x, in IMAGE coordinates; the lattice below is in sweep coordinates
    img = np.arange(2 * W * H, dtype=np.int64).reshape(2, H, W)
    idx = img if which % 2 == 1 else img[:, ::-1, ::-1]
else:
    base = 0
    for k in range(W + H - 1):
        lo_, hi_ = max(0, k - (W - 1)), min(H - 1, k)
        cnt = hi_ - lo_ + 1
        ys = np.arange(lo_, hi_ + 1)
        for v in (0, 1):
            idx[v, ys, k - ys] = base + v * cnt + (ys - lo_)
        base += 2 * cnt
    assert base == len(t)
for v in (0, 1):
    seg = {"successor not claimed yet when its predecessor published (-> decoded)": 0.0, "hand-over (predecessor published -> its plane is here)": 0.0, "planes -> chain passes begin (compare, plane_param)": 0.0,
           "evaluation (chain passes + finish, or nothing)": 0.0, "decided -> published": 0.0}
    xs_, ys_ = W - 1, H - 1
    n2 = n1 = n0 = 0
    end = t[idx[v, ys_, xs_], 7]
    while xs_ > 0 or ys_ > 0:
        i = idx[v, ys_, xs_]
        cands = []
        if xs_ > 0: cands.append((t[idx[v, ys_, xs_ - 1], 7], xs_ - 1, ys_))
        if ys_ > 0: cands.append((t[idx[v, ys_ - 1, xs_], 7], xs_, ys_ - 1))
        tp, px_, py_ = max(cands)
        late = max(0.0, t[i, 1] - tp)  # the successor's workgroup was still busy with (or had not yet claimed) something else
        seg["successor not claimed yet when its predecessor published (-> decoded)"] += late
        seg["hand-over (predecessor published -> its plane is here)"] += t[i, 2] - tp - late
        seg["planes -> chain passes begin (compare, plane_param)"] += t[i, 3] - t[i, 2]
        seg["evaluation (chain passes + finish, or nothing)"] += t[i, 6] - t[i, 3]
        seg["decided -> published"] += t[i, 7] - t[i, 6]
        ev_ = t[i, 6] - t[i, 3]
        if t[i, 4] > 0: n2 += 1
        elif ev_ > 1.0: n1 += 1
        else: n0 += 1
        xs_, ys_ = px_, py_
    tot = sum(seg.values())
    print(f"\ncritical path of view {v}: {W + H - 2} pixels, {tot:.0f} us (sweep span {t[:, 7].max() - t0:.0f}); two-candidate pixels on it {n2}, one-candidate {n1}, none {n0}")
    for k_, x_ in seg.items():
        print(f"  {k_:62s} {x_:8.0f} us = {x_ / (W + H - 2):6.2f} us per pixel ({100 * x_ / tot:4.1f} %)")
