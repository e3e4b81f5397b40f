#!/usr/bin/env python
"""k_refine's weight gather (`lut.w[sad]`, 768 f64 entries in LDS, one ds_read_b64 per tap and wave) -- 31 % of its LDS cycles are bank
conflicts (profiles/refine_pmc.json).  Would another index -> slot layout of the table help (round-5 review, item 4b)?  Measured on
the CPU before anything is built: the gather pattern is a pure function of the image -- a wave holds 64 consecutive window centres
p of one row, and at window offset (dx, dy) lane l reads entry SAD(I(p_l), I(p_l + (dx, dy))).  For the level-0 images of a pair this
script walks sampled rows x all 35 x 35 offsets and reports, per gather: distinct entries among the 64 lanes, the SAD histogram, and the
LDS cycles under the bank model the kernel is written for (64 banks x 4 B = 256-byte rows, an f64 entry = one bank PAIR, a wave64
ds_read_b64 = two passes of 32 lanes; a pass costs max over bank pairs of the DISTINCT addresses that fall on it; same address =
broadcast) for three layouts: identity (slot = SAD, what the kernel has), frequency-ranked (the r-th most frequent SAD in slot r:
the 32 most frequent values on 32 different bank pairs), and the unreachable bound `ceil(distinct / 32)`.

    python tools/lut_conflict_sim.py [C3 | real]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crossscalepatchmatch_amd import realdata, synth  # noqa: E402


def gathers(img, rows, half=17):
    """yield int arrays [n_waves, 64] of SADs, one per (row, dy, dx)"""
    h, w, _ = img.shape
    I = img.astype(np.int32)
    nw = w // 64
    for y in rows:
        for dy in range(-half, half + 1):
            qy = y + dy
            if qy < 0 or qy >= h:
                continue
            for dx in range(-half, half + 1):
                x0 = max(0, -dx)
                x1 = min(w, w - dx)
                sad = np.full(w, -1, np.int32)
                sad[x0:x1] = np.abs(I[y, x0:x1] - I[qy, x0 + dx:x1 + dx]).sum(axis=1)
                yield sad[:nw * 64].reshape(nw, 64)


def pass_cycles(slots):
    """slots [n, 32] -> cycles of one 32-lane pass: max over bank pairs of distinct slots on it (slot < 0 = lane masked)"""
    n = slots.shape[0]
    cyc = np.zeros(n, np.int32)
    s = np.sort(slots, axis=1)
    first = np.ones_like(s, bool)
    first[:, 1:] = s[:, 1:] != s[:, :-1]
    first &= s >= 0
    bank = s % 32
    for b in range(32):
        cyc = np.maximum(cyc, ((bank == b) & first).sum(axis=1))
    return np.maximum(cyc, 1)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "C3"
    if which == "real":
        item = realdata.load_full() or realdata.load_crop()
        img, name = item[1], f"Middlebury motorcycle {item[1].shape[1]}x{item[1].shape[0]}"
    else:
        cfg, img, _, _, _ = synth.make_config(which)
        name = f"{which} synthetic {cfg['w']}x{cfg['h']}"
    h = img.shape[0]
    rows = list(range(20, h - 20, max(1, (h - 40) // 6)))[:6]
    all_g = np.concatenate(list(gathers(img, rows)))
    valid = all_g >= 0
    hist = np.bincount(all_g[valid], minlength=766)
    rank = np.empty(766, np.int64)
    rank[np.argsort(-hist, kind="stable")] = np.arange(766)
    full = valid.all(axis=1)
    g = all_g[full]
    s = np.sort(g, axis=1)
    distinct = 1 + (s[:, 1:] != s[:, :-1]).sum(axis=1)
    out = [f"{name}: {len(g)} weight gathers of full waves (rows {rows}, 35 x 35 offsets), level 0"]
    cdf = np.cumsum(hist) / hist.sum()
    out.append("SAD distribution: median %d, 90 %% <= %d, 99 %% <= %d; the 32 most frequent values cover %.1f %%, the 48 most frequent %.1f %% of the reads; "
               "they span SAD %d..%d" % (np.searchsorted(cdf, 0.5), np.searchsorted(cdf, 0.9), np.searchsorted(cdf, 0.99), 100 * np.sort(hist)[::-1][:32].sum() / hist.sum(),
                                          100 * np.sort(hist)[::-1][:48].sum() / hist.sum(), np.argsort(-hist)[:48].min(), np.argsort(-hist)[:48].max()))
    out.append("distinct entries among the 64 lanes of a gather: mean %.1f; <= 8: %.1f %%, <= 16: %.1f %%, <= 32: %.1f %%, > 48: %.1f %%" %
               (distinct.mean(), 100 * np.mean(distinct <= 8), 100 * np.mean(distinct <= 16), 100 * np.mean(distinct <= 32), 100 * np.mean(distinct > 48)))
    for label, slots in (("identity (slot = SAD)", g), ("frequency-ranked slots", rank[g])):
        cyc = pass_cycles(slots[:, :32]) + pass_cycles(slots[:, 32:])
        out.append(f"  {label:24s}: {cyc.mean():.2f} LDS cycles per gather (2 = conflict-free), conflict share (cycles - 2) / cycles = {100 * (cyc - 2).sum() / cyc.sum():.1f} %")
    d0 = 1 + (np.sort(g[:, :32], axis=1)[:, 1:] != np.sort(g[:, :32], axis=1)[:, :-1]).sum(axis=1)
    d1 = 1 + (np.sort(g[:, 32:], axis=1)[:, 1:] != np.sort(g[:, 32:], axis=1)[:, :-1]).sum(axis=1)
    out.append(f"  mean distinct entries per 32-lane pass: {(d0.mean() + d1.mean()) / 2:.1f} -- balls-in-bins: that many random entries over 32 bank pairs")
    print("\n".join(out))


if __name__ == "__main__":
    main()
