"""Seeded synthetic stereo pairs: the reference ships no images and there is no
dataset in this image.  Left texture = band-limited noise (24 random sinusoids per channel, amplitude 40,
mean 128) + iid Gaussian sigma 6; ground-truth disparity = Voronoi partition into slanted planes
(d in [0.08, 0.85]*max_dis, |slopes| <= 0.04); right image = the left image sampled at x + d_R(x, y) with
linear interpolation + independent noise sigma 2.  numpy only; inputs, not part of any parity claim.
"""
import numpy as np

# BASELINE.json configs: name -> (W, H, max_dis, dis_scale, scale_num, reg_lambda, regions, seed)
CONFIGS = {
    "C1": dict(w=450, h=375, max_dis=60, dis_scale=4, scale_num=0, reg_lambda=0.0, regions=8, seed=1001, use_pp=False),
    "C2": dict(w=450, h=375, max_dis=60, dis_scale=4, scale_num=5, reg_lambda=0.3, regions=8, seed=1002, use_pp=False),
    "C3": dict(w=1242, h=375, max_dis=128, dis_scale=1, scale_num=5, reg_lambda=0.3, regions=12, seed=2000, use_pp=False),
    # C4 = 200 pairs of C3's shape, seeds 2000 + i (make_config("C4", index=i) == make_config("C3", index=i))
    "C4": dict(w=1242, h=375, max_dis=128, dis_scale=1, scale_num=5, reg_lambda=0.3, regions=12, seed=2000, use_pp=False),
    "C5": dict(w=3000, h=2000, max_dis=256, dis_scale=1, scale_num=5, reg_lambda=0.3, regions=24, seed=3001, use_pp=True),
}


def make_pair(w, h, max_dis, regions=8, seed=0):
    """returns (left_bgr u8 [h,w,3], right_bgr u8, gt_left f64 [h,w] (NaN = occluded), gt_right f64)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    left = np.zeros((h, w, 3))
    for c in range(3):
        acc = np.zeros((h, w))
        for _ in range(24):
            fx, fy = rng.uniform(-0.35, 0.35, 2)
            ph = rng.uniform(0, 2 * np.pi)
            acc += np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
        left[..., c] = 128.0 + 40.0 * acc / np.sqrt(12.0)
    left += rng.normal(0.0, 6.0, left.shape)
    left_u8 = np.clip(np.rint(left), 0, 255).astype(np.uint8)
    # ground truth in the right view: Voronoi cells, each a slanted plane
    sx, sy = rng.uniform(0, w, regions), rng.uniform(0, h, regions)
    cell = np.argmin((xx[None] - sx[:, None, None]) ** 2 + (yy[None] - sy[:, None, None]) ** 2, axis=0)
    d0 = rng.uniform(0.08, 0.85, regions) * max_dis
    a, b = rng.uniform(-0.04, 0.04, regions), rng.uniform(-0.04, 0.04, regions)
    gt_r = d0[cell] + a[cell] * (xx - sx[cell]) + b[cell] * (yy - sy[cell])
    gt_r = np.clip(gt_r, 1.0, max_dis - 2.0)
    # right(x) = left(x + d_R(x)), linear interpolation, edge clamp
    xs = np.clip(xx + gt_r, 0, w - 1)
    x0 = np.floor(xs).astype(np.int64)
    x1 = np.minimum(x0 + 1, w - 1)
    t = (xs - x0)[..., None]
    rows = np.arange(h)[:, None]
    lf = left_u8.astype(np.float64)
    right = (1 - t) * lf[rows, x0] + t * lf[rows, x1]
    right += rng.normal(0.0, 2.0, right.shape)
    right_u8 = np.clip(np.rint(right), 0, 255).astype(np.uint8)
    # left-view ground truth by forward splatting (nearest column, larger disparity wins)
    gt_l = np.full((h, w), np.nan)
    tx = np.rint(xx + gt_r).astype(np.int64)
    ok = (tx >= 0) & (tx < w)
    order = np.argsort(gt_r, axis=1)
    for y in range(h):
        o = order[y]
        m = ok[y, o]
        gt_l[y, tx[y, o][m]] = gt_r[y, o][m]
    return left_u8, right_u8, gt_l, gt_r


# Pairs the noise textures of make_pair never produce: exact ties between candidate planes, cells that are exactly 0, cells that
# all saturate, truncation thresholds hit exactly.  The reference's accept rules are strict `<` (cs_patchmatch.cc:182,192,201,209,
# 270,335), so on such inputs the traversal ORDER decides -- which is what a parallel implementation can get wrong.
ADVERSARIAL_KINDS = ("blocks", "saturated", "dup_rows", "periodic", "black", "white", "identical", "stripes", "half_flat")


def _shift_right_view(left, d):
    """right(x) = left(x + d) with the last column repeated (integer disparity d everywhere)"""
    h, w = left.shape[:2]
    xs = np.minimum(np.arange(w) + int(d), w - 1)
    return np.ascontiguousarray(left[:, xs])


def make_adversarial(kind, w, h, max_dis, seed=0):
    """returns (left_bgr u8 [h,w,3], right_bgr u8): tie-heavy / saturated / degenerate stereo pairs, no ground truth.
    blocks:     constant-colour rectangles (w/5 x h/4), right = left shifted by max_dis/3 -- every plane inside a block costs the same
    saturated:  the same rectangles with channels drawn from {0, 255} only -- |colour difference| and gradient both truncate
    dup_rows:   a noise texture whose rows 2k and 2k+1 are equal in both views -- the up and down neighbours tie
    periodic:   a texture of horizontal period 6 (< max_dis): disparities d and d + 6 cost the same
    black/white: constant 0 / 255 in both views -- every interior cell is exactly 0, min_cost == 0, no candidate is ever `<`
    identical:  L == R (a noise texture): the true disparity 0 is the reference's "impossible" disparity (pre_cs_pc.cc:166)
    stripes:    vertical 0/255 stripes 4 px wide, right shifted by 2: the gradient term saturates everywhere, colour ties
    half_flat:  left half constant 128, right half noise -- flat and textured windows in one wavefront"""
    rng = np.random.default_rng(seed)
    if kind in ("blocks", "saturated"):
        bw, bh = max(w // 5, 1), max(h // 4, 1)
        ny, nx = -(-h // bh), -(-w // bw)
        if kind == "blocks":
            cols = rng.integers(0, 256, (ny, nx, 3))
        else:
            cols = rng.choice([0, 255], (ny, nx, 3))
        left = np.repeat(np.repeat(cols, bh, 0), bw, 1)[:h, :w].astype(np.uint8)
        return left, _shift_right_view(left, max(max_dis // 3, 1))
    if kind == "dup_rows":
        half = rng.integers(0, 256, (-(-h // 2), w, 3))
        left = np.repeat(half, 2, 0)[:h].astype(np.uint8)
        return left, _shift_right_view(left, max(max_dis // 4, 1))
    if kind == "periodic":
        tile = rng.integers(0, 256, (h, 6, 3))
        left = np.tile(tile, (1, -(-w // 6), 1))[:, :w].astype(np.uint8)
        return left, np.ascontiguousarray(np.roll(left, -2, axis=1))
    if kind in ("black", "white"):
        img = np.full((h, w, 3), 0 if kind == "black" else 255, np.uint8)
        return img, img.copy()
    if kind == "identical":
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        return img, img.copy()
    if kind == "stripes":
        col = (((np.arange(w) // 4) % 2) * 255).astype(np.uint8)
        left = np.broadcast_to(col[None, :, None], (h, w, 3)).copy()
        return left, _shift_right_view(left, 2)
    if kind == "half_flat":
        left = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        left[:, : w // 2] = 128
        return left, _shift_right_view(left, max(max_dis // 2, 1))
    raise ValueError(f"unknown adversarial kind {kind!r}; one of {ADVERSARIAL_KINDS}")


def make_config(name, index=0):
    cfg = dict(CONFIGS[name])
    l, r, gl, gr = make_pair(cfg["w"], cfg["h"], cfg["max_dis"], cfg["regions"], cfg["seed"] + index)
    return cfg, l, r, gl, gr


def bad_fraction(disp, gt, thresh):
    m = np.isfinite(gt)
    return float(np.mean(np.abs(disp[m] - gt[m]) > thresh)) if m.any() else float("nan")
