"""Real photographs for the tests and for bench.py's accuracy key: the Middlebury-2014 "Motorcycle" pair (Scharstein et al.,
GCPR 2014) that scikit-image ships in this image, and the half-size crop of it committed under tests/data/ (made by
tests/data/make_real_pair.py).  The reference ships no images; BASELINE.json's configs name Middlebury and KITTI pairs and the
reference's only entry point reads PNGs (main.cc:68-69).  Inputs only -- PIL decodes here, the product decodes with
host/image_io.cc (which tests/test_gpu_realpair.py holds against PIL pixel for pixel).
"""
import os

import numpy as np

SKIMAGE_DATA = os.environ.get("CSPM_SKIMAGE_DATA", "/opt/conda/lib/python3.9/site-packages/skimage/data")
CROP_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data")

# the flags the full-size pair is run with (ground truth 7.2 .. 59.9 px): main.cc's flags, D = 64, 8-bit maps in 1/4 px
FULL = dict(w=741, h=500, max_dis=64, dis_scale=4, scale_num=5, reg_lambda=0.3)
HALF = dict(w=370, h=250, max_dis=32, dis_scale=8, scale_num=5, reg_lambda=0.3)
CROP = dict(w=200, h=128, max_dis=32, dis_scale=8, scale_num=5, reg_lambda=0.3)


def full_files():
    """(left.png, right.png, disp.npz) of the 741x500 pair, or None when scikit-image's data directory is not on this machine"""
    f = [os.path.join(SKIMAGE_DATA, n) for n in ("motorcycle_left.png", "motorcycle_right.png", "motorcycle_disp.npz")]
    return f if all(os.path.exists(p) for p in f) else None


def crop_files():
    return [os.path.join(CROP_DIR, n) for n in ("motorcycle_half_crop_left.png", "motorcycle_half_crop_right.png", "motorcycle_half_crop_gt_x256.png")]


def _bgr(path_or_image):
    from PIL import Image
    im = path_or_image if hasattr(path_or_image, "convert") else Image.open(path_or_image)
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])  # cv::imread(CV_LOAD_IMAGE_COLOR) gives BGR


def load_crop():
    """(cfg, left_bgr, right_bgr, gt_left [NaN = unknown]) of the committed 200x128 half-size crop"""
    from PIL import Image
    lf, rf, gf = crop_files()
    gt = np.asarray(Image.open(gf)).astype(np.float64) / 256.0
    gt[gt == 0] = np.nan
    return dict(CROP), _bgr(lf), _bgr(rf), gt


def load_full():
    """(cfg, left_bgr, right_bgr, gt_left [NaN = unknown]) of the 741x500 pair; None when it is not on this machine"""
    f = full_files()
    if f is None:
        return None
    gt = np.load(f[2])["arr_0"].astype(np.float64)
    gt[~np.isfinite(gt)] = np.nan
    return dict(FULL), _bgr(f[0]), _bgr(f[1]), gt


def load_half():
    """the 741x500 pair at half size (2x2 box mean, PIL's BOX filter -> 370x250), ground truth = median of the known values of each
    2x2 block, halved: the size SURVEY.md 8(c) quotes the unmodified reference's bad-2.0 on (10.9-11.1 %, D = 32)"""
    f = full_files()
    if f is None:
        return None
    import warnings
    from PIL import Image
    out = []
    for p in f[:2]:
        im = Image.open(p).convert("RGB")
        out.append(_bgr(im.resize((im.width // 2, im.height // 2), Image.BOX)))
    g = np.load(f[2])["arr_0"].astype(np.float64)
    g[~np.isfinite(g)] = np.nan
    h, w = g.shape[0] // 2 * 2, g.shape[1] // 2 * 2
    b = g[:h, :w].reshape(h // 2, 2, w // 2, 2).transpose(0, 2, 1, 3).reshape(h // 2, w // 2, 4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gt = np.nanmedian(b, axis=2) / 2.0
    return dict(HALF), out[0], out[1], gt


def bad_fraction(disp, gt, thresh=2.0):
    """fraction of the pixels with known ground truth whose disparity is off by more than `thresh` px"""
    m = np.isfinite(gt)
    return float(np.mean(np.abs(disp[m] - gt[m]) > thresh)) if m.any() else float("nan")
