"""Cuts the committed real-image fixture out of the Middlebury-2014 "Motorcycle" pair that scikit-image ships
(skimage/data/motorcycle_{left,right}.png + motorcycle_disp.npz: 741x500 RGB, ground truth of the left view, inf = unknown).
It is Middlebury data (Scharstein et al., GCPR 2014), not the reference's: rookiepig/CrossScalePatchMatch ships no images.

Output (tests/data/, ~135 KB together): the pair at HALF size (2x2 box mean, as PIL's Image.BOX resize rounds it), columns
[80, 280) x rows [70, 198) of it -- tank, engine and the occlusion edge of the front fork against the background -- as two
8-bit RGB PNGs written by PIL (a foreign encoder for host/image_io.cc), and the ground truth of the same window at half
size (median of the finite values of each 2x2 block, halved; 0 = unknown) as a 16-bit PNG in 1/256 px.

    python tests/data/make_real_pair.py [directory with the skimage files]
"""
import os
import sys

import numpy as np
from PIL import Image

SRC = "/opt/conda/lib/python3.9/site-packages/skimage/data"
X0, Y0, W, H = 80, 70, 200, 128


def half_size_gt(g):
    h, w = g.shape[0] // 2 * 2, g.shape[1] // 2 * 2
    b = g[:h, :w].reshape(h // 2, 2, w // 2, 2).transpose(0, 2, 1, 3).reshape(h // 2, w // 2, 4).astype(np.float64)
    b[~np.isfinite(b)] = np.nan
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # blocks that are unknown altogether
        m = np.nanmedian(b, axis=2) / 2.0
    return np.where(np.isfinite(m), m, 0.0)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else SRC
    out = os.path.dirname(os.path.abspath(__file__))
    for side in ("left", "right"):
        im = Image.open(os.path.join(src, f"motorcycle_{side}.png")).convert("RGB")
        half = im.resize((im.width // 2, im.height // 2), Image.BOX)
        half.crop((X0, Y0, X0 + W, Y0 + H)).save(os.path.join(out, f"motorcycle_half_crop_{side}.png"), optimize=True)
    gt = half_size_gt(np.load(os.path.join(src, "motorcycle_disp.npz"))["arr_0"])[Y0:Y0 + H, X0:X0 + W]
    q = np.clip(np.rint(gt * 256.0), 0, 65535).astype(np.uint16)
    Image.fromarray(q).save(os.path.join(out, "motorcycle_half_crop_gt_x256.png"), optimize=True)
    print("wrote", W, "x", H, "gt range", float(gt[gt > 0].min()), float(gt.max()), "unknown", float(np.mean(gt == 0)))


if __name__ == "__main__":
    main()
