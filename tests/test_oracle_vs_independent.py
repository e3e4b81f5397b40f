"""The C oracle against independent re-derivations: scipy for the documented OpenCV contracts, and tests/pyref.py
(a second restatement of the reference sources in Python) for everything the reference itself defines.
PARITY UNPINNED: none of this is reference output -- the reference cannot be built in this image."""
import ctypes as C

import numpy as np
import pytest
import scipy.ndimage as ndi

import pyref
from oracle import pyoracle as po


def _tiny(w, h, D, seed):
    from crossscalepatchmatch_amd import synth
    l, r, _, _ = synth.make_pair(w, h, D, regions=2, seed=seed)
    return l, r


def test_pyrdown_matches_scipy_mirror_convolution():
    """pyrDown contract: separable [1 4 6 4 1]/16, BORDER_REFLECT_101 (= scipy 'mirror'), (v+128)>>8, ceil(size/2)."""
    rng = np.random.default_rng(4)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    for (h, w) in ((48, 64), (41, 77), (7, 5), (2, 3), (1, 9), (9, 1)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        out = np.zeros(((h + 1) // 2, (w + 1) // 2, 3), np.uint8)
        po.lib().csor_pyrdown_bgr8(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        t = ndi.convolve1d(img.astype(np.int64), k, axis=1, mode="mirror")
        t = ndi.convolve1d(t, k, axis=0, mode="mirror")
        want = ((t[::2, ::2] + 128) >> 8).astype(np.uint8)
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(out, pyref.pyrdown(img))


def test_gray_and_gradient():
    rng = np.random.default_rng(5)
    h, w = 9, 13
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    rgb = np.ascontiguousarray(bgr[..., ::-1].astype(np.float64))
    gray = np.zeros((h, w), np.float32)
    G = np.zeros((h, w))
    L = po.lib()
    L.csor_rgb2gray_f32(rgb.ctypes.data_as(C.POINTER(C.c_double)), w, h, gray.ctypes.data_as(C.POINTER(C.c_float)))
    L.csor_sobel_x_ks1(gray.ctypes.data_as(C.POINTER(C.c_float)), w, h, G.ctypes.data_as(C.POINTER(C.c_double)))
    np.testing.assert_array_equal(G, pyref.gray_grad(bgr))
    assert np.all(G[:, 0] == 0) and np.all(G[:, -1] == 0)  # REFLECT_101: zero at both borders
    # Sobel ksize=1 == correlate with [-1 0 1], mirror border
    np.testing.assert_array_equal(G, ndi.correlate1d(gray.astype(np.float64), [-1.0, 0.0, 1.0], axis=1, mode="mirror"))
    # a constant colour image has gray = that value (0.299f+0.587f+0.114f rounds to 1) and zero gradient
    flat = np.full((3, 4, 3), 200.0)
    g2 = np.zeros((3, 4), np.float32)
    L.csor_rgb2gray_f32(flat.ctypes.data_as(C.POINTER(C.c_double)), 4, 3, g2.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.all(np.abs(g2 - 200.0) < 1e-4)


@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (3, 0.3)])
def test_volumes_and_max_cost(scale_num, lam):
    l, r = _tiny(26, 15, 9, 1)
    pc = po.PlaneCost(l, r, 9, 5, scale_num, lam)
    ref = pyref.PlaneCost(l, r, 9, 5, scale_num, lam)
    for s in range(pc.levels):
        assert pc.dims(s) == ref.dims[s]
        for v in (0, 1):
            np.testing.assert_array_equal(pc.image(v, s), ref.img[v][s])
            np.testing.assert_array_equal(pc.volume(v, s), ref.vol[v][s])
            assert pc.max_cost(v, s) == ref.max_cost[v][s]
            # the device order's cells (last multiply-add contracted; pyref: exact rational fma): rounding-level difference only
            np.testing.assert_array_equal(pc.volume_dev(v, s), ref.vol_dev[v][s])
            assert pc.max_cost_dev(v, s) == ref.max_cost_dev[v][s]
            np.testing.assert_allclose(pc.volume_dev(v, s), pc.volume(v, s), rtol=4e-16, atol=0)
            assert np.any(pc.volume_dev(v, s) != pc.volume(v, s))
    # GRD cost is bounded by ALPHA*TAU_CLR + (1-ALPHA)*TAU_GRD = 2.8 and non-negative
    assert 0.0 <= pc.volume(0, 0).min() and pc.volume(0, 0).max() <= 0.1 * 10.0 + (1 - 0.1) * 2.0
    # border branch: left view, x < d uses the constant-3 "other" pixel
    lf = l[..., ::-1].astype(np.float64)
    G = pyref.gray_grad(l)
    d, y, x = 5, 3, 2
    clr = min((abs(lf[y, x, 0] - 3) + abs(lf[y, x, 1] - 3) + abs(lf[y, x, 2] - 3)) * 0.3333333333, 10.0)
    want = 0.1 * clr + (1 - 0.1) * min(abs(G[y, x] - 3), 2.0)
    assert pc.volume(0, 0)[d, y, x] == want


@pytest.mark.parametrize("wnd", [7, 9])
@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (3, 0.3), (5, 1.0)])
def test_get_plane_cost_serial_order(scale_num, lam, wnd):
    l, r = _tiny(30, 22, 10, 2)
    pc = po.PlaneCost(l, r, 10, wnd, scale_num, lam)
    ref = pyref.PlaneCost(l, r, 10, wnd, scale_num, lam)
    rng = np.random.default_rng(8)
    for i in range(60):
        x, y, v = int(rng.integers(0, 30)), int(rng.integers(0, 22)), int(rng.integers(0, 2))
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        if i == 0: n = np.array([0.0, 0.0, 1.0])
        if i == 1: n = np.array([0.6, 0.8, 1e-10]); x, y = 0, 0
        z = rng.uniform(-3, 14)
        prm = po.plane_param(n, [x, y, z])
        np.testing.assert_array_equal(prm, pyref.plane_param(n, [float(x), float(y), z]))
        got = pc.cost(x, y, n, prm, v, po.SUM_SERIAL)
        assert got == ref.cost(x, y, n, prm, v), (i, x, y, v)
        dev = pc.cost(x, y, n, prm, v, po.SUM_DEVICE)
        assert dev == ref.cost(x, y, n, prm, v, rowmod=7), (i, x, y, v)  # the device order, restated independently
        assert abs(dev - got) <= 1e-12 * max(1.0, abs(got))  # same terms, other summation order


def test_threshold_variant_is_result_preserving():
    l, r = _tiny(40, 30, 12, 3)
    pc = po.PlaneCost(l, r, 12, 35, 5, 0.3)
    rng = np.random.default_rng(9)
    for i in range(300):
        x, y, v = int(rng.integers(0, 40)), int(rng.integers(0, 30)), int(rng.integers(0, 2))
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        prm = po.plane_param(n, [x, y, rng.uniform(0, 12)])
        for order in (po.SUM_SERIAL, po.SUM_DEVICE):
            full = pc.cost(x, y, n, prm, v, order)
            for thr in (full * 0.5, full, np.nextafter(full, np.inf), full * 2):
                c, taps = pc.cost_thresh(x, y, n, prm, v, order, thr)
                if full >= thr:
                    assert c == np.inf and taps <= pc.taps(x, y)
                else:
                    assert c == full and taps == pc.taps(x, y)
    # exact tap count: interior pixel of a big-enough level has (2*17+1)^2 taps at level 0
    assert po.PlaneCost(l, r, 12, 35, 0, 0.0).taps(20, 17) == sum(1 for dy in range(-17, 18) for dx in range(-17, 18)
                                                                    if 0 <= 20 + dx < 40 and 0 <= 17 + dy < 30)


@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (2, 0.3)])
def test_whole_patchmatch_against_second_restatement(scale_num, lam):
    """init + 2 x (spatial raster, view, refine) + PlaneToDisp + PostProcessing on a 14x9 pair, window 5."""
    l, r = _tiny(14, 9, 6, 4)
    pc = po.PlaneCost(l, r, 6, 5, scale_num, lam)
    rpc = pyref.PlaneCost(l, r, 6, 5, scale_num, lam)
    pm = po.PatchMatch(l, r, 6, 16)
    ref = pyref.PatchMatch(l, r, 6, 16, seed=77)
    kw = dict(seed=77, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL)
    pm.init(pc, **kw); ref.init(rpc)
    for it in range(2):
        for phase in ("spatial", "view", "refine"):
            getattr(pm, phase)(it, pc, **kw); getattr(ref, phase)(it, rpc)
            for v in (0, 1):
                P = pm.planes(v)
                np.testing.assert_array_equal(P[..., 0:3], ref.n[v], err_msg=f"{phase} {it} norm")
                np.testing.assert_array_equal(P[..., 3:6], ref.p[v], err_msg=f"{phase} {it} point")
                np.testing.assert_array_equal(P[..., 6:9], ref.prm[v], err_msg=f"{phase} {it} param")
                np.testing.assert_array_equal(pm.min_cost(v), ref.cost[v], err_msg=f"{phase} {it} cost")
    pm.plane_to_disp(); ref.plane_to_disp()
    for v in (0, 1):
        np.testing.assert_array_equal(pm.dis(v), ref.dis[v])
    pm.postprocess(); ref.postprocess()
    for v in (0, 1):
        np.testing.assert_array_equal(pm.dis(v), ref.dis[v])


@pytest.mark.parametrize("kind", ["blocks", "saturated", "dup_rows", "periodic", "black", "identical", "stripes"])
def test_whole_patchmatch_on_adversarial_pairs_against_second_restatement(kind):
    """The tie-heavy / saturated pairs of synth.make_adversarial (round-4 review, item 1): the oracle's strict-`<` accept rules,
    view-propagation collisions and post-processing against the second restatement, reference (serial) order, 16x10, window 5,
    cross-scale with lambda 0.3 and with the CLI default lambda 0."""
    from crossscalepatchmatch_amd import synth
    l, r = synth.make_adversarial(kind, 16, 10, 6, seed=3)
    for scale_num, lam in ((2, 0.3), (2, 0.0)):
        pc = po.PlaneCost(l, r, 6, 5, scale_num, lam)
        rpc = pyref.PlaneCost(l, r, 6, 5, scale_num, lam)
        pm = po.PatchMatch(l, r, 6, 16)
        ref = pyref.PatchMatch(l, r, 6, 16, seed=78)
        kw = dict(seed=78, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL)
        pm.init(pc, **kw); ref.init(rpc)
        for it in range(2):
            for phase in ("spatial", "view", "refine"):
                getattr(pm, phase)(it, pc, **kw); getattr(ref, phase)(it, rpc)
                for v in (0, 1):
                    P = pm.planes(v)
                    np.testing.assert_array_equal(P[..., 6:9], ref.prm[v], err_msg=f"{kind} {phase} {it} param")
                    np.testing.assert_array_equal(pm.min_cost(v), ref.cost[v], err_msg=f"{kind} {phase} {it} cost")
        pm.plane_to_disp(); ref.plane_to_disp()
        for v in (0, 1):
            np.testing.assert_array_equal(pm.dis(v), ref.dis[v])
        pm.postprocess(); ref.postprocess()
        for v in (0, 1):
            np.testing.assert_array_equal(pm.dis(v), ref.dis[v])


@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (3, 0.3)])
def test_census_volumes_and_costs(scale_num, lam):
    """CenCC (cc/cen_cc.cc:4-137): 9x9 census with wrap-around borders, Hamming volumes, 80 outside; then GetPlaneCost."""
    l, r = _tiny(31, 23, 9, 6)
    pc = po.PlaneCost(l, r, 9, 7, scale_num, lam, cc="CEN")
    ref = pyref.PlaneCost(l, r, 9, 7, scale_num, lam, cc="CEN")
    for s in range(pc.levels):
        for v in (0, 1):
            np.testing.assert_array_equal(pc.volume(v, s), ref.vol[v][s])
            assert pc.max_cost(v, s) == ref.max_cost[v][s] == 80.0
    vol = pc.volume(0, 0)
    assert np.all(vol[3, :, :3] == 80.0) and np.all(vol == np.floor(vol)) and vol.min() >= 0 and vol.max() <= 80
    rng = np.random.default_rng(2)
    for i in range(40):
        x, y, v = int(rng.integers(0, 31)), int(rng.integers(0, 23)), int(rng.integers(0, 2))
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        prm = po.plane_param(n, [x, y, rng.uniform(0, 9)])
        assert pc.cost(x, y, n, prm, v, po.SUM_SERIAL) == ref.cost(x, y, n, prm, v)


def test_census_gray_contract():
    """8U RGB2GRAY of OpenCV 2.4: (R*4899 + G*9617 + B*1868 + 8192) >> 14; white -> 255, pure channels -> 76/150/29."""
    import ctypes as C
    px = np.array([[[255, 255, 255], [0, 0, 255], [0, 255, 0], [255, 0, 0], [10, 20, 30]]], np.uint8)  # BGR
    img = np.repeat(px, 9, axis=0)
    pc = po.PlaneCost(img, img, 2, 3, 0, 0.0, cc="CEN")
    g = ((px[..., 2].astype(int) * 4899 + px[..., 1].astype(int) * 9617 + px[..., 0].astype(int) * 1868 + 8192) >> 14)[0]
    assert list(g) == [255, 76, 150, 29, 22]  # e.g. (30*4899 + 20*9617 + 10*1868 + 8192) >> 14 = 22
    # identical views: zero cost wherever the other view is inside the image, 80 outside
    assert np.all(pc.volume(0, 0)[0] == 0) and np.all(pc.volume(0, 0)[1][:, 0] == 80)


@pytest.mark.parametrize("wnd", [5, 9])
@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (3, 0.3), (5, 1.0)])
def test_grdpc_cspc_plane_cost(scale_num, lam, wnd):
    """GrdPC (plane_cost/grd_pc.cc:72-176) / CSPC (cspc.cc:107-183): the volume-free variants -- the other view's colour and
    gradient interpolated at x -+ q_disp, wrap-around HandleBorder, constant "impossible disparity" cost."""
    l, r = _tiny(30, 22, 10, 12)
    pc = po.PlaneCost(l, r, 10, wnd, scale_num, lam, cc="IMG")
    ref = pyref.PlaneCost(l, r, 10, wnd, scale_num, lam, cc="IMG")
    assert pc.max_cost(0, 0) == ref.max_cost[0][0] == 0.1 * 10.0 + (1 - 0.1) * 2.0
    rng = np.random.default_rng(18)
    seen_valid = 0
    for i in range(80):
        x, y, v = int(rng.integers(0, 30)), int(rng.integers(0, 22)), int(rng.integers(0, 2))
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        if i < 20: n = np.array([0.0, 0.0, 1.0])  # fronto-parallel: every tap in range
        if i in (20, 21): x = 0 if v == 0 else 29  # other_x leaves the image: the wrap-around of HandleBorder
        z = rng.uniform(0.5, 9.5)
        prm = po.plane_param(n, [x, y, z])
        got = pc.cost(x, y, n, prm, v, po.SUM_SERIAL)
        assert got == ref.cost(x, y, n, prm, v), (i, x, y, v)
        dev = pc.cost(x, y, n, prm, v, po.SUM_DEVICE)
        assert dev == ref.cost(x, y, n, prm, v, rowmod=7), (i, x, y, v)
        assert abs(dev - got) <= 1e-12 * max(1.0, abs(got))
        seen_valid += got < 0.999 * pc.max_cost(0, 0) * pc.taps(x, y) * (1.0 if scale_num == 0 else 2.0)
    assert seen_valid > 20


def test_grdpc_matches_the_pixel_cost_of_a_true_shift():
    """A right image that is the left shifted by exactly 3 px: a fronto-parallel plane at d = 3.5 interpolates halfway between
    two columns (floor_wgt = 0.5) -- cost > 0; at d = 3 + 1e-9 the tap sits on the true match and every cell is ~0."""
    rng = np.random.default_rng(5)
    l = rng.integers(0, 256, (16, 40, 3), dtype=np.uint8)
    r = np.roll(l, -3, axis=1)
    pc = po.PlaneCost(l, r, 8, 5, 0, 0.0, cc="IMG")
    n = np.array([0.0, 0.0, 1.0])
    near = pc.cost(20, 8, n, po.plane_param(n, [20, 8, 3.000000001]), 0, po.SUM_SERIAL)
    half = pc.cost(20, 8, n, po.plane_param(n, [20, 8, 3.5]), 0, po.SUM_SERIAL)
    assert near < 1e-6 < half


def test_grdpc_whole_patchmatch_against_second_restatement():
    l, r = _tiny(14, 9, 6, 14)
    for scale_num, lam in ((0, 0.0), (2, 0.3)):
        pc = po.PlaneCost(l, r, 6, 5, scale_num, lam, cc="IMG")
        rpc = pyref.PlaneCost(l, r, 6, 5, scale_num, lam, cc="IMG")
        pm = po.PatchMatch(l, r, 6, 16)
        ref = pyref.PatchMatch(l, r, 6, 16, seed=31)
        kw = dict(seed=31, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL)
        pm.init(pc, **kw); ref.init(rpc)
        for phase in ("spatial", "view", "refine"):
            getattr(pm, phase)(0, pc, **kw); getattr(ref, phase)(0, rpc)
            for v in (0, 1):
                np.testing.assert_array_equal(pm.planes(v)[..., 6:9], ref.prm[v], err_msg=phase)
                np.testing.assert_array_equal(pm.min_cost(v), ref.cost[v], err_msg=phase)
