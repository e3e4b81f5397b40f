"""GPU parity, cost side: pyramid (pre_cs_pc.cc:36-55), GRD volumes (cc/grd_cc.cpp:60-154), max_cost,
scale weights, and batched IPlaneCost::GetPlaneCost (pre_ss_pc.cc:74-118, pre_cs_pc.cc:133-188) --
HIP path through the C ABI vs the oracle, bit-exact (f64 arithmetic, same operation order)."""
import numpy as np
import pytest

from conftest import random_planes
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

CASES = [("ss", 0, 0.0), ("cs0", 5, 0.0), ("cs03", 5, 0.3), ("cs1", 5, 1.0), ("cs3lv", 3, 0.3)]


def _build(ctx, pair, scale_num, lam, wnd=35, volumes=False):
    ctx.set_images(pair["l"], pair["r"])
    ctx.build_cost_grd(pair["max_dis"], wnd, scale_num, lam, volumes=volumes)
    return po.PlaneCost(pair["l"], pair["r"], pair["max_dis"], wnd, scale_num, lam)


@pytest.mark.parametrize("volumes", [False, True], ids=["fused", "volumes"])
@pytest.mark.parametrize("pairname", ["small_pair", "odd_pair"])
@pytest.mark.parametrize("name,scale_num,lam", CASES)
def test_pyramid_volumes_bit_exact(gpu_ctx, request, pairname, name, scale_num, lam, volumes):
    pair = request.getfixturevalue(pairname)
    pc = _build(gpu_ctx, pair, scale_num, lam, volumes=volumes)
    assert gpu_ctx.levels == pc.levels
    np.testing.assert_array_equal(gpu_ctx.scale_weights(), pc.scale_wgt())
    for s in range(pc.levels):
        assert gpu_ctx.level_dims(s) == pc.dims(s)
        for v in (0, 1):
            np.testing.assert_array_equal(gpu_ctx.level_image(v, s), pc.image(v, s))
            # the cells the plane cost reads on the device: GRD cells of the device order (last multiply-add contracted)
            np.testing.assert_array_equal(gpu_ctx.cost_volume(v, s), pc.volume_dev(v, s))
            assert gpu_ctx.max_cost(v, s) == pc.max_cost_dev(v, s)


def test_grd_build_cv_host_boundary(small_pair):
    """CCMethod::buildCV / buildRightCV on caller-owned CV_64FC3 buffers (cc_method.h:31-32)."""
    import ctypes as C
    import crossscalepatchmatch_amd as cs
    L = cs.load_library()
    lib = po.lib()
    h, w, D = small_pair["h"], small_pair["w"], small_pair["max_dis"] + 1
    rng = np.random.default_rng(5)
    # arbitrary doubles, not only u8-valued ones: the boundary takes CV_64FC3
    l = small_pair["l"][..., ::-1].astype(np.float64) + rng.uniform(-0.4, 0.4, (h, w, 3))
    r = small_pair["r"][..., ::-1].astype(np.float64) + rng.uniform(-0.4, 0.4, (h, w, 3))
    l, r = np.ascontiguousarray(l), np.ascontiguousarray(r)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for right in (0, 1):
        got = np.zeros((D, h, w))
        want = np.zeros((D, h, w))
        rc = L.cspm_grd_build_cv_host(0, dp(l), dp(r), w, h, D, right, dp(got))
        assert rc == 0, L.cspm_last_error(None)
        (lib.csor_grd_build_right_cv if right else lib.csor_grd_build_cv)(dp(l), dp(r), w, h, D, dp(want))
        np.testing.assert_array_equal(got, want)


def test_reference_cells_at_the_pyramid_levels(gpu_ctx, odd_pair):
    """The volumes the device keeps (cspm_get_cost_slab, CSPM_OPT_GRD_VOLUMES, max_cost) hold the DEVICE cells -- myCostGrd with its last
    multiply-add contracted -- and differ from GrdCC::buildCV's by <= 1 ulp.  The reference's own cells stay reachable at every
    pyramid level through the CCMethod::buildCV boundary: the level images the device built (pyrDown chain) fed to
    cspm_grd_build_cv_host give the oracle's reference-arithmetic volumes (pc.volume) bit for bit, and the device cells of the same
    level (pc.volume_dev == cspm_get_cost_slab) sit within one ulp of them."""
    import ctypes as C
    import crossscalepatchmatch_amd as cs
    L = cs.load_library()
    pc = _build(gpu_ctx, odd_pair, 3, 0.3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for s in range(3):
        w, h, D = gpu_ctx.level_dims(s)
        rgb = [np.ascontiguousarray(gpu_ctx.level_image(v, s)[..., ::-1].astype(np.float64)) for v in (0, 1)]  # CV_64FC3 RGB, as PreCSPC hands them on (pre_cs_pc.cc:61-64)
        for right in (0, 1):
            got = np.zeros((D + 1, h, w))
            assert L.cspm_grd_build_cv_host(0, dp(rgb[0]), dp(rgb[1]), w, h, D + 1, right, dp(got)) == 0, L.cspm_last_error(None)
            ref = pc.volume(right, s)
            np.testing.assert_array_equal(got, ref, err_msg=f"reference cells, level {s}, view {right}")
            dev = gpu_ctx.cost_volume(right, s)
            np.testing.assert_array_equal(dev, pc.volume_dev(right, s))
            assert np.max(np.abs(dev - ref) / np.spacing(np.maximum(np.abs(ref), 1e-300))) <= 1.0


@pytest.mark.parametrize("volumes", [False, True], ids=["fused", "volumes"])
@pytest.mark.parametrize("pairname", ["small_pair", "odd_pair"])
@pytest.mark.parametrize("name,scale_num,lam", CASES)
def test_plane_cost_batch(gpu_ctx, request, pairname, name, scale_num, lam, volumes):
    """T2: >= 10^4 random (x, y, plane, view) tuples incl. corners, |nz|~0, out-of-range disparities."""
    pair = request.getfixturevalue(pairname)
    pc = _build(gpu_ctx, pair, scale_num, lam, volumes=volumes)
    rng = np.random.default_rng(99)
    n = 5200
    for view in (0, 1):
        xy, norm, point, param = random_planes(rng, n, pair["w"], pair["h"], pair["max_dis"])
        got = gpu_ctx.plane_cost_batch(view, xy, np.concatenate([norm, param], 1))
        lane = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], view, po.SUM_DEVICE) for i in range(n)])
        np.testing.assert_array_equal(got, lane)  # same summation order: bit-exact
        idx = rng.choice(n, 600, replace=False)
        idx[:14] = np.arange(14)
        ser = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], view, po.SUM_SERIAL) for i in idx])
        np.testing.assert_allclose(got[idx], ser, rtol=1e-12, atol=0)  # reference order: rounding only


def test_small_window(gpu_ctx, small_pair):
    """wnd_size is a constructor argument (pre_ss_pc.h:20-22); 35 is only main.cc's constant."""
    for wnd in (1, 3, 9, 35, 41, 45):
        pc = _build(gpu_ctx, small_pair, 3, 0.3, wnd)
        rng = np.random.default_rng(wnd)
        xy, norm, point, param = random_planes(rng, 64, small_pair["w"], small_pair["h"], small_pair["max_dis"])
        got = gpu_ctx.plane_cost_batch(0, xy, np.concatenate([norm, param], 1))
        want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], 0, po.SUM_DEVICE) for i in range(64)])
        np.testing.assert_array_equal(got, want)


def test_foreign_cost_volume_upload(gpu_ctx, small_pair):
    """A CCMethod plugin the library knows nothing about: host volumes uploaded slab by slab."""
    pc = po.PlaneCost(small_pair["l"], small_pair["r"], small_pair["max_dis"], 35, 3, 0.3)
    rng = np.random.default_rng(3)
    for s in range(pc.levels):
        for v in (0, 1):
            vol = pc.volume(v, s)
            vol[...] = rng.uniform(0.0, 5.0, vol.shape)  # overwrite the oracle's volumes in place
    pc.refresh_max_cost()
    gpu_ctx.set_images(small_pair["l"], small_pair["r"])
    gpu_ctx.begin_cost(small_pair["max_dis"], 35, 3, 0.3)
    for s in range(pc.levels):
        for v in (0, 1):
            vol = pc.volume(v, s)
            for d in range(vol.shape[0]):
                gpu_ctx.upload_cost_slab(v, s, d, vol[d])
    gpu_ctx.finish_cost()
    for s in range(pc.levels):
        for v in (0, 1):
            assert gpu_ctx.max_cost(v, s) == pc.max_cost(v, s)
    xy, norm, point, param = random_planes(rng, 256, small_pair["w"], small_pair["h"], small_pair["max_dis"])
    got = gpu_ctx.plane_cost_batch(1, xy, np.concatenate([norm, param], 1))
    want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], 1, po.SUM_DEVICE) for i in range(256)])
    np.testing.assert_array_equal(got, want)


def test_foreign_volume_with_negative_cells_disables_the_early_exit(gpu_ctx, small_pair):
    """A plugin cost may be negative (NCC-style) while its max is >= 0: partial sums are then not monotone and the early
    exit would reject planes the reference accepts.  cspm_finish_cost reduces the MIN of the uploaded volumes too and
    withdraws the early-exit licence: the whole PatchMatch (early_exit=1, the default) still equals the oracle."""
    pc = po.PlaneCost(small_pair["l"], small_pair["r"], small_pair["max_dis"], 35, 3, 0.3)
    rng = np.random.default_rng(8)
    gpu_ctx.set_images(small_pair["l"], small_pair["r"])
    gpu_ctx.begin_cost(small_pair["max_dis"], 35, 3, 0.3)
    for s in range(pc.levels):
        for v in (0, 1):
            vol = pc.volume(v, s)
            vol[...] = rng.uniform(-4.0, 1.0, vol.shape)  # mostly negative cells, positive max
            for d in range(vol.shape[0]):
                gpu_ctx.upload_cost_slab(v, s, d, vol[d])
    pc.refresh_max_cost()
    gpu_ctx.finish_cost()
    assert gpu_ctx.max_cost(0, 0) == pc.max_cost(0, 0) > 0
    pm = po.PatchMatch(small_pair["l"], small_pair["r"], small_pair["max_dis"], 4)
    pm.run(2, pc, False, seed=6, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(2, seed=6, schedule=0, early_exit=1)
    for v in (0, 1):
        npar, cost = gpu_ctx.get_planes(v)
        np.testing.assert_array_equal(cost, pm.min_cost(v))
        np.testing.assert_array_equal(npar[..., :3], pm.planes(v)[..., 0:3])


def test_padded_host_rows_and_rois(gpu_ctx, small_pair):
    """cspm_set_images takes a row stride (a cv::Mat ROI): only 3*w bytes of each row are read, the last row included --
    the buffer below ENDS right after the last pixel of the last row."""
    import ctypes as C
    import crossscalepatchmatch_amd as cs
    L = cs.load_library()
    h, w = small_pair["h"], small_pair["w"]
    stride = 3 * w + 37
    bufs = []
    for img in (small_pair["l"], small_pair["r"]):
        b = np.full(stride * (h - 1) + 3 * w, 0xAB, np.uint8)
        for y in range(h):
            b[y * stride:y * stride + 3 * w] = img[y].reshape(-1)
        bufs.append(b)
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    assert L.cspm_set_images(gpu_ctx.p, u8(bufs[0]), u8(bufs[1]), w, h, stride) == 0
    gpu_ctx.w, gpu_ctx.h = w, h
    gpu_ctx.build_cost_grd(small_pair["max_dis"], 35, 2, 0.3)
    for v, img in ((0, small_pair["l"]), (1, small_pair["r"])):
        np.testing.assert_array_equal(gpu_ctx.level_image(v, 0), img)


def test_scale_weights_small_level_counts(gpu_ctx, small_pair):
    """scale_num 1..3 take cv::invert's closed-form path (det2 / det3), 4.. the LU path; device == oracle for all."""
    gpu_ctx.set_images(small_pair["l"], small_pair["r"])
    for sn in (1, 2, 3, 4, 5):
        for lam in (0.0, 0.3, 1.0):
            gpu_ctx.build_cost_grd(small_pair["max_dis"], 35, sn, lam)
            o = np.zeros(8)
            assert po.lib().csor_scale_weights(sn, lam, o.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))) == 0
            np.testing.assert_array_equal(gpu_ctx.scale_weights(), o[:sn])
            if sn >= 2:
                assert abs(gpu_ctx.scale_weights().sum() - 1.0) < 1e-12  # rows of inv(M) sum to 1: M (S >= 2) has unit row sums


def test_errors(gpu_ctx, small_pair):
    import crossscalepatchmatch_amd as cs
    gpu_ctx.set_images(small_pair["l"], small_pair["r"])
    with pytest.raises(cs.CspmError):
        gpu_ctx.build_cost_grd(0, 35, 0, 0.0)       # max_dis < 1
    with pytest.raises(cs.CspmError):
        gpu_ctx.build_cost_grd(16, 35, 99, 0.0)     # too many levels
    with pytest.raises(cs.CspmError):
        gpu_ctx.patchmatch(3)                        # no cost built (state error, not a crash)
    gpu_ctx.build_cost_grd(16, 35, 0, 0.0)
    with pytest.raises(cs.CspmError):
        gpu_ctx.plane_cost_batch(0, [[999, 0]], [[0, 0, 1, 0, 0, 5]])  # pixel outside the image


def test_golden_cost_fixture(gpu_ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cost_64x48_d16.npz"))
    for name, sn, lam in (("ss", 0, 0.0), ("cs", 5, 0.3)):
        gpu_ctx.set_images(g["l"], g["r"])
        gpu_ctx.build_cost_grd(int(g["max_dis"]), 35, sn, lam)
        np.testing.assert_array_equal(gpu_ctx.scale_weights(), g[f"{name}_wgt"])
        for v in (0, 1):
            np.testing.assert_array_equal(gpu_ctx.cost_slab(v, 0, 5), g[f"{name}_vol0_d5_dev"][v])
            for s in range(gpu_ctx.levels):
                assert gpu_ctx.max_cost(v, s) == g[f"{name}_maxc_dev"][v, s]
            got = gpu_ctx.plane_cost_batch(v, g[f"{name}_v{v}_xy"], g[f"{name}_v{v}_np"])
            np.testing.assert_array_equal(got, g[f"{name}_v{v}_device"])
            np.testing.assert_allclose(got, g[f"{name}_v{v}_serial"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("volumes", [False, True], ids=["fused", "volumes"])
@pytest.mark.parametrize("pairname", ["small_pair", "odd_pair"])
@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (5, 0.3)])
def test_census_cost(gpu_ctx, request, pairname, scale_num, lam, volumes):
    """CenCC (cc/cen_cc.cc:4-137) on the device: volumes of every level, max_cost, batched GetPlaneCost, host boundary."""
    import ctypes as C
    import crossscalepatchmatch_amd as cs
    pair = request.getfixturevalue(pairname)
    gpu_ctx.set_images(pair["l"], pair["r"])
    gpu_ctx.build_cost_cen(pair["max_dis"], 35, scale_num, lam, volumes=volumes)
    pc = po.PlaneCost(pair["l"], pair["r"], pair["max_dis"], 35, scale_num, lam, cc="CEN")
    for s in range(pc.levels):
        for v in (0, 1):
            np.testing.assert_array_equal(gpu_ctx.cost_volume(v, s), pc.volume(v, s))
            assert gpu_ctx.max_cost(v, s) == pc.max_cost(v, s)
    rng = np.random.default_rng(4)
    xy, norm, point, param = random_planes(rng, 800, pair["w"], pair["h"], pair["max_dis"])
    got = gpu_ctx.plane_cost_batch(0, xy, np.concatenate([norm, param], 1))
    want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], 0, po.SUM_DEVICE) for i in range(800)])
    np.testing.assert_array_equal(got, want)
    if scale_num == 0 and not volumes:  # CCMethod::buildCV / buildRightCV on caller-owned CV_64FC3 buffers
        L, lib = cs.load_library(), po.lib()
        h, w, D = pair["h"], pair["w"], pair["max_dis"] + 1
        l = np.ascontiguousarray(pair["l"][..., ::-1].astype(np.float64) + rng.uniform(-0.49, 0.49, (h, w, 3)))
        r = np.ascontiguousarray(pair["r"][..., ::-1].astype(np.float64) + rng.uniform(-0.49, 0.49, (h, w, 3)))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        for right in (0, 1):
            a, b = np.zeros((D, h, w)), np.zeros((D, h, w))
            assert L.cspm_cen_build_cv_host(0, dp(l), dp(r), w, h, D, right, dp(a)) == 0
            (lib.csor_cen_build_right_cv if right else lib.csor_cen_build_cv)(dp(l), dp(r), w, h, D, dp(b))
            np.testing.assert_array_equal(a, b)
