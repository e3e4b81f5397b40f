"""GPU parity for the volume-free plane costs: GrdPC (plane_cost/grd_pc.cc:11-176) and CSPC (plane_cost/cspc.cc:11-183) as
device IPlaneCost variants (cspm_build_cost_img) -- batched GetPlaneCost and the whole CSPatchMatch pipeline on top of them,
HIP path through the C ABI vs the oracle, bit-exact."""
import numpy as np
import pytest

from conftest import random_planes
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

CASES = [("grdpc", 0, 0.0), ("cspc03", 5, 0.3), ("cspc1", 5, 1.0), ("cspc3lv", 3, 0.3)]


def _build(ctx, pair, scale_num, lam, wnd=35):
    ctx.set_images(pair["l"], pair["r"])
    ctx.build_cost_img(pair["max_dis"], wnd, scale_num, lam)
    return po.PlaneCost(pair["l"], pair["r"], pair["max_dis"], wnd, scale_num, lam, cc="IMG")


@pytest.mark.parametrize("pairname", ["small_pair", "odd_pair"])
@pytest.mark.parametrize("name,scale_num,lam", CASES)
def test_plane_cost_batch(gpu_ctx, request, pairname, name, scale_num, lam):
    pair = request.getfixturevalue(pairname)
    pc = _build(gpu_ctx, pair, scale_num, lam)
    assert gpu_ctx.levels == pc.levels
    np.testing.assert_array_equal(gpu_ctx.scale_weights(), pc.scale_wgt())
    for s in range(pc.levels):
        assert gpu_ctx.level_dims(s) == pc.dims(s)
        for v in (0, 1):
            np.testing.assert_array_equal(gpu_ctx.level_image(v, s), pc.image(v, s))
            assert gpu_ctx.max_cost(v, s) == pc.max_cost(v, s) == 0.1 * 10.0 + (1 - 0.1) * 2.0
    rng = np.random.default_rng(199)
    n = 2600
    for view in (0, 1):
        xy, norm, point, param = random_planes(rng, n, pair["w"], pair["h"], pair["max_dis"])
        got = gpu_ctx.plane_cost_batch(view, xy, np.concatenate([norm, param], 1))
        want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], view, po.SUM_DEVICE) for i in range(n)])
        np.testing.assert_array_equal(got, want)
        idx = rng.choice(n, 300, replace=False)
        ser = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], view, po.SUM_SERIAL) for i in idx])
        np.testing.assert_allclose(got[idx], ser, rtol=1e-12, atol=0)  # the reference's order: rounding only


def test_no_volumes(gpu_ctx, small_pair):
    _build(gpu_ctx, small_pair, 3, 0.3)
    import crossscalepatchmatch_amd as cs
    with pytest.raises(cs.CspmError, match="no cost volumes"):
        gpu_ctx.cost_volume(0, 0)


@pytest.mark.parametrize("wnd", [1, 5, 9, 13, 45])
@pytest.mark.parametrize("name,scale_num,lam", [("grdpc", 0, 0.0), ("cspc", 5, 0.3)])
def test_window_sizes_through_both_engines(gpu_ctx, odd_pair, name, scale_num, lam, wnd):
    pc = _build(gpu_ctx, odd_pair, scale_num, lam, wnd)
    pm = po.PatchMatch(odd_pair["l"], odd_pair["r"], odd_pair["max_dis"], 4)
    pm.run(2, pc, False, seed=wnd, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(2, seed=wnd, schedule=po.SCHED_RASTER)
    _assert_state_equal(gpu_ctx, pm, f"wnd {wnd}")


def _assert_state_equal(ctx, pm, what):
    for v in (0, 1):
        npar, cost = ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3], err_msg=f"{what}: norm, view {v}")
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9], err_msg=f"{what}: param, view {v}")
        np.testing.assert_array_equal(cost, pm.min_cost(v), err_msg=f"{what}: min_cost, view {v}")


@pytest.mark.parametrize("pairname", ["mid_pair", "odd_pair"])
@pytest.mark.parametrize("name,scale_num,lam", [("grdpc", 0, 0.0), ("cspc", 5, 0.3)])
@pytest.mark.parametrize("sched", [po.SCHED_REDBLACK, po.SCHED_RASTER])
def test_whole_pipeline_bit_exact(gpu_ctx, request, pairname, name, scale_num, lam, sched):
    """CSPatchMatch::PatchMatch (cs_patchmatch.cc:51-109) + PostProcessing with `new GrdPC` / `new CSPC` as the plane cost."""
    pair = request.getfixturevalue(pairname)
    pc = _build(gpu_ctx, pair, scale_num, lam)
    pm = po.PatchMatch(pair["l"], pair["r"], pair["max_dis"], 4)
    pm.run(3, pc, False, seed=21, schedule=sched, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(3, seed=21, schedule=sched)
    _assert_state_equal(gpu_ctx, pm, "final")
    pm.postprocess()
    l, r = gpu_ctx.postprocess(4)
    np.testing.assert_array_equal(l, pm.dis(0))
    np.testing.assert_array_equal(r, pm.dis(1))


def test_early_exit_is_result_preserving(gpu_ctx, small_pair):
    pc = _build(gpu_ctx, small_pair, 5, 0.3)
    gpu_ctx.patchmatch(2, seed=5, schedule=po.SCHED_RASTER, early_exit=True)
    a = [gpu_ctx.get_planes(v) for v in (0, 1)]
    gpu_ctx.patchmatch(2, seed=5, schedule=po.SCHED_RASTER, early_exit=False)
    for v in (0, 1):
        np.testing.assert_array_equal(a[v][0], gpu_ctx.get_planes(v)[0])
        np.testing.assert_array_equal(a[v][1], gpu_ctx.get_planes(v)[1])


def test_disparity_range_wider_than_the_lds_strip(gpu_ctx):
    """max_dis = 400 on a 420-pixel-wide pair: level 0 of the row engine cannot stage the other view's row window (64 centres +
    window + disparity range > 384 slots) and reads both views from global memory; the wrapped pad columns are exercised by
    every pixel near the left (left view) / right (right view) border."""
    from crossscalepatchmatch_amd import synth
    l, r, _, _ = synth.make_pair(420, 10, 60, regions=3, seed=78)
    D = 400
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_img(D, 9, 3, 0.3)
    pc = po.PlaneCost(l, r, D, 9, 3, 0.3, cc="IMG")
    pm = po.PatchMatch(l, r, D, 1)
    pm.run(1, pc, False, seed=8, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(1, seed=8, schedule=po.SCHED_RASTER)
    _assert_state_equal(gpu_ctx, pm, "wide disparity range")


def test_good_disparities_on_a_synthetic_pair(gpu_ctx, mid_pair):
    """sanity beyond parity: the GrdPC / CSPC costs find the ground truth of the synthetic pair"""
    for scale_num, lam in ((0, 0.0), (5, 0.3)):
        _build(gpu_ctx, mid_pair, scale_num, lam)
        gpu_ctx.patchmatch(3, seed=2, schedule=po.SCHED_RASTER)
        d = gpu_ctx.disparity_f64(0)
        gt = mid_pair["gl"]
        m = np.s_[4:-4, mid_pair["max_dis"] + 4:-4]
        assert np.mean(np.abs(d[m] - gt[m]) > 1.0) < 0.25  # raw left disparities, occlusions included, no post-processing


def test_golden_fixture(gpu_ctx):
    """the committed answer (tests/golden/imgcost_64x48_d16.npz, made by tests/golden/make_golden.py from the oracle)"""
    import hashlib
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imgcost_64x48_d16.npz"))
    for name, sn, lam in (("grdpc", 0, 0.0), ("cspc", 5, 0.3)):
        gpu_ctx.set_images(g["l"], g["r"])
        gpu_ctx.build_cost_img(int(g["max_dis"]), 35, sn, lam)
        for v in (0, 1):
            got = gpu_ctx.plane_cost_batch(v, g[f"{name}_v{v}_xy"], g[f"{name}_v{v}_np"])
            np.testing.assert_array_equal(got, g[f"{name}_v{v}_device"])
            np.testing.assert_allclose(got, g[f"{name}_v{v}_serial"], rtol=1e-12, atol=0)
        gpu_ctx.patchmatch(2, seed=int(g["seed"]), schedule=po.SCHED_RASTER)
        for v in (0, 1):
            np.testing.assert_array_equal(gpu_ctx.disparity_u8(v, int(g["dis_scale"])), g[f"{name}_dis"][v])
            npar, _ = gpu_ctx.get_planes(v)
            assert hashlib.sha256(np.ascontiguousarray(npar).tobytes()).hexdigest() == str(g[f"{name}_plane_sha"][v])
        l, r = gpu_ctx.postprocess(int(g["dis_scale"]))
        np.testing.assert_array_equal(l, g[f"{name}_pp"][0])
        np.testing.assert_array_equal(r, g[f"{name}_pp"][1])
