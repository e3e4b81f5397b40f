"""GPU parity on inputs and flags the noise-texture suite cannot reach (round-4 review, "Next" item 1).

(a) Adversarial pairs (crossscalepatchmatch_amd/synth.py make_adversarial): constant-colour blocks, 0/255 saturation, duplicated
    rows, a periodic texture with period < max_dis, all-black / all-white, L == R, saturated stripes, half flat.  On those inputs
    candidate planes TIE exactly, min_cost reaches 0 and whole volumes saturate -- the reference's strict `<` accept rules
    (cs_patchmatch.cc:182,192,201,209,270,335), the `min_cost == 0` early-exit thresholds and `max_cost` when no cell saturates
    (pre_cs_pc.cc:75-82) decide the outcome.  Whole pipeline + post-processing against the oracle in the device order:
    array_equal on planes, stored costs, raw and post-processed maps.
(b) The reference CLI's own default `--use_cs=true --reg_lambda=0` (main.cc:34: scale weights [1,0,0,0,0], pre_cs_pc.cc:86-109) and
    lambda = 1 through the whole pipeline, early exit on and off.
"""
import numpy as np
import pytest

from crossscalepatchmatch_amd import synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

W, H, D, SCALE = 144, 80, 24, 4


def _build(ctx, l, r, cc, scale_num, lam, max_dis=D, **kw):
    ctx.set_images(l, r)
    if cc == "GRD":
        ctx.build_cost_grd(max_dis, 35, scale_num, lam, **kw)
    elif cc == "CEN":
        ctx.build_cost_cen(max_dis, 35, scale_num, lam)
    else:
        ctx.build_cost_img(max_dis, 35, scale_num, lam)
    return po.PlaneCost(l, r, max_dis, 35, scale_num, lam, cc=cc), po.PatchMatch(l, r, max_dis, SCALE)


def _assert_equal(ctx, pm, what):
    for v in (0, 1):
        npar, cost = ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3], err_msg=f"{what}: norm, view {v}")
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9], err_msg=f"{what}: param, view {v}")
        np.testing.assert_array_equal(cost, pm.min_cost(v), err_msg=f"{what}: min_cost, view {v}")
        np.testing.assert_array_equal(ctx.disparity_u8(v, SCALE), pm.dis(v), err_msg=f"{what}: 8-bit map, view {v}")
        np.testing.assert_array_equal(ctx.disparity_f64(v), pm.disp_f64(v), err_msg=f"{what}: disparity, view {v}")


def _run_and_compare(ctx, l, r, cc, scale_num, lam, sched, iters=3, seed=1, early_exit=1, what="", **kw):
    pc, pm = _build(ctx, l, r, cc, scale_num, lam, **kw)
    pm.run(iters, pc, False, seed=seed, schedule=sched, sum_order=po.SUM_DEVICE, rb_rounds=1, rb_neighbours=4, wavefront=True)
    ctx.patchmatch(iters, seed=seed, schedule=sched, rb_rounds=1, rb_neighbours=4, early_exit=early_exit)
    _assert_equal(ctx, pm, what)
    pm.postprocess()
    lo, ro = ctx.postprocess(SCALE)
    np.testing.assert_array_equal(lo, pm.dis(0), err_msg=f"{what}: post-processed left map")
    np.testing.assert_array_equal(ro, pm.dis(1), err_msg=f"{what}: post-processed right map")
    return pm


@pytest.mark.parametrize("name,cc,scale_num,lam", [("grd_ss", "GRD", 0, 0.0), ("grd_cs", "GRD", 5, 0.3)])
@pytest.mark.parametrize("kind", synth.ADVERSARIAL_KINDS)
def test_adversarial_pairs_grd_raster(gpu_ctx, kind, name, cc, scale_num, lam):
    """PreSSPC / PreCSPC + GRD, reference raster order, every adversarial kind."""
    l, r = synth.make_adversarial(kind, W, H, D, seed=5)
    pm = _run_and_compare(gpu_ctx, l, r, cc, scale_num, lam, po.SCHED_RASTER, what=f"{kind}/{name}")
    if kind in ("black", "white") and scale_num == 0:
        assert np.mean(pm.min_cost(0) == 0.0) > 0.5  # not vacuous: most pixels sit at cost 0, nothing is ever `<`


@pytest.mark.parametrize("cc", ["CEN", "IMG"])
@pytest.mark.parametrize("kind", ["blocks", "black", "dup_rows", "periodic"])
def test_adversarial_pairs_other_costs(gpu_ctx, kind, cc):
    """census (integer cells: ties are the norm) and the volume-free GrdPC / CSPC costs, cross-scale, raster order."""
    l, r = synth.make_adversarial(kind, W, H, D, seed=6)
    _run_and_compare(gpu_ctx, l, r, cc, 5, 0.3, po.SCHED_RASTER, iters=2, what=f"{kind}/{cc}")


@pytest.mark.parametrize("kind", ["blocks", "saturated", "dup_rows", "black"])
def test_adversarial_pairs_redblack(gpu_ctx, kind):
    """the red-black option: four neighbours compete for every pixel, ties resolve by the fixed neighbour order"""
    l, r = synth.make_adversarial(kind, W, H, D, seed=7)
    _run_and_compare(gpu_ctx, l, r, "GRD", 5, 0.3, po.SCHED_REDBLACK, iters=2, what=f"{kind}/redblack")


@pytest.mark.parametrize("variant", ["computed_tables", "volumes", "raster_launches"])
@pytest.mark.parametrize("kind", ["blocks", "black"])
def test_adversarial_pairs_other_cell_sources(gpu_ctx, kind, variant):
    """the same ties through the other ways a cell reaches a tap (computed tables, materialised volumes) and the per-diagonal sweep"""
    from crossscalepatchmatch_amd import capi
    l, r = synth.make_adversarial(kind, W, H, D, seed=8)
    kw = {}
    if variant == "computed_tables":
        kw = dict(table_volumes=False)
    if variant == "volumes":
        kw = dict(volumes=True)
    try:
        if variant == "raster_launches":
            gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 1)
        _run_and_compare(gpu_ctx, l, r, "GRD", 5, 0.3, po.SCHED_RASTER, iters=2, what=f"{kind}/{variant}", **kw)
    finally:
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)


@pytest.mark.parametrize("early_exit", [0, 1])
@pytest.mark.parametrize("lam", [0.0, 1.0])
@pytest.mark.parametrize("src", ["noise", "blocks", "black"])
def test_cross_scale_lambda_0_and_1_whole_pipeline(gpu_ctx, mid_pair, src, lam, early_exit):
    """`--use_cs=true` with main.cc:34's default `--reg_lambda=0.0` (scale weights exactly [1,0,0,0,0]: four levels are evaluated
    and multiplied by zero, pre_cs_pc.cc:180) and with lambda = 1, early exit on and off: the per-row exit's `need` is +inf / NaN
    for a zero level weight by design -- the whole pipeline must still equal the oracle."""
    if src == "noise":
        l, r, max_dis = mid_pair["l"], mid_pair["r"], mid_pair["max_dis"]
    else:
        l, r = synth.make_adversarial(src, 112, 64, 16, seed=9)
        max_dis = 16
    pc, pm = _build(gpu_ctx, l, r, "GRD", 5, lam, max_dis=max_dis)
    np.testing.assert_array_equal(gpu_ctx.scale_weights(), pc.scale_wgt())
    if lam == 0.0:
        assert list(pc.scale_wgt()) == [1.0, 0.0, 0.0, 0.0, 0.0]
    pm.run(3, pc, False, seed=2, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
    gpu_ctx.patchmatch(3, seed=2, schedule=po.SCHED_RASTER, early_exit=early_exit)
    _assert_equal(gpu_ctx, pm, f"{src} lambda {lam} early_exit {early_exit}")
    pm.postprocess()
    lo, ro = gpu_ctx.postprocess(SCALE)
    np.testing.assert_array_equal(lo, pm.dis(0))
    np.testing.assert_array_equal(ro, pm.dis(1))


@pytest.mark.parametrize("lam", [0.0, 1.0])
def test_lambda_0_and_1_census_and_img(gpu_ctx, small_pair, lam):
    l, r, max_dis = small_pair["l"], small_pair["r"], small_pair["max_dis"]
    for cc in ("CEN", "IMG"):
        pc, pm = _build(gpu_ctx, l, r, cc, 5, lam, max_dis=max_dis)
        pm.run(2, pc, False, seed=3, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
        gpu_ctx.patchmatch(2, seed=3, schedule=po.SCHED_RASTER)
        _assert_equal(gpu_ctx, pm, f"{cc} lambda {lam}")


def test_adversarial_plane_cost_batch(gpu_ctx):
    """GetPlaneCost itself (chain engine) on the saturated / flat pairs: fronto-parallel planes at integer disparities
    (floor_wgt == 1 exactly), at the range ends and outside."""
    rng = np.random.default_rng(4)
    for kind in ("saturated", "black", "periodic"):
        l, r = synth.make_adversarial(kind, W, H, D, seed=10)
        for sn, lam in ((0, 0.0), (5, 0.0), (5, 0.3)):
            pc, _ = _build(gpu_ctx, l, r, "GRD", sn, lam)
            n = 96
            xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
            d = rng.choice([0.0, 0.5, 1.0, 5.0, 8.0, D - 1.0, D - 0.5, float(D), D + 3.0, -2.0], n)
            norm = np.tile([0.0, 0.0, 1.0], (n, 1))
            slant = rng.random(n) < 0.4
            norm[slant] = rng.normal(size=(int(slant.sum()), 3))
            norm /= np.linalg.norm(norm, axis=1, keepdims=True)
            param = np.stack([po.plane_param(norm[i], [xy[i, 0], xy[i, 1], d[i]]) for i in range(n)])
            for v in (0, 1):
                got = gpu_ctx.plane_cost_batch(v, xy, np.concatenate([norm, param], 1))
                want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], v, po.SUM_DEVICE) for i in range(n)])
                np.testing.assert_array_equal(got, want, err_msg=f"{kind} scale_num {sn} lambda {lam} view {v}")


@pytest.mark.parametrize("kind", ["saturated", "stripes", "identical", "white", "noise"])
def test_sweep_packed_pixels_equal_the_12_byte_pixels(gpu_ctx, kind):
    """With CSPM_OPT_SWEEP_PACKED the raster sweep reads packed 8-byte pixels {36-bit fixed-point gradient, colour} (an option: measured
    slower, off by default); every other kernel and cspm_plane_cost_batch read the 12-byte pixels {f64 gradient, colour}.  Lossless by construction: no pixel may
    be reported unrepresentable, and the sweep must give the same planes and costs either way -- on the inputs with the largest
    gradients (0/255 stripes and blocks), on noise, cross-scale and single scale, persistent and per-diagonal sweeps."""
    from crossscalepatchmatch_amd import capi
    if kind == "noise":
        l, r = synth.make_pair(W, H, D, regions=3, seed=12)[:2]
    else:
        l, r = synth.make_adversarial(kind, W, H, D, seed=11)
    for sn, lam in ((5, 0.3), (0, 0.0)):
        out = []
        for packed, launches in ((1, 0), (0, 0), (1, 1)):
            try:
                gpu_ctx.set_option(capi.OPT_SWEEP_PACKED, packed)
                gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, launches)
                gpu_ctx.set_images(l, r)
                gpu_ctx.build_cost_grd(D, 35, sn, lam)
                assert gpu_ctx.get_option(capi.OPT_SWEEP_PACKED_ACTIVE) == packed
                assert gpu_ctx.get_option(capi.OPT_SWEEP_PACKED_BAD) == 0
                gpu_ctx.patchmatch(2, seed=6, schedule=po.SCHED_RASTER)
                out.append([gpu_ctx.get_planes(v) for v in (0, 1)])
            finally:
                gpu_ctx.set_option(capi.OPT_SWEEP_PACKED, 0)
                gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)
        for k in (1, 2):
            for v in (0, 1):
                np.testing.assert_array_equal(out[0][v][0], out[k][v][0], err_msg=f"{kind} scale_num {sn} variant {k} view {v}")
                np.testing.assert_array_equal(out[0][v][1], out[k][v][1], err_msg=f"{kind} scale_num {sn} variant {k} view {v}")
        # the stored costs (written by the sweep from packed pixels, by the row kernels from strips) are what cspm_plane_cost_batch
        # computes from the 12-byte pixels
        rng = np.random.default_rng(2)
        ys, xs = rng.integers(0, H, 200), rng.integers(0, W, 200)
        for v in (0, 1):
            npar, cost = out[0][v]
            np.testing.assert_array_equal(gpu_ctx.plane_cost_batch(v, np.stack([xs, ys], 1), npar[ys, xs]), cost[ys, xs])


@pytest.mark.parametrize("kind", ["blocks", "black", "noise", "tall"])
def test_sweep_dataflow_scheduling_equals_ordered_claims(gpu_ctx, kind):
    """The persistent raster sweep hands out its pixels by ordered claims (the default) or by dataflow (CSPM_OPT_SWEEP_FLOW = 1: the
    workgroup that makes a pixel ready continues with it; an option, measured slower); the per-diagonal launches are the third schedule.  The dependencies are
    the reference's raster order in all three: identical planes and costs -- on tie-heavy pairs (where the order of evaluation would
    show if it leaked into a decision), on noise, on a tall narrow pair (long columns: continuation mostly downwards), single- and
    cross-scale, with 1 and 3 workgroups per CU."""
    import os
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import capi
    if kind == "noise":
        l, r = synth.make_pair(W, H, D, regions=3, seed=12)[:2]
    elif kind == "tall":
        l, r = synth.make_pair(40, 200, 12, regions=3, seed=13)[:2]
    else:
        l, r = synth.make_adversarial(kind, W, H, D, seed=14)
    max_dis = 12 if kind == "tall" else D
    for sn, lam in ((5, 0.3), (0, 0.0)):
        out = []
        for flow, launches in ((1, 0), (0, 0), (0, 1)):
            try:
                gpu_ctx.set_option(capi.OPT_SWEEP_FLOW, flow)
                gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, launches)
                gpu_ctx.set_images(l, r)
                gpu_ctx.build_cost_grd(max_dis, 35, sn, lam)
                gpu_ctx.patchmatch(3, seed=6, schedule=po.SCHED_RASTER)
                out.append([gpu_ctx.get_planes(v) for v in (0, 1)])
                assert gpu_ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 0
            finally:
                gpu_ctx.set_option(capi.OPT_SWEEP_FLOW, 0)
                gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)
        for wg in (1, 3):
            os.environ["CSPM_SWEEP_WG"] = str(wg)
            os.environ["CSPM_SWEEP_FLOW"] = "1"
            try:
                ctx = cs.StereoContext(0)
            finally:
                del os.environ["CSPM_SWEEP_WG"]
                del os.environ["CSPM_SWEEP_FLOW"]
            try:
                ctx.set_images(l, r)
                ctx.build_cost_grd(max_dis, 35, sn, lam)
                ctx.patchmatch(3, seed=6, schedule=po.SCHED_RASTER)
                out.append([ctx.get_planes(v) for v in (0, 1)])
                assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 0
            finally:
                ctx.close()
        for k in range(1, len(out)):
            for v in (0, 1):
                np.testing.assert_array_equal(out[0][v][0], out[k][v][0], err_msg=f"{kind} scale_num {sn} variant {k} view {v}")
                np.testing.assert_array_equal(out[0][v][1], out[k][v][1], err_msg=f"{kind} scale_num {sn} variant {k} view {v}")
