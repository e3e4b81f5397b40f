"""GPU parity, PatchMatch side (cs_patchmatch.cc:51-345, 590-601): every phase and the whole pipeline
through the C ABI vs the oracle run with the same schedule, RNG and summation order -- bit-exact
planes, costs and 8-bit disparity maps.  Plus the north-star bar against the reference-order oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

MODES = [("ss", 0, 0.0), ("cs", 5, 0.3)]


def _setup(ctx, pair, scale_num, lam, volumes=False, sweep_pairs=False, table_volumes=True):
    """volumes: False = fused cells + the device-cell volumes the row kernels' tables are DMA-filled from (the default build),
    "computed" = fused cells, tables computed (CSPM_OPT_TABLE_VOLUMES = 0: what a pair too large for the volumes gets), "pairs" =
    the default + paired-cell volumes for the raster sweep (CSPM_OPT_SWEEP_PAIRS), True = materialised f64 volumes read directly"""
    if volumes == "pairs":
        volumes, sweep_pairs = False, True
    if volumes == "computed":
        volumes, table_volumes = False, False
    ctx.set_images(pair["l"], pair["r"])
    ctx.build_cost_grd(pair["max_dis"], 35, scale_num, lam, volumes=volumes, sweep_pairs=sweep_pairs, table_volumes=table_volumes)
    from crossscalepatchmatch_amd import capi
    assert ctx.get_option(capi.OPT_SWEEP_PAIRS_ACTIVE) == int(bool(sweep_pairs) and not volumes)
    assert ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE) == int(bool(table_volumes) and not volumes)
    pc = po.PlaneCost(pair["l"], pair["r"], pair["max_dis"], 35, scale_num, lam)
    pm = po.PatchMatch(pair["l"], pair["r"], pair["max_dis"], 4)
    return pc, pm


def _assert_state_equal(ctx, pm, what):
    for v in (0, 1):
        npar, cost = ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3], err_msg=f"{what}: norm, view {v}")
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9], err_msg=f"{what}: param, view {v}")
        np.testing.assert_array_equal(cost, pm.min_cost(v), err_msg=f"{what}: min_cost, view {v}")


@pytest.mark.parametrize("volumes", [False, "computed", "pairs", True], ids=["fused", "computed_tables", "sweep_pairs", "volumes"])
@pytest.mark.parametrize("name,scale_num,lam", MODES)
@pytest.mark.parametrize("sched", [po.SCHED_REDBLACK, po.SCHED_RASTER])
def test_phase_by_phase(gpu_ctx, small_pair, name, scale_num, lam, sched, volumes):
    pc, pm = _setup(gpu_ctx, small_pair, scale_num, lam, volumes)
    kw_o = dict(seed=777, schedule=sched, sum_order=po.SUM_DEVICE, rb_rounds=2, rb_neighbours=4)
    kw_g = dict(seed=777, schedule=sched, rb_rounds=2, rb_neighbours=4, early_exit=1)
    pm.init(pc, **kw_o); gpu_ctx.pm_init(**kw_g)
    _assert_state_equal(gpu_ctx, pm, "init")
    for it in (0, 1):
        pm.spatial(it, pc, **kw_o); gpu_ctx.pm_spatial(it, **kw_g)
        _assert_state_equal(gpu_ctx, pm, f"spatial {it}")
        pm.view(it, pc, **kw_o); gpu_ctx.pm_view(it, **kw_g)
        _assert_state_equal(gpu_ctx, pm, f"view {it}")
        pm.refine(it, pc, **kw_o); gpu_ctx.pm_refine(it, **kw_g)
        _assert_state_equal(gpu_ctx, pm, f"refine {it}")


@pytest.mark.parametrize("pairname", ["mid_pair", "odd_pair"])
@pytest.mark.parametrize("name,scale_num,lam", MODES)
@pytest.mark.parametrize("sched", [po.SCHED_REDBLACK, po.SCHED_RASTER, "raster_pairs", "raster_computed_tables"])
def test_whole_pipeline_bit_exact(gpu_ctx, request, pairname, name, scale_num, lam, sched):
    """T3/T4: PatchMatch(3, plane_cost, false) + PlaneToDisp."""
    pair = request.getfixturevalue(pairname)
    src = False
    if sched == "raster_pairs":
        sched, src = po.SCHED_RASTER, "pairs"
    if sched == "raster_computed_tables":
        sched, src = po.SCHED_RASTER, "computed"
    pc, pm = _setup(gpu_ctx, pair, scale_num, lam, src)
    pm.run(3, pc, False, seed=4242, schedule=sched, sum_order=po.SUM_DEVICE, rb_rounds=1, rb_neighbours=4)
    gpu_ctx.patchmatch(3, seed=4242, schedule=sched, rb_rounds=1, rb_neighbours=4, early_exit=1)
    _assert_state_equal(gpu_ctx, pm, "final")
    for v in (0, 1):
        np.testing.assert_array_equal(gpu_ctx.disparity_u8(v, 4), pm.dis(v))
        np.testing.assert_array_equal(gpu_ctx.disparity_f64(v), pm.disp_f64(v))


def test_north_star_bar_vs_reference_order(gpu_ctx, mid_pair):
    """>= 99.5 % of pixels within 0.5 px of the reference-order CPU path (raster sweep, serial
    summation) on identical inputs and identical random numbers."""
    for name, scale_num, lam in MODES:
        pc, pm = _setup(gpu_ctx, mid_pair, scale_num, lam)
        pm.run(3, pc, False, seed=31337, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL)
        gpu_ctx.patchmatch(3, seed=31337, schedule=po.SCHED_RASTER, early_exit=1)
        for v in (0, 1):
            d = np.abs(gpu_ctx.disparity_f64(v) - pm.disp_f64(v))
            assert np.mean(d <= 0.5) >= 0.995, (name, v, float(np.mean(d <= 0.5)))


def test_early_exit_is_result_preserving(gpu_ctx, small_pair):
    _setup(gpu_ctx, small_pair, 5, 0.3)
    out = []
    for ee in (0, 1):
        gpu_ctx.patchmatch(2, seed=5, schedule=po.SCHED_REDBLACK, early_exit=ee)
        out.append([gpu_ctx.get_planes(v) for v in (0, 1)])
    for v in (0, 1):
        np.testing.assert_array_equal(out[0][v][0], out[1][v][0])
        np.testing.assert_array_equal(out[0][v][1], out[1][v][1])


def test_row_shared_rng_quirk(gpu_ctx, small_pair):
    """USE_OMP quirk (cs_patchmatch.cc:129-131): every row draws the same stream."""
    pc, pm = _setup(gpu_ctx, small_pair, 0, 0.0)
    pm.init(pc, seed=9, rng_mode=po.RNG_ROW_SHARED, sum_order=po.SUM_DEVICE)
    gpu_ctx.pm_init(seed=9, rng_mode=1)
    _assert_state_equal(gpu_ctx, pm, "init row-shared")
    npar, _ = gpu_ctx.get_planes(0)
    assert np.all(npar[:, :, :3] == npar[0:1, :, :3])  # identical normals down every column


@pytest.mark.parametrize("view_sort", [1, 0])
def test_set_planes_roundtrip_and_view_ties(gpu_ctx, small_pair, view_sort):
    """Collisions in view propagation: all source pixels of a row carry the same fronto-parallel plane,
    so several of them hit one target with EQUAL cost; the first in traversal order must win -- whether the proposals are evaluated
    in target-column order (CSPM_OPT_VIEW_SORT, the default: equal targets then sit in neighbouring lanes in no particular order)
    or in source order."""
    from crossscalepatchmatch_amd import capi
    gpu_ctx.set_option(capi.OPT_VIEW_SORT, view_sort)
    pc, pm = _setup(gpu_ctx, small_pair, 0, 0.0)
    h, w, D = small_pair["h"], small_pair["w"], small_pair["max_dis"]
    rng = np.random.default_rng(1)
    for it in (0, 1):
        for v in (0, 1):
            P = pm.planes(v)
            d = np.repeat(rng.integers(1, D - 1, (h, 1)).astype(np.float64), w, 1) + rng.choice([0.0, 0.3], (h, w))
            P[..., 0:2] = 0.0; P[..., 2] = 1.0
            P[..., 3] = np.arange(w)[None, :]; P[..., 4] = np.arange(h)[:, None]; P[..., 5] = d
            P[..., 6:8] = 0.0; P[..., 8] = d
            pm.min_cost(v)[...] = 1e9
            gpu_ctx.set_planes(v, np.concatenate([P[..., 0:3], P[..., 6:9]], -1), pm.min_cost(v))
        pm.view(it, pc, sum_order=po.SUM_DEVICE); gpu_ctx.pm_view(it)
        _assert_state_equal(gpu_ctx, pm, f"view ties iter {it}")


@pytest.mark.parametrize("name,scale_num,lam", MODES)
def test_postprocessing_bit_exact(gpu_ctx, mid_pair, name, scale_num, lam):
    """PostProcessing (cs_patchmatch.cc:508-588): LR check, fill, 35x35 weighted median -- 8-bit maps identical."""
    pc, pm = _setup(gpu_ctx, mid_pair, scale_num, lam)
    pm.run(3, pc, True, seed=99, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(3, seed=99, schedule=po.SCHED_RASTER)
    raw = [gpu_ctx.disparity_u8(v, 4) for v in (0, 1)]
    l, r = gpu_ctx.postprocess(4)
    np.testing.assert_array_equal(l, pm.dis(0))
    np.testing.assert_array_equal(r, pm.dis(1))
    assert np.mean(l != raw[0]) > 0.005  # the test is not vacuous: post-processing changed pixels


def test_postprocessing_hand_made_fields(gpu_ctx):
    """PostProcessing on plane fields chosen for its corner cases, 300 columns (more than the 256 lanes of the row scan):
    a row without one consistent pixel (nothing to fill from), rows whose only consistent pixels sit at one end,
    a fully consistent row, zero disparities (never consistent, cs_patchmatch.cc:362), and noise everywhere else."""
    from crossscalepatchmatch_amd import synth
    w, h, D = 300, 24, 16
    l, r, _, _ = synth.make_pair(w, h, D, regions=3, seed=31)
    pair = dict(l=l, r=r, max_dis=D)
    pc, pm = _setup(gpu_ctx, pair, 0, 0.0)
    pm.init(pc, seed=5, sum_order=po.SUM_DEVICE)
    gpu_ctx.pm_init(seed=5)
    rng = np.random.default_rng(7)
    d = [rng.integers(0, D - 1, (h, w)).astype(np.float64) + rng.choice([0.0, 0.25, 0.5], (h, w)) for _ in (0, 1)]
    for v in (0, 1):
        d[v][0, :] = 5.0                      # row 0: consistent except where x -/+ 5 leaves the image
        d[v][2, :] = 3.0; d[v][3, :] = 3.0
    d[0][1, :] = 0.0; d[1][1, :] = 9.0        # row 1: left disparity 0 -> never consistent; right finds 0 vs 9 -> never
    d[1][2, : w - 8] = 11.0                   # row 2: only the right end agrees
    d[1][3, 8:] = 11.0                        # row 3: only the left end agrees
    for v in (0, 1):
        P = pm.planes(v)
        P[..., 0:2] = 0.0; P[..., 2] = 1.0
        P[..., 3] = np.arange(w)[None, :]; P[..., 4] = np.arange(h)[:, None]; P[..., 5] = d[v]
        P[..., 6:8] = 0.0; P[..., 8] = d[v]
        # slanted planes on a few rows so that FillInvalid's "smaller of the two planes at x" differs from both neighbours
        P[5:9, :, 6] = rng.uniform(-0.2, 0.2, (4, w)); P[5:9, :, 8] = d[v][5:9] - P[5:9, :, 6] * np.arange(w)[None, :]
        gpu_ctx.set_planes(v, np.concatenate([P[..., 0:3], P[..., 6:9]], -1), pm.min_cost(v))
    pm.plane_to_disp()
    raw = [gpu_ctx.disparity_u8(v, 4) for v in (0, 1)]
    for v in (0, 1):
        np.testing.assert_array_equal(raw[v], pm.dis(v))
    pm.postprocess()
    lo, ro = gpu_ctx.postprocess(4)
    np.testing.assert_array_equal(lo, pm.dis(0))
    np.testing.assert_array_equal(ro, pm.dis(1))
    assert np.mean(lo != raw[0]) > 0.3                 # most of the noise was replaced


def test_golden_pipeline_fixture(gpu_ctx):
    """The committed answer of tests/golden/make_golden.py (oracle-generated; the reference is unbuildable here)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_96x64_d16.npz"))
    for name, sn, lam in MODES:
        gpu_ctx.set_images(g["l"], g["r"])
        gpu_ctx.build_cost_grd(int(g["max_dis"]), 35, sn, lam)
        for sname, sched in (("raster_device", po.SCHED_RASTER), ("redblack_device", po.SCHED_REDBLACK)):
            gpu_ctx.patchmatch(3, seed=int(g["seed"]), schedule=sched, rb_rounds=1, rb_neighbours=4)
            k = f"{name}_{sname}"
            for v in (0, 1):
                np.testing.assert_array_equal(gpu_ctx.disparity_u8(v, int(g["dis_scale"])), g[k + "_dis"][v])
                _, cost = gpu_ctx.get_planes(v)
                assert cost.sum() == g[k + "_cost_sum"][v]
            l, r = gpu_ctx.postprocess(int(g["dis_scale"]))
            np.testing.assert_array_equal(np.stack([l, r]), g[k + "_pp"])


@pytest.mark.parametrize("pairname", ["odd_pair", "mid_pair"])
def test_raster_sweep_persistent_equals_per_diagonal_launches(gpu_ctx, request, pairname):
    """The persistent dataflow sweep (per-pixel done flags across workgroups / XCDs) and the one-launch-per-
    anti-diagonal sweep are the same computation; both equal the oracle's serial raster loop."""
    from crossscalepatchmatch_amd import capi
    pair = request.getfixturevalue(pairname)
    pc, pm = _setup(gpu_ctx, pair, 5, 0.3)
    pm.run(3, pc, False, seed=606, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    try:
        for launches in (1, 0, 0):  # the persistent variant twice: epochs advance, flags are reused
            gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, launches)
            gpu_ctx.patchmatch(3, seed=606, schedule=po.SCHED_RASTER)
            _assert_state_equal(gpu_ctx, pm, f"raster launches={launches}")
    finally:
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)


@pytest.mark.parametrize("volumes", [False, True], ids=["fused", "volumes"])
@pytest.mark.parametrize("scale_num,lam", [(0, 0.0), (5, 0.3)])
def test_census_pipeline_bit_exact(gpu_ctx, mid_pair, scale_num, lam, volumes):
    """--cc_name=CEN: the whole PatchMatch + post-processing with the census cost, reference raster order."""
    gpu_ctx.set_images(mid_pair["l"], mid_pair["r"])
    gpu_ctx.build_cost_cen(mid_pair["max_dis"], 35, scale_num, lam, volumes=volumes)
    pc = po.PlaneCost(mid_pair["l"], mid_pair["r"], mid_pair["max_dis"], 35, scale_num, lam, cc="CEN")
    pm = po.PatchMatch(mid_pair["l"], mid_pair["r"], mid_pair["max_dis"], 4)
    pm.run(3, pc, False, seed=11, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(3, seed=11, schedule=po.SCHED_RASTER)
    _assert_state_equal(gpu_ctx, pm, "census final")
    pm.postprocess()
    l, r = gpu_ctx.postprocess(4)
    np.testing.assert_array_equal(l, pm.dis(0))
    np.testing.assert_array_equal(r, pm.dis(1))


@pytest.mark.parametrize("wnd", [1, 3, 9, 13, 41, 45])
@pytest.mark.parametrize("name,scale_num,lam", MODES)
def test_window_sizes_through_both_engines(gpu_ctx, odd_pair, name, scale_num, lam, wnd):
    """wnd_size is a constructor argument (pre_ss_pc.h:20-22); 35 = 5 x 7 is only main.cc's constant.  Windows that are not a
    multiple of 7 exercise the row engine's tail groups, windows wider / taller than the coarse levels its column mask and
    skipped rows, 45 the largest chain-engine layout (5 passes)."""
    gpu_ctx.set_images(odd_pair["l"], odd_pair["r"])
    gpu_ctx.build_cost_grd(odd_pair["max_dis"], wnd, scale_num, lam)
    pc = po.PlaneCost(odd_pair["l"], odd_pair["r"], odd_pair["max_dis"], wnd, scale_num, lam)
    pm = po.PatchMatch(odd_pair["l"], odd_pair["r"], odd_pair["max_dis"], 4)
    pm.run(2, pc, False, seed=wnd, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(2, seed=wnd, schedule=po.SCHED_RASTER)
    _assert_state_equal(gpu_ctx, pm, f"wnd {wnd}")


@pytest.mark.parametrize("cc", ["GRD", "CEN"])
def test_disparity_range_wider_than_the_lds_strip(gpu_ctx, cc):
    """max_dis = 400 on a 150-pixel-wide pair: the other view's row window (64 centres + window + disparity range) exceeds
    the strip a wave can stage (384 slots), so level 0 of the row engine reads both views from global memory, and view
    propagation wraps around the image border (HandleBorder); coarser levels are staged again."""
    from crossscalepatchmatch_amd import synth
    l, r, _, _ = synth.make_pair(150, 24, 40, regions=3, seed=77)
    D = 400
    gpu_ctx.set_images(l, r)
    (gpu_ctx.build_cost_grd if cc == "GRD" else gpu_ctx.build_cost_cen)(D, 35, 3, 0.3)
    pc = po.PlaneCost(l, r, D, 35, 3, 0.3, cc=cc)
    pm = po.PatchMatch(l, r, D, 1)
    pm.run(1, pc, False, seed=8, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(1, seed=8, schedule=po.SCHED_RASTER)
    _assert_state_equal(gpu_ctx, pm, "wide disparity range")


def test_set_planes_makes_the_sweep_distrust_stored_costs(gpu_ctx, small_pair):
    """The raster sweep skips a neighbour whose plane is bitwise the pixel's own -- valid only while every stored cost is the
    cost of the stored plane.  cspm_set_planes can break that (here: all planes equal, all costs huge); the sweep must then
    evaluate, as the reference does, and lower every cost."""
    pc, pm = _setup(gpu_ctx, small_pair, 5, 0.3)
    h, w = small_pair["h"], small_pair["w"]
    for v in (0, 1):
        P = pm.planes(v)
        P[..., 0:2] = 0.0; P[..., 2] = 1.0
        P[..., 3] = np.arange(w)[None, :]; P[..., 4] = np.arange(h)[:, None]; P[..., 5] = 6.5
        P[..., 6:8] = 0.0; P[..., 8] = 6.5
        pm.min_cost(v)[...] = 1e9
        gpu_ctx.set_planes(v, np.concatenate([P[..., 0:3], P[..., 6:9]], -1), pm.min_cost(v))
    pm.spatial(0, pc, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.pm_spatial(0, schedule=po.SCHED_RASTER)
    _assert_state_equal(gpu_ctx, pm, "sweep after set_planes")
    assert pm.min_cost(0)[5, 5] < 1e8


@pytest.mark.parametrize("w,h,D", [(1, 1, 2), (2, 1, 2), (1, 7, 3), (9, 2, 4), (5, 40, 4), (37, 3, 6), (20, 20, 25)])
def test_degenerate_image_sizes(gpu_ctx, w, h, D):
    """Ragged / tiny inputs: single pixels, single rows and columns, windows larger than the image, max_dis > width
    (view propagation would index out of range in the reference; both sides skip those candidates)."""
    rng = np.random.default_rng(w * 100 + h)
    l = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    r = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    for sn, lam in ((0, 0.0), (5, 0.3)):
        for sched in (po.SCHED_RASTER, po.SCHED_REDBLACK):
            gpu_ctx.set_images(l, r)
            gpu_ctx.build_cost_grd(D, 35, sn, lam)
            pc = po.PlaneCost(l, r, D, 35, sn, lam)
            pm = po.PatchMatch(l, r, D, 2)
            pm.run(2, pc, True, seed=3, schedule=sched, sum_order=po.SUM_DEVICE)
            gpu_ctx.patchmatch(2, seed=3, schedule=sched)
            for v in (0, 1):
                npar, cost = gpu_ctx.get_planes(v)
                np.testing.assert_array_equal(npar[..., 3:], pm.planes(v)[..., 6:9])
                np.testing.assert_array_equal(cost, pm.min_cost(v))
            lo, ro = gpu_ctx.postprocess(2)
            np.testing.assert_array_equal(lo, pm.dis(0))
            np.testing.assert_array_equal(ro, pm.dis(1))


def test_device_resident_io_stream_and_timing(gpu_ctx, small_pair):
    """The entry points bench.py and the batch driver use: images already in HBM, a caller-provided HIP stream,
    8-bit maps written to device memory, per-kernel-class hipEvent timing."""
    import torch
    import crossscalepatchmatch_amd as cs
    pc, pm = _setup(gpu_ctx, small_pair, 5, 0.3)
    pm.run(2, pc, False, seed=21, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    dev = torch.device("cuda", 0)
    h, w = small_pair["h"], small_pair["w"]
    d_l, d_r = torch.from_numpy(small_pair["l"]).to(dev), torch.from_numpy(small_pair["r"]).to(dev)
    ctx = cs.StereoContext(0)
    stream = torch.cuda.Stream(device=dev)
    try:
        own = ctx.stream_ptr()
        assert own != 0
        ctx.set_stream(stream.cuda_stream)
        assert ctx.stream_ptr() == stream.cuda_stream
        ctx.enable_timing(True)
        ctx.reset_timing()
        ctx.set_images_device(d_l.data_ptr(), d_r.data_ptr(), w, h, w * 3)
        ctx.build_cost_grd(small_pair["max_dis"], 35, 5, 0.3)
        ctx.patchmatch(2, seed=21, schedule=cs.SCHED_RASTER)
        outs = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
        for v in (0, 1):
            ctx.disparity_u8_device(v, 4, outs[v].data_ptr())
        ctx.synchronize()
        for v in (0, 1):
            np.testing.assert_array_equal(outs[v].cpu().numpy(), pm.dis(v))
        t = ctx.timing()
        assert t["init"]["launches"] == 1 and t["init"]["evals"] == 2 * w * h
        # one launch per iteration runs all halving steps of PlaneRefinement; `evals` counts every candidate plane
        assert t["refine"]["launches"] == 2 and t["refine"]["evals"] == 2 * 2 * w * h * po.lib().csor_refine_steps(small_pair["max_dis"])
        assert t["spatial"]["launches"] == 2 and t["view"]["launches"] == 4
        assert all(t[k]["ms"] > 0 for k in ("grd", "init", "spatial", "view", "refine"))
        assert ctx.taps_per_view_pass() == sum(pc.taps(x, y) for y in range(h) for x in range(w))
        ctx.set_stream(0)
        assert ctx.stream_ptr() == own  # back on the stream the context owns
    finally:
        ctx.close()


def test_pairs_in_flight_do_not_interfere(gpu_ctx, small_pair, mid_pair, odd_pair):
    """bench.py keeps three pairs in flight on one GPU (three contexts, three streams, kernels overlapping on the device; the
    persistent sweeps of different contexts spin next to each other).  Every context must produce what it produces alone."""
    import crossscalepatchmatch_amd as cs
    pairs = [small_pair, mid_pair, odd_pair]
    alone = []
    for p in pairs:
        gpu_ctx.set_images(p["l"], p["r"])
        gpu_ctx.build_cost_grd(p["max_dis"], 35, 5, 0.3)
        gpu_ctx.patchmatch(3, seed=5, schedule=po.SCHED_RASTER)
        alone.append([gpu_ctx.get_planes(v) for v in (0, 1)])
    ctxs = [cs.StereoContext(0) for _ in pairs]
    try:
        for rep in range(2):  # the second round reuses every buffer and the next sweep epochs
            for c, p in zip(ctxs, pairs):  # everything is enqueued, nothing waits
                c.set_images(p["l"], p["r"])
                c.build_cost_grd(p["max_dis"], 35, 5, 0.3)
                c.patchmatch(3, seed=5, schedule=po.SCHED_RASTER)
            for c, want in zip(ctxs, alone):
                for v in (0, 1):
                    npar, cost = c.get_planes(v)
                    np.testing.assert_array_equal(npar, want[v][0])
                    np.testing.assert_array_equal(cost, want[v][1])
    finally:
        for c in ctxs:
            c.close()


def test_sweep_timeout_falls_back_to_per_diagonal_launches(gpu_ctx, mid_pair):
    """A persistent sweep that gives up waiting (CSPM_OPT_SWEEP_TIMEOUT_MS; 0 makes every longer wait fail) is slowness, not a
    wrong result: the synchronising call repeats the one PatchMatch that ran since the last check with per-diagonal launches and
    returns the same planes.  Several unchecked runs cannot be repeated: that is reported as an error."""
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import capi
    ctx = cs.StereoContext(0)
    try:
        ctx.set_images(mid_pair["l"], mid_pair["r"])
        ctx.build_cost_grd(mid_pair["max_dis"], 35, 5, 0.3)
        ctx.patchmatch(2, seed=9, schedule=0)
        want = [ctx.get_planes(v) for v in (0, 1)]
        assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 0 and ctx.get_option(capi.OPT_SWEEP_TIMEOUT_MS) == 3000
        ctx.set_option(capi.OPT_SWEEP_TIMEOUT_MS, 0)
        ctx.patchmatch(2, seed=9, schedule=0)
        got = [ctx.get_planes(v) for v in (0, 1)]  # the getter synchronises, sees the timeout and repeats the run
        assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 1
        for v in (0, 1):
            np.testing.assert_array_equal(got[v][0], want[v][0])
            np.testing.assert_array_equal(got[v][1], want[v][1])
        ctx.patchmatch(2, seed=9, schedule=0)
        ctx.patchmatch(2, seed=10, schedule=0)  # two runs enqueued, nothing checked in between
        with pytest.raises(cs.CspmError, match="timed out"):
            ctx.synchronize()
        ctx.set_option(capi.OPT_SWEEP_TIMEOUT_MS, 3000)
        ctx.patchmatch(2, seed=9, schedule=0)  # the context is usable again
        np.testing.assert_array_equal(ctx.get_planes(0)[1], want[0][1])
    finally:
        ctx.close()


def test_sweep_timeout_behind_asynchronous_outputs(gpu_ctx, mid_pair):
    """The maps a caller asks for BEHIND an asynchronous run (cspm_get_disparity_u8, cspm_disparity_u8_device,
    cspm_postprocess_device, batch.HipPairFn with one pair per context) must come from the planes of the REPEATED run when the
    persistent sweep timed out -- never from the aborted sweep's planes with CSPM_OK."""
    import torch
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import batch, capi
    dev = torch.device("cuda", 0)
    h, w, D = mid_pair["h"], mid_pair["w"], mid_pair["max_dis"]
    ctx = cs.StereoContext(0)
    try:
        ctx.set_images(mid_pair["l"], mid_pair["r"])
        ctx.build_cost_grd(D, 35, 5, 0.3)
        ctx.patchmatch(2, seed=9, schedule=0)
        want_u8 = [ctx.disparity_u8(v, 4) for v in (0, 1)]
        want_pp = ctx.postprocess(4)
        ctx.set_option(capi.OPT_SWEEP_TIMEOUT_MS, 0)
        # host getter: the check comes before PlaneToDisp
        ctx.patchmatch(2, seed=9, schedule=0)
        np.testing.assert_array_equal(ctx.disparity_u8(0, 4), want_u8[0])
        assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 1
        # device-resident map enqueued behind the run, then a synchronising call
        outs = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
        ctx.patchmatch(2, seed=9, schedule=0)
        for v in (0, 1):
            ctx.disparity_u8_device(v, 4, outs[v].data_ptr())
        ctx.synchronize()
        assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 2
        for v in (0, 1):
            np.testing.assert_array_equal(outs[v].cpu().numpy(), want_u8[v])
        # post-processed maps likewise
        ctx.patchmatch(2, seed=9, schedule=0)
        ctx.postprocess_device(4, outs[0].data_ptr(), outs[1].data_ptr())
        ctx.synchronize()
        assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 3
        for v in (0, 1):
            np.testing.assert_array_equal(outs[v].cpu().numpy(), want_pp[v])
    finally:
        ctx.close()
    # the batch driver: one pair per context, every context's sweep times out, finalize() repeats each run and its maps
    fn = batch.HipPairFn(0, in_flight=2)
    try:
        for c in fn.ctxs:
            c.set_option(capi.OPT_SWEEP_TIMEOUT_MS, 0)
        pairs = torch.stack([torch.stack([torch.from_numpy(mid_pair["l"]), torch.from_numpy(mid_pair["r"])])] * 2).to(dev)
        params = dict(w=w, h=h, max_dis=D, dis_scale=4, scale_num=5, reg_lambda=0.3, iters=2, seed=9, schedule=0, use_pp=0, cc=batch.CC_CODES["GRD"])
        maps = batch.run_batch(pairs, params, fn, device="cuda:0", dist=None)  # pair k runs with seed 9 + k
        assert sum(c.get_option(capi.OPT_SWEEP_FALLBACKS) for c in fn.ctxs) == 2
        gpu_ctx.set_images(mid_pair["l"], mid_pair["r"])
        gpu_ctx.build_cost_grd(D, 35, 5, 0.3)
        for k in range(2):
            gpu_ctx.patchmatch(2, seed=9 + k, schedule=0)
            for v in (0, 1):
                np.testing.assert_array_equal(maps[k, v].cpu().numpy(), gpu_ctx.disparity_u8(v, 4))
    finally:
        fn.close()


def test_sweep_row_bands_and_paired_cells(gpu_ctx):
    """Two options of the persistent sweep that change who evaluates what and where the cells come from, never the result:
    CSPM_SWEEP_BANDS (one claim queue per row band / XCD) and CSPM_OPT_SWEEP_PAIRS (paired-cell volumes) -- alone and together,
    on an image with more rows than bands and several workgroups per band: identical planes and costs."""
    import os
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import capi, synth
    w, h, D = 200, 150, 24
    l, r, _, _ = synth.make_pair(w, h, D, regions=4, seed=404)
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(D, 35, 5, 0.3)
    gpu_ctx.patchmatch(2, seed=3, schedule=0)
    want = [gpu_ctx.get_planes(v) for v in (0, 1)]
    for bands, pairs in ((8, False), (3, True), (8, True), (1, True)):
        os.environ["CSPM_SWEEP_BANDS"] = str(bands)
        try:
            ctx = cs.StereoContext(0)
        finally:
            del os.environ["CSPM_SWEEP_BANDS"]
        try:
            ctx.set_images(l, r)
            ctx.build_cost_grd(D, 35, 5, 0.3, sweep_pairs=pairs)
            assert ctx.get_option(capi.OPT_SWEEP_PAIRS_ACTIVE) == int(pairs)
            ctx.patchmatch(2, seed=3, schedule=0)
            for v in (0, 1):
                npar, cost = ctx.get_planes(v)
                np.testing.assert_array_equal(npar, want[v][0], err_msg=f"bands {bands} pairs {pairs} view {v}")
                np.testing.assert_array_equal(cost, want[v][1], err_msg=f"bands {bands} pairs {pairs} view {v}")
            assert ctx.get_option(capi.OPT_SWEEP_FALLBACKS) == 0
        finally:
            ctx.close()


@pytest.mark.parametrize("w,h,D", [(1, 1, 2), (9, 2, 4), (37, 3, 6), (77, 41, 21), (200, 70, 24)])
def test_row_kernels_claimed_column_bands(gpu_ctx, w, h, D):
    """The row kernels hand their items out in two ways (cspm_rows.h row_item): interleaved row blocks, and column bands claimed from
    eight counters with 25 % surplus workgroups -- by default only launches of several rounds of resident waves (k_init / k_refine of
    a KITTI-size pair) take the latter.  CSPM_ROW_CLAIM=1 forces it on small and ragged images: every pixel must still be evaluated
    exactly once -- planes, costs and post-processed maps equal the default context's."""
    import os
    import crossscalepatchmatch_amd as cs
    rng = np.random.default_rng(w * 1000 + h)
    l = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    r = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    os.environ["CSPM_ROW_CLAIM"] = "1"
    try:
        ctx = cs.StereoContext(0)
    finally:
        del os.environ["CSPM_ROW_CLAIM"]
    try:
        for sn, lam in ((0, 0.0), (5, 0.3)):
            out = []
            for c in (gpu_ctx, ctx):
                c.set_images(l, r)
                c.build_cost_grd(D, 35, sn, lam)
                c.patchmatch(2, seed=3, schedule=0)
                out.append([c.get_planes(v) for v in (0, 1)] + list(c.postprocess(2)))
            for v in (0, 1):
                np.testing.assert_array_equal(out[0][v][0], out[1][v][0])
                np.testing.assert_array_equal(out[0][v][1], out[1][v][1])
                np.testing.assert_array_equal(out[0][2 + v], out[1][2 + v])
    finally:
        ctx.close()


def test_new_cost_object_withdraws_the_stored_costs(gpu_ctx, small_pair):
    """The sweep skips a neighbour plane that is bitwise the pixel's own plane (its cost would be the stored min_cost) -- only
    while every stored cost was computed by the CURRENT cost object.  Rebuilding the cost (census -> GRD here) on the same
    plane field must make the sweep evaluate again, as the reference's loop would (cs_patchmatch.cc:163-216)."""
    l, r, D = small_pair["l"], small_pair["r"], small_pair["max_dis"]
    pc_cen = po.PlaneCost(l, r, D, 35, 5, 0.3, cc="CEN")
    pc_grd = po.PlaneCost(l, r, D, 35, 5, 0.3)
    pm = po.PatchMatch(l, r, D, 4)
    kw = dict(seed=17, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_cen(D, 35, 5, 0.3)
    pm.init(pc_cen, **kw); gpu_ctx.pm_init(seed=17)
    pm.spatial(0, pc_cen, **kw); gpu_ctx.pm_spatial(0, seed=17)  # propagation leaves runs of bitwise identical planes
    _assert_state_equal(gpu_ctx, pm, "census sweep")
    gpu_ctx.build_cost_grd(D, 35, 5, 0.3)                          # same images, same plane field, another cost
    pm.spatial(1, pc_grd, **kw); gpu_ctx.pm_spatial(1, seed=17)
    _assert_state_equal(gpu_ctx, pm, "GRD sweep over census costs")


def test_wide_disparity_range(gpu_ctx):
    """max_dis = 300: the level-0 strips would be wider than the 384 slots a wave stages (that level reads global memory), the
    other levels stage 2 x 384-slot strip sets per wave and the row kernels need more than the 64 KB of dynamic LDS a launch gets
    without opting in.  Whole pipeline == the oracle."""
    from crossscalepatchmatch_amd import synth
    w, h, D = 420, 20, 300
    l, r, _, _ = synth.make_pair(w, h, D, regions=3, seed=77)
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(D, 35, 5, 0.3)
    pc = po.PlaneCost(l, r, D, 35, 5, 0.3)
    pm = po.PatchMatch(l, r, D, 1)
    pm.run(1, pc, False, seed=8, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    gpu_ctx.patchmatch(1, seed=8, schedule=0)
    _assert_state_equal(gpu_ctx, pm, "max_dis 300")


@pytest.mark.parametrize("scale_num", [5, 4])
def test_folded_sweep_workgroups_equal_the_oracle(gpu_ctx, odd_pair, mid_pair, scale_num):
    """CSPM_OPT_SWEEP_FOLD (round 6, what callers with several pairs in flight switch on): sweep workgroups of levels - 1 waves, the
    coarsest level's window passes shared by the waves of levels 1.. -- the persistent sweep, the per-diagonal launches and the whole
    pipeline give the oracle's planes bit for bit, with five levels (four waves) and with four (three waves)."""
    from crossscalepatchmatch_amd import capi
    for pair in (odd_pair, mid_pair):
        pc, pm = _setup(gpu_ctx, pair, scale_num, 0.3)
        try:
            gpu_ctx.set_option(capi.OPT_SWEEP_FOLD, 1)
            pm.init(pc, seed=3, sum_order=po.SUM_DEVICE); gpu_ctx.pm_init(seed=3)
            for it in (0, 1):
                for launches in (0, 1):
                    gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, launches)
                    start = [gpu_ctx.get_planes(v) for v in (0, 1)]
                    gpu_ctx.pm_spatial(it, seed=3)
                    if launches == 0:
                        first = [gpu_ctx.get_planes(v) for v in (0, 1)]
                        for v in (0, 1):
                            gpu_ctx.set_planes(v, *start[v])
                    else:
                        for v in (0, 1):
                            np.testing.assert_array_equal(gpu_ctx.get_planes(v)[0], first[v][0])
                            np.testing.assert_array_equal(gpu_ctx.get_planes(v)[1], first[v][1])
                pm.spatial(it, pc, seed=3, sum_order=po.SUM_DEVICE)
                _assert_state_equal(gpu_ctx, pm, f"folded sweep, {scale_num} levels, iteration {it}")
            gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)
            pm.run(2, pc, False, seed=8, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
            gpu_ctx.patchmatch(2, seed=8, schedule=po.SCHED_RASTER)
            _assert_state_equal(gpu_ctx, pm, f"folded sweep, {scale_num} levels, whole run")
        finally:
            gpu_ctx.set_option(capi.OPT_SWEEP_FOLD, 0)
            gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)
