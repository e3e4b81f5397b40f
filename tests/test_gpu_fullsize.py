"""Full-size runs (BASELINE.json configs) on the GPU: the oracle cannot do a whole KITTI-size pair in seconds,
so parity at these sizes is checked (a) exactly, on sampled plane evaluations against the oracle, and (b) through
size-independent properties of the algorithm: determinism, monotone costs, stored cost == re-evaluated cost,
early exit on/off and fused/volume cost sources giving identical plane fields."""
import numpy as np
import pytest

from conftest import random_planes
from crossscalepatchmatch_amd import synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    cfg, l, r, gl, gr = synth.make_config("C3")
    return cfg, l, r, gl, gr


def test_c3_sampled_plane_costs_equal_the_oracle(gpu_ctx, c3):
    cfg, l, r, _, _ = c3
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])  # ~1.1 GB of f64 volumes on the host
    assert [gpu_ctx.level_dims(s) for s in range(5)] == [(1242, 375, 128), (621, 188, 64), (311, 94, 32), (156, 47, 16), (78, 24, 8)]
    np.testing.assert_array_equal(gpu_ctx.scale_weights(), pc.scale_wgt())
    rng = np.random.default_rng(17)
    for v in (0, 1):
        for s in range(5):
            assert gpu_ctx.max_cost(v, s) == pc.max_cost_dev(v, s)
        np.testing.assert_array_equal(gpu_ctx.cost_slab(v, 0, 77), pc.volume_dev(v, 0)[77])
        np.testing.assert_array_equal(gpu_ctx.cost_slab(v, 2, 32), pc.volume_dev(v, 2)[32])
        xy, norm, point, param = random_planes(rng, 400, cfg["w"], cfg["h"], cfg["max_dis"])
        got = gpu_ctx.plane_cost_batch(v, xy, np.concatenate([norm, param], 1))
        want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], v, po.SUM_DEVICE) for i in range(len(xy))])
        np.testing.assert_array_equal(got, want)


def _state(ctx):
    return [ctx.get_planes(v) for v in (0, 1)]


def test_c3_properties(gpu_ctx, c3):
    cfg, l, r, gl, gr = c3
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    kw = dict(seed=2024, schedule=0)
    # monotone: no phase may increase any pixel's cost
    gpu_ctx.pm_init(**kw)
    prev = [c for _, c in _state(gpu_ctx)]
    for phase in ("pm_spatial", "pm_view", "pm_refine"):
        getattr(gpu_ctx, phase)(0, **kw)
        cur = [c for _, c in _state(gpu_ctx)]
        for v in (0, 1):
            assert np.all(cur[v] <= prev[v]), phase
            assert np.any(cur[v] < prev[v]), phase
        prev = cur
    # whole run, twice: deterministic
    gpu_ctx.patchmatch(3, **kw)
    a = _state(gpu_ctx)
    dis_a = [gpu_ctx.disparity_u8(v, cfg["dis_scale"]) for v in (0, 1)]
    gpu_ctx.patchmatch(3, **kw)
    b = _state(gpu_ctx)
    for v in (0, 1):
        np.testing.assert_array_equal(a[v][0], b[v][0])
        np.testing.assert_array_equal(a[v][1], b[v][1])
    # stored min_cost == cost of the stored plane, re-evaluated (no early exit), on a sample of pixels
    rng = np.random.default_rng(3)
    for v in (0, 1):
        npar, cost = a[v]
        ys, xs = rng.integers(0, cfg["h"], 3000), rng.integers(0, cfg["w"], 3000)
        got = gpu_ctx.plane_cost_batch(v, np.stack([xs, ys], 1), npar[ys, xs])
        np.testing.assert_array_equal(got, cost[ys, xs])
        # unit normals, parameters consistent with normals
        n = npar[..., :3]
        np.testing.assert_allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-9)
        # 8-bit map == saturate(round_half_even(d * dis_scale))
        d = gpu_ctx.disparity_f64(v)
        np.testing.assert_array_equal(dis_a[v], np.clip(np.rint(d * cfg["dis_scale"]), 0, 255).astype(np.uint8))
    # early exit off: identical plane field
    gpu_ctx.patchmatch(3, early_exit=0, **kw)
    c = _state(gpu_ctx)
    for v in (0, 1):
        np.testing.assert_array_equal(a[v][0], c[v][0])
        np.testing.assert_array_equal(a[v][1], c[v][1])
    # the result is a disparity map of the scene, not noise (noise floor of this synthetic pair is ~6 %)
    assert synth.bad_fraction(gpu_ctx.disparity_f64(0), gl, 2.0) < 0.12
    assert synth.bad_fraction(gpu_ctx.disparity_f64(1), gr, 2.0) < 0.12
    # materialised f64 cost volumes (the reference's data flow) give the same plane field as fused cells
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=True)
    gpu_ctx.patchmatch(3, **kw)
    e = _state(gpu_ctx)
    for v in (0, 1):
        np.testing.assert_array_equal(a[v][0], e[v][0])
        np.testing.assert_array_equal(a[v][1], e[v][1])
    gpu_ctx.build_cost_grd(16, 35, 0, 0.0)  # drop the 1.1 GB volumes


def test_c1_single_scale_config(gpu_ctx):
    """BASELINE.json configs[0]: 450x375, max_dis=60, GRD, use_cs=false, use_pp=false -- sampled exact costs + properties."""
    cfg, l, r, gl, _ = synth.make_config("C1")
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, 0, 0.0)
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, 0, 0.0)
    rng = np.random.default_rng(5)
    xy, norm, point, param = random_planes(rng, 500, cfg["w"], cfg["h"], cfg["max_dis"])
    got = gpu_ctx.plane_cost_batch(0, xy, np.concatenate([norm, param], 1))
    want = np.array([pc.cost(xy[i, 0], xy[i, 1], norm[i], param[i], 0, po.SUM_DEVICE) for i in range(500)])
    np.testing.assert_array_equal(got, want)
    gpu_ctx.patchmatch(3, seed=1, schedule=0)
    npar, cost = gpu_ctx.get_planes(0)
    ys, xs = rng.integers(0, cfg["h"], 1000), rng.integers(0, cfg["w"], 1000)
    want = np.array([pc.cost(xs[i], ys[i], npar[ys[i], xs[i], :3], npar[ys[i], xs[i], 3:], 0, po.SUM_DEVICE) for i in range(1000)])
    np.testing.assert_array_equal(cost[ys, xs], want)  # the oracle agrees with every stored cost it is asked about
    assert synth.bad_fraction(gpu_ctx.disparity_f64(0), gl, 2.0) < 0.25


def test_c2_whole_pipeline_bit_exact(gpu_ctx):
    """BASELINE.json configs[1]: teddy-size 450x375, max_dis=60, GRD, use_cs=true (5 levels, lambda 0.3) -- the WHOLE
    pipeline (init + 3 x (raster sweep, view propagation, refinement) + PlaneToDisp) through the C ABI, every plane, cost
    and disparity compared with the oracle (device summation order).  ~5.9e10 window taps on the host: a minute or two."""
    cfg, l, r, gl, gr = synth.make_config("C2")
    assert (cfg["w"], cfg["h"], cfg["max_dis"], cfg["scale_num"]) == (450, 375, 60, 5)
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    assert [gpu_ctx.level_dims(s) for s in range(5)] == [(450, 375, 60), (225, 188, 30), (113, 94, 15), (57, 47, 7), (29, 24, 3)]
    gpu_ctx.patchmatch(3, seed=12345, schedule=0)
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)  # the oracle's sweep as a wavefront: identical, faster
    for v in (0, 1):
        npar, cost = gpu_ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3])
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9])
        np.testing.assert_array_equal(cost, pm.min_cost(v))
        np.testing.assert_array_equal(gpu_ctx.disparity_u8(v, cfg["dis_scale"]), pm.dis(v))
        np.testing.assert_array_equal(gpu_ctx.disparity_f64(v), pm.disp_f64(v))
    assert synth.bad_fraction(gpu_ctx.disparity_f64(0), gl, 2.0) < 0.2
    # ... and the north-star bar on the same run (one GPU run, one oracle cost object for both legs): the oracle in the REFERENCE order
    # (serial raster sweep, serial window sum, no FMA), identical inputs and random numbers, >= 99.5 % of both views within 0.5 px
    pm2 = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm2.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL, wavefront=True)  # C1's leg keeps the sweep serial
    for v in (0, 1):
        d = np.abs(gpu_ctx.disparity_f64(v) - pm2.disp_f64(v))
        assert float(np.mean(d <= 0.5)) >= 0.995, (v, float(np.mean(d <= 0.5)), float(d.max()))


@pytest.mark.parametrize("name", ["C1"])
def test_north_star_bar_full_c1_c2(gpu_ctx, name):
    """The north-star bar at BASELINE.json's own sizes (C2's leg lives in test_c2_whole_pipeline_bit_exact, which shares its GPU run and
    oracle cost object): the WHOLE of C1 (single scale) and C2 (cross-scale, 5 levels) on the GPU
    (device order: ROWTREE7 + contracted multiply-adds) against the CPU oracle in the REFERENCE order (serial raster sweep, serial
    window sum, no FMA) on identical inputs and identical random numbers: >= 99.5 % of the pixels of both views within 0.5 px.
    (~15 s for C1, ~50 s for C2 on the GPU box's 16 usable cores.)"""
    cfg, l, r, _, _ = synth.make_config(name)
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    gpu_ctx.patchmatch(3, seed=12345, schedule=0)
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL)  # threads: pyoracle's default = the CPUs the container may use
    for v in (0, 1):
        d = np.abs(gpu_ctx.disparity_f64(v) - pm.disp_f64(v))
        within = float(np.mean(d <= 0.5))
        assert within >= 0.995, (name, v, within, float(d.max()))


def test_c5_full_resolution_three_iterations_and_postprocessing(gpu_ctx):
    """BASELINE.json configs[4]: 3000x2000, max_dis=256, cross-scale, use_pp=true, 3 iterations.  f64 volumes would be 28 GB; the
    fused cost needs ~0.3 GB.  The oracle cannot run PatchMatch at this size, but PostProcessing (cs_patchmatch.cc:508-588) is cheap
    on the CPU: the GPU's OWN final plane field is handed to the oracle, which runs PlaneToDisp + LeftRightCheck + FillInvalid +
    WeightedMedian on it -- the 8-bit maps of cspm_postprocess must be identical at full resolution (rows wider than one scan
    block of k_fill_rows, > 64 K inconsistent pixels for k_weighted_median's work list).  Plus self-consistency of stored costs."""
    cfg, l, r, gl, _ = synth.make_config("C5")
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    assert gpu_ctx.level_dims(4) == (188, 125, 16)
    gpu_ctx.patchmatch(3, seed=7, schedule=0)
    fields = [gpu_ctx.get_planes(v) for v in (0, 1)]
    npar, cost = fields[0]
    rng = np.random.default_rng(9)
    ys, xs = rng.integers(0, cfg["h"], 2000), rng.integers(0, cfg["w"], 2000)
    got = gpu_ctx.plane_cost_batch(0, np.stack([xs, ys], 1), npar[ys, xs])
    np.testing.assert_array_equal(got, cost[ys, xs])
    raw = [gpu_ctx.disparity_u8(v, cfg["dis_scale"]) for v in (0, 1)]
    lo, ro = gpu_ctx.postprocess(cfg["dis_scale"])
    assert lo.shape == (2000, 3000)
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    for v in (0, 1):
        P = pm.planes(v)  # 9 doubles per pixel: norm, point, param -- PlaneToDisp reads the parameters only
        P[..., 0:3] = fields[v][0][..., 0:3]
        P[..., 6:9] = fields[v][0][..., 3:6]
    pm.plane_to_disp()
    for v in (0, 1):
        np.testing.assert_array_equal(raw[v], pm.dis(v), err_msg=f"PlaneToDisp, view {v}")
    pm.postprocess()
    np.testing.assert_array_equal(lo, pm.dis(0), err_msg="post-processed left map")
    np.testing.assert_array_equal(ro, pm.dis(1), err_msg="post-processed right map")
    changed = float(np.mean(lo != raw[0]))
    assert changed > 0.005 and int(np.sum(lo != raw[0])) > 65536, changed  # not vacuous: far more than 64 K pixels were rewritten
    assert synth.bad_fraction(lo.astype(np.float64) / cfg["dis_scale"], gl, 2.0) < 0.2
    gpu_ctx.set_images(np.zeros((8, 8, 3), np.uint8), np.zeros((8, 8, 3), np.uint8))  # release the big buffers


def test_c3_whole_pair_bit_exact(gpu_ctx, c3):
    """The headline configuration end to end (BASELINE.json configs[2]: 1242x375, max_dis 128, GRD, 5 levels, lambda 0.3, 3 iterations,
    raster sweeps): one WHOLE pair through the HIP path and through the CPU oracle in the same device order -- all 2 x 465 750
    planes, every stored cost, both 8-bit maps and the unquantised disparities identical.  ~1.9e11 window taps on the host: three to
    five minutes on the GPU box's 16 usable cores (the oracle's raster sweep runs as an anti-diagonal wavefront here, which
    tests/test_oracle_primitives.py shows equal to its serial loop)."""
    cfg, l, r, gl, _ = c3
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    gpu_ctx.patchmatch(3, seed=12345, schedule=0)
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
    for v in (0, 1):
        npar, cost = gpu_ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3], err_msg=f"normals, view {v}")
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9], err_msg=f"plane parameters, view {v}")
        np.testing.assert_array_equal(cost, pm.min_cost(v), err_msg=f"stored costs, view {v}")
        np.testing.assert_array_equal(gpu_ctx.disparity_u8(v, cfg["dis_scale"]), pm.dis(v), err_msg=f"8-bit map, view {v}")
        np.testing.assert_array_equal(gpu_ctx.disparity_f64(v), pm.disp_f64(v))
    assert synth.bad_fraction(gpu_ctx.disparity_f64(0), gl, 2.0) < 0.2
    # The north-star bar on the same WHOLE pair (round-4 review, Weak 3; merged into this test in round 6 so that the GPU runs once and the
    # two oracle legs share the 1.1 GB cost object): the CPU oracle in the REFERENCE order (serial window sum, no FMA, the reference's
    # raster traversal -- walked anti-diagonal by anti-diagonal, `wavefront`: the same in-place serial result,
    # tests/test_oracle_primitives.py), identical inputs and random numbers: >= 99.5 % of all 2 x 465 750 pixels within 0.5 px.
    # bench.py's cpu_baseline leg repeats this comparison with the sweep serial, as the reference runs it.
    pm2 = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm2.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL, wavefront=True)
    for v in (0, 1):
        d = np.abs(gpu_ctx.disparity_f64(v) - pm2.disp_f64(v))
        within = float(np.mean(d <= 0.5))
        assert d.shape == (375, 1242)
        assert within >= 0.995, (v, within, float(d.max()))


def test_c5_shaped_crop_d256_whole_pipeline_bit_exact(gpu_ctx):
    """BASELINE.json configs[4]'s shape, oracle-checked (round-4 review, Weak 4: C5's PatchMatch was only self-checked): a 704x160
    window of the C5 pair at its full disparity range -- max_dis 256, 5 levels, lambda 0.3, use_pp -- through the whole pipeline
    against the oracle in the device order: 356-slot level-0 strips (13.8 KB of LDS per wave, the LDS opt-in), DMA-filled tables
    from D = 256 volumes, ranges that span up to 256 disparities, 11 halving steps of refinement, then post-processing.
    array_equal on planes, stored costs, raw and post-processed maps.  ~1.5e10 window taps on the host."""
    from crossscalepatchmatch_amd import capi
    cfg, l, r, _, _ = synth.make_config("C5")
    y0, x0, w, h = 600, 1000, 704, 160
    lc, rc = np.ascontiguousarray(l[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(r[y0:y0 + h, x0:x0 + w])
    D = cfg["max_dis"]
    assert D == 256 and cfg["scale_num"] == 5 and cfg["use_pp"]
    gpu_ctx.set_images(lc, rc)
    gpu_ctx.build_cost_grd(D, 35, cfg["scale_num"], cfg["reg_lambda"])
    assert gpu_ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE) == 1
    assert [gpu_ctx.level_dims(s)[2] for s in range(5)] == [256, 128, 64, 32, 16]
    gpu_ctx.patchmatch(2, seed=7, schedule=0)
    pc = po.PlaneCost(lc, rc, D, 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(lc, rc, D, cfg["dis_scale"])
    pm.run(2, pc, False, seed=7, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
    for v in (0, 1):
        npar, cost = gpu_ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3], err_msg=f"normals, view {v}")
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9], err_msg=f"plane parameters, view {v}")
        np.testing.assert_array_equal(cost, pm.min_cost(v), err_msg=f"stored costs, view {v}")
        np.testing.assert_array_equal(gpu_ctx.disparity_u8(v, cfg["dis_scale"]), pm.dis(v), err_msg=f"8-bit map, view {v}")
    pm.postprocess()
    lo, ro = gpu_ctx.postprocess(cfg["dis_scale"])
    np.testing.assert_array_equal(lo, pm.dis(0), err_msg="post-processed left map")
    np.testing.assert_array_equal(ro, pm.dis(1), err_msg="post-processed right map")
    # the same crop with computed tables (what a pair too large for the volumes gets): identical
    want = [gpu_ctx.get_planes(v) for v in (0, 1)]
    gpu_ctx.build_cost_grd(D, 35, cfg["scale_num"], cfg["reg_lambda"], table_volumes=False)
    gpu_ctx.patchmatch(2, seed=7, schedule=0)
    for v in (0, 1):
        npar, cost = gpu_ctx.get_planes(v)
        np.testing.assert_array_equal(npar, want[v][0])
        np.testing.assert_array_equal(cost, want[v][1])
    gpu_ctx.build_cost_grd(D, 35, cfg["scale_num"], cfg["reg_lambda"], table_volumes=True)  # leave the shared context on its default


def test_c3_computed_tables_equal_dma_filled_tables(gpu_ctx, c3):
    """The row kernels' cell tables come from the device-cell volumes by LDS-DMA (the default when the volumes fit) or are computed
    in LDS (pairs too large for the volumes, CSPM_OPT_TABLE_VOLUMES = 0): the same cells either way -- at the headline size, over
    three iterations, every plane and cost identical."""
    cfg, l, r, _, _ = c3
    out = []
    for tv in (True, False):
        gpu_ctx.set_images(l, r)
        gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], table_volumes=tv)
        gpu_ctx.patchmatch(3, seed=4711, schedule=0)
        out.append([gpu_ctx.get_planes(v) for v in (0, 1)])
    for v in (0, 1):
        np.testing.assert_array_equal(out[0][v][0], out[1][v][0])
        np.testing.assert_array_equal(out[0][v][1], out[1][v][1])


def test_c3_persistent_sweep_vs_per_diagonal_launches(gpu_ctx, c3):
    """Full-size stress of the inter-workgroup hand-off (per-pixel done flags, agent-scope atomics across the 8 XCDs):
    the persistent sweep must give, repeatedly, exactly the plane field of the one-launch-per-diagonal sweep, whose
    ordering is enforced by kernel boundaries."""
    from crossscalepatchmatch_amd import capi
    cfg, l, r, _, _ = c3
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    kw = dict(seed=77, schedule=0)
    try:
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 1)
        gpu_ctx.pm_init(**kw)
        start = [gpu_ctx.get_planes(v) for v in (0, 1)]
        for it in (0, 1):
            gpu_ctx.pm_spatial(it, **kw)
        want = [gpu_ctx.get_planes(v) for v in (0, 1)]
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)
        for rep in range(2):
            for v in (0, 1):
                gpu_ctx.set_planes(v, *start[v])
            for it in (0, 1):
                gpu_ctx.pm_spatial(it, **kw)
            for v in (0, 1):
                got = gpu_ctx.get_planes(v)
                np.testing.assert_array_equal(got[0], want[v][0], err_msg=f"repetition {rep}, view {v}")
                np.testing.assert_array_equal(got[1], want[v][1], err_msg=f"repetition {rep}, view {v}")
    finally:
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)


def test_c3_folded_sweep_equals_one_wave_per_level(gpu_ctx, c3):
    """CSPM_OPT_SWEEP_FOLD at the headline size (what bench.py, batch.HipPairFn and cspm_main's batch workers run with several pairs in
    flight): four-wave sweep workgroups, the 78 x 24 level's three window passes shared by the waves of levels 1-3 -- three whole
    iterations identical to the one-wave-per-level sweep, plane for plane."""
    from crossscalepatchmatch_amd import capi
    cfg, l, r, _, _ = c3
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    out = []
    try:
        for fold in (0, 1):
            gpu_ctx.set_option(capi.OPT_SWEEP_FOLD, fold)
            gpu_ctx.patchmatch(3, seed=77, schedule=0)
            out.append([gpu_ctx.get_planes(v) for v in (0, 1)])
    finally:
        gpu_ctx.set_option(capi.OPT_SWEEP_FOLD, 0)
    for v in (0, 1):
        np.testing.assert_array_equal(out[0][v][0], out[1][v][0])
        np.testing.assert_array_equal(out[0][v][1], out[1][v][1])


def test_c3_view_propagation_target_order_equals_source_order(gpu_ctx, c3):
    """ViewPropagation's proposals evaluated in the order of their target column (round 6, CSPM_OPT_VIEW_SORT = 1, the default) or of
    their source column: the accept rule is applied per target pixel afterwards (k_view_resolve), so three whole iterations at the
    headline size give identical planes and costs -- and the phase alone too, from one random field."""
    from crossscalepatchmatch_amd import capi
    cfg, l, r, _, _ = c3
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    out = []
    try:
        for vs in (1, 0):
            gpu_ctx.set_option(capi.OPT_VIEW_SORT, vs)
            gpu_ctx.pm_init(seed=5)
            gpu_ctx.pm_view(0, seed=5)
            first = [gpu_ctx.get_planes(v) for v in (0, 1)]
            gpu_ctx.patchmatch(3, seed=4711, schedule=0)
            out.append((first, [gpu_ctx.get_planes(v) for v in (0, 1)]))
    finally:
        gpu_ctx.set_option(capi.OPT_VIEW_SORT, 1)
    for k in (0, 1):
        for v in (0, 1):
            np.testing.assert_array_equal(out[0][k][v][0], out[1][k][v][0])
            np.testing.assert_array_equal(out[0][k][v][1], out[1][k][v][1])
