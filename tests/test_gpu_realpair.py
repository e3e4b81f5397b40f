"""A real photograph through everything (round-5 review, Missing 3: every other test, the goldens, smoke and bench.py draw their
images from synth.py).  The Middlebury-2014 "Motorcycle" pair that scikit-image ships in this image (741x500 RGB + ground truth,
SURVEY.md 8(c)) and the half-size 200x128 crop of it committed under tests/data/ (tests/data/make_real_pair.py) -- large
saturated highlights next to fine texture, photographic gradients, occlusions, a true range of ~60 px: what neither generator
of synth.py produces at once.  The reference's only entry point reads PNGs (main.cc:68-69) and writes 8-bit maps
(main.cc:131-134): the same here, through the command line AND through the C ABI, against the oracle in the device order (bit for
bit) and in the reference order (the north-star bar), with the accuracy against the real ground truth printed and written to
gpurun_out/real_pair.json.

The committed crop can never silently vanish; the full pair is skipped LOUDLY when scikit-image's data directory is absent."""
import json
import os
import subprocess

import numpy as np
import pytest

from crossscalepatchmatch_amd import realdata as rd
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "crossscalepatchmatch_amd", "host")
CLI = os.path.join(ROOT, "crossscalepatchmatch_amd", "cspm_main")


def _pil_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _pil_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"Pf"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0  # little-endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w)[::-1]


def _record(key, value):
    """accuracy figures next to the test log: gpurun_out/ is merged back from the GPU box"""
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "real_pair.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print(f"[real pair] {key}: {value}")


@pytest.fixture(scope="module")
def io_check():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_io_check")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "helpers", "host_io_check.cc"),
                           os.path.join(HOST, "image_io.cc"), "-lz"])
    return exe


def _files_under_test():
    files = rd.crop_files()[:2]
    full = rd.full_files()
    return files + (full[:2] if full else [])


def test_committed_crop_is_there_and_is_the_documented_window():
    """the fixture cannot vanish: three files, 200x128, and -- where scikit-image's pair is on the machine -- exactly columns
    [80, 280) x rows [70, 198) of the half-size pair"""
    cfg, l, r, gt = rd.load_crop()
    assert l.shape == r.shape == (128, 200, 3) and gt.shape == (128, 200) and l.dtype == np.uint8
    assert 0.9 < float(np.mean(np.isfinite(gt))) <= 1.0 and 4.0 < np.nanmin(gt) < np.nanmax(gt) < cfg["max_dis"]
    assert len(np.unique(l)) > 200 and not np.array_equal(l, r)  # a photograph, not a constant
    half = rd.load_half()
    if half is not None:
        np.testing.assert_array_equal(l, half[1][70:198, 80:280])
        np.testing.assert_array_equal(r, half[2][70:198, 80:280])
        assert np.nanmax(np.abs(gt - half[3][70:198, 80:280])) <= 1 / 256.0


def test_host_image_io_decodes_foreign_pngs_like_pil(io_check, tmp_path):
    """host/image_io.cc (the product's imread, main.cc:68-69) on PNGs it did not write itself -- PIL's encoder for the committed
    crop, the original Middlebury files where present (adaptive filters, 8-bit RGB with a pHYs chunk): the decoded pixels are PIL's."""
    from pngio import read_pnm
    files = _files_under_test()
    assert len(files) >= 2
    for k, f in enumerate(files):
        oc, og = str(tmp_path / f"c{k}.ppm"), str(tmp_path / f"g{k}.pgm")
        subprocess.check_call([io_check, f, oc, og], stdout=subprocess.DEVNULL)
        want = _pil_rgb(f)
        np.testing.assert_array_equal(read_pnm(oc), want, err_msg=f)
        np.testing.assert_array_equal(read_pnm(og), want[..., 1], err_msg=f)
    if rd.full_files() is None:
        print("NOTE: scikit-image's motorcycle pair is not on this machine; only the committed crop was decoded")


def _gpu_run(ctx, cc, cfg, l, r, iters, seed):
    ctx.set_images(l, r)
    if cc == "GRD":
        ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    elif cc == "CEN":
        ctx.build_cost_cen(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    else:
        ctx.build_cost_img(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    ctx.patchmatch(iters, seed=seed, schedule=0)


def _assert_equals_oracle(ctx, pm, cfg, tag):
    """planes, stored costs, raw 8-bit maps, unquantised disparities; then both post-processed maps"""
    for v in (0, 1):
        npar, cost = ctx.get_planes(v)
        P = pm.planes(v)
        np.testing.assert_array_equal(npar[..., :3], P[..., 0:3], err_msg=f"{tag}: normals, view {v}")
        np.testing.assert_array_equal(npar[..., 3:], P[..., 6:9], err_msg=f"{tag}: plane parameters, view {v}")
        np.testing.assert_array_equal(cost, pm.min_cost(v), err_msg=f"{tag}: stored costs, view {v}")
        np.testing.assert_array_equal(ctx.disparity_u8(v, cfg["dis_scale"]), pm.dis(v), err_msg=f"{tag}: 8-bit map, view {v}")
        np.testing.assert_array_equal(ctx.disparity_f64(v), pm.disp_f64(v), err_msg=f"{tag}: disparities, view {v}")
    pm.postprocess()
    lo, ro = ctx.postprocess(cfg["dis_scale"])
    np.testing.assert_array_equal(lo, pm.dis(0), err_msg=f"{tag}: post-processed left map")
    np.testing.assert_array_equal(ro, pm.dis(1), err_msg=f"{tag}: post-processed right map")
    return lo, ro


@pytest.mark.gpu
@pytest.mark.parametrize("cc,scale_num", [("GRD", 5), ("GRD", 0), ("CEN", 5), ("CEN", 0), ("IMG", 5), ("IMG", 0)])
def test_real_crop_every_cost_bit_exact(gpu_ctx, cc, scale_num):
    """the committed crop through GRD, census and GrdPC / CSPC, single- and cross-scale, 3 iterations + post-processing:
    identical to the oracle in the device order; accuracy against the real ground truth on record"""
    cfg, l, r, gt = rd.load_crop()
    cfg["scale_num"], cfg["reg_lambda"] = scale_num, 0.3 if scale_num else 0.0
    _gpu_run(gpu_ctx, cc, cfg, l, r, 3, 12345)
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, scale_num, cfg["reg_lambda"], cc=cc)
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
    lo, _ = _assert_equals_oracle(gpu_ctx, pm, cfg, f"{cc}/{scale_num}")
    bad_raw = rd.bad_fraction(gpu_ctx.disparity_f64(0), gt, 2.0)
    bad_pp = rd.bad_fraction(lo.astype(np.float64) / cfg["dis_scale"], gt, 2.0)
    _record(f"crop_200x128_D32_{cc}_{'cs' if scale_num else 'ss'}_bad2", {"raw": bad_raw, "post_processed": bad_pp})
    assert bad_pp < 0.35, (cc, scale_num, bad_pp)  # a disparity map of the motorcycle, not noise (GRD cross-scale: ~0.14)


@pytest.mark.gpu
def test_real_crop_cli_png_in_png_out(gpu_ctx, tmp_path):
    """cspm_main on the committed PNGs (decoded by host/image_io.cc) == the C ABI on PIL's pixels: the CLI's maps and float PFM"""
    cfg, l, r, _ = rd.load_crop()
    lf, rf, _ = rd.crop_files()
    subprocess.check_call([CLI, f"--l_img_file={lf}", f"--r_img_file={rf}", f"--l_dis_file={tmp_path}/ld.png", f"--r_dis_file={tmp_path}/rd.png",
                           f"--l_disp_pfm={tmp_path}/l.pfm", f"--max_dis={cfg['max_dis']}", f"--dis_scale={cfg['dis_scale']}", "--cc_name=GRD",
                           "--use_cs=true", "--reg_lambda=0.3", "--use_pp=true", "--seed=12345"], stdout=subprocess.DEVNULL)
    _gpu_run(gpu_ctx, "GRD", cfg, l, r, 3, 12345)
    np.testing.assert_array_equal(_read_pfm(str(tmp_path / "l.pfm")), gpu_ctx.disparity_f64(0).astype(np.float32))
    lo, ro = gpu_ctx.postprocess(cfg["dis_scale"])
    np.testing.assert_array_equal(_pil_gray(str(tmp_path / "ld.png")), lo)  # PIL reads what image_io.cc wrote
    np.testing.assert_array_equal(_pil_gray(str(tmp_path / "rd.png")), ro)


@pytest.mark.gpu
def test_real_full_pair_cli_and_c_abi_against_the_oracle(gpu_ctx, tmp_path):
    """The whole 741x500 Middlebury pair with the reference's flags -- --max_dis=64 --dis_scale=4 --cc_name=GRD --use_cs=true
    --reg_lambda=0.3 --use_pp=true -- (a) through the command line, PNG in / PNG out; (b) through the C ABI; (c) the oracle in the
    DEVICE order: all 2 x 370 500 planes, stored costs, raw and post-processed maps identical; (d) the oracle in the REFERENCE
    order (serial window sum, no FMA, the reference's traversal): >= 99.5 % of the pixels of both views within 0.5 px
    (main.cc:68-69,131-134; cs_patchmatch.cc:51-109,508-588).  Odd width: levels 741, 371, 186, 93, 47 columns.
    ~1.3e11 window taps on the host, twice: about two minutes on 16 threads."""
    full = rd.load_full()
    if full is None:
        pytest.skip(f"REAL-PAIR TEST NOT RUN: {rd.SKIMAGE_DATA}/motorcycle_*.png is not on this machine "
                    "(the committed half-size crop was still tested)")
    cfg, l, r, gt = full
    lf, rf, _ = rd.full_files()
    assert l.shape == (500, 741, 3)
    flags = [f"--max_dis={cfg['max_dis']}", f"--dis_scale={cfg['dis_scale']}", "--cc_name=GRD", "--use_cs=true", "--reg_lambda=0.3",
             "--use_pp=true", "--seed=12345"]
    out = subprocess.check_output([CLI, f"--l_img_file={lf}", f"--r_img_file={rf}", f"--l_dis_file={tmp_path}/ld.png",
                                   f"--r_dis_file={tmp_path}/rd.png", f"--l_disp_pfm={tmp_path}/l.pfm", f"--r_disp_pfm={tmp_path}/r.pfm"] + flags).decode()
    assert "Total Time:" in out
    _gpu_run(gpu_ctx, "GRD", cfg, l, r, 3, 12345)
    assert [gpu_ctx.level_dims(s) for s in range(5)] == [(741, 500, 64), (371, 250, 32), (186, 125, 16), (93, 63, 8), (47, 32, 4)]
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
    gpu_disp = [gpu_ctx.disparity_f64(v) for v in (0, 1)]
    lo, ro = _assert_equals_oracle(gpu_ctx, pm, cfg, "motorcycle 741x500")
    # the command line saw the same pixels and wrote the same maps
    np.testing.assert_array_equal(_pil_gray(str(tmp_path / "ld.png")), lo)
    np.testing.assert_array_equal(_pil_gray(str(tmp_path / "rd.png")), ro)
    for v, side in ((0, "l"), (1, "r")):
        np.testing.assert_array_equal(_read_pfm(str(tmp_path / f"{side}.pfm")), gpu_disp[v].astype(np.float32))
    assert float(np.mean(lo != gpu_ctx.disparity_u8(0, cfg["dis_scale"]))) > 0.01  # post-processing did something on a real occlusion pattern
    # reference order: the north-star bar
    pm2 = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm2.run(3, pc, True, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL, wavefront=True)
    within = []
    for v in (0, 1):
        d = np.abs(gpu_disp[v] - pm2.disp_f64(v))
        within.append(float(np.mean(d <= 0.5)))
        assert within[-1] >= 0.995, (v, within[-1], float(d.max()))
    pp_equal = float(np.mean(lo == pm2.dis(0)))
    # accuracy against the real ground truth (left view, known pixels): GPU, and the reference-order CPU result next to it
    rec = {
        "gpu_raw_bad2": rd.bad_fraction(gpu_disp[0], gt, 2.0), "gpu_post_processed_bad2": rd.bad_fraction(lo.astype(np.float64) / cfg["dis_scale"], gt, 2.0),
        "cpu_reference_order_raw_bad2": rd.bad_fraction(pm2.disp_f64(0), gt, 2.0),
        "cpu_reference_order_post_processed_bad2": rd.bad_fraction(pm2.dis(0).astype(np.float64) / cfg["dis_scale"], gt, 2.0),
        "within_0.5px_of_reference_order": within, "post_processed_maps_equal_reference_order": pp_equal,
        "mean_disparity_gpu": float(np.nanmean(np.where(np.isfinite(gt), gpu_disp[0], np.nan))), "mean_disparity_gt": float(np.nanmean(gt)),
    }
    _record("full_741x500_D64_GRD_cs", rec)
    assert rec["gpu_post_processed_bad2"] < 0.25 and abs(rec["gpu_raw_bad2"] - rec["cpu_reference_order_raw_bad2"]) < 0.005, rec


@pytest.mark.gpu
def test_real_half_size_pair_accuracy_next_to_the_reference_probe(gpu_ctx):
    """370x250, D = 32: the size SURVEY.md 8(c) quotes the UNMODIFIED reference on (bad-2.0 against the ground truth 10.9-11.1 %,
    its runs differing from each other because it seeds from the clock).  The GPU's figure must sit in that neighbourhood, and its
    planes equal the oracle's at this size too (levels 370, 185, 93, 47, 24)."""
    half = rd.load_half()
    if half is None:
        pytest.skip(f"REAL-PAIR TEST NOT RUN: {rd.SKIMAGE_DATA}/motorcycle_*.png is not on this machine")
    cfg, l, r, gt = half
    _gpu_run(gpu_ctx, "GRD", cfg, l, r, 3, 12345)
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)
    lo, _ = _assert_equals_oracle(gpu_ctx, pm, cfg, "motorcycle 370x250")
    bad = rd.bad_fraction(gpu_ctx.disparity_f64(0), gt, 2.0)
    _record("half_370x250_D32_GRD_cs", {"gpu_raw_bad2": bad, "gpu_post_processed_bad2": rd.bad_fraction(lo.astype(np.float64) / cfg["dis_scale"], gt, 2.0),
                                        "reference_probe_raw_bad2": "0.109-0.111 (SURVEY.md 8(c), unmodified reference, clock-seeded)"})
    assert 0.07 < bad < 0.15, bad


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["redblack", "row_shared_rng", "volumes", "computed_tables", "per_diagonal", "folded_sweep", "lambda0", "lambda1", "source_order_view"])
def test_real_crop_every_mode_bit_exact(gpu_ctx, mode):
    """the photograph through the modes the synthetic pairs are tested in: the red-black schedule, the reference's USE_OMP random
    streams (one per row), materialised f64 volumes, computed instead of DMA-filled tables, per-diagonal sweep launches, the folded
    sweep, the reference CLI's default lambda = 0 and lambda = 1, view propagation in source order -- GRD cross-scale, 2 iterations
    + post-processing, every plane and map identical to the oracle run the same way"""
    from crossscalepatchmatch_amd import capi
    cfg, l, r, _ = rd.load_crop()
    lam = {"lambda0": 0.0, "lambda1": 1.0}.get(mode, 0.3)
    sched = po.SCHED_REDBLACK if mode == "redblack" else po.SCHED_RASTER
    rng_mode = po.RNG_ROW_SHARED if mode == "row_shared_rng" else po.RNG_PER_PIXEL
    try:
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, int(mode == "per_diagonal"))
        gpu_ctx.set_option(capi.OPT_SWEEP_FOLD, int(mode == "folded_sweep"))
        gpu_ctx.set_option(capi.OPT_VIEW_SORT, int(mode != "source_order_view"))
        gpu_ctx.set_images(l, r)
        gpu_ctx.build_cost_grd(cfg["max_dis"], 35, 5, lam, volumes=(mode == "volumes"), table_volumes=(mode != "computed_tables"))
        gpu_ctx.patchmatch(2, seed=77, schedule=sched, rng_mode=rng_mode, rb_rounds=2 if mode == "redblack" else 1)
        pc = po.PlaneCost(l, r, cfg["max_dis"], 35, 5, lam)
        pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
        pm.run(2, pc, False, seed=77, schedule=sched, sum_order=po.SUM_DEVICE, rng_mode=rng_mode, rb_rounds=2 if mode == "redblack" else 1,
               wavefront=(sched == po.SCHED_RASTER))
        _assert_equals_oracle(gpu_ctx, pm, cfg, mode)
    finally:
        gpu_ctx.set_option(capi.OPT_RASTER_LAUNCHES, 0)
        gpu_ctx.set_option(capi.OPT_SWEEP_FOLD, 0)
        gpu_ctx.set_option(capi.OPT_VIEW_SORT, 1)
