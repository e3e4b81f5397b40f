"""The C++ host layer (crossscalepatchmatch_amd/host): image I/O + gflags-compatible parser on CPU; the
reference-style command line end to end on the GPU (-m gpu)."""
import os
import subprocess

import numpy as np
import pytest

import pngio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "crossscalepatchmatch_amd", "host")


@pytest.fixture(scope="module")
def io_check(tmp_path_factory):
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_io_check")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "helpers", "host_io_check.cc"),
                           os.path.join(HOST, "image_io.cc"), "-lz"])
    return exe


@pytest.mark.parametrize("filt", [0, 1, 2])
def test_png_and_pnm_roundtrip_and_flags(io_check, tmp_path, filt):
    rng = np.random.default_rng(filt)
    rgb = rng.integers(0, 256, (13, 17, 3)).astype(np.uint8)
    src = str(tmp_path / "in.png")
    pngio.write_png(src, rgb, filter_type=filt)
    for ext in ("png", "ppm"):
        oc, og = str(tmp_path / f"c.{ext}"), str(tmp_path / ("g.png" if ext == "png" else "g.pgm"))
        out = subprocess.check_output([io_check, src, oc, og, "--l_img_file=a b.png", "--max_dis", "60", "--use_cs", "--nouse_pp",
                                       "-reg_lambda=0.3"]).decode().strip()
        assert out == "a b.png|60|1|0|0.29999999999999999"
        rd = pngio.read_png if ext == "png" else pngio.read_pnm
        np.testing.assert_array_equal(rd(oc), rgb)
        np.testing.assert_array_equal(rd(og), rgb[..., 1])
    # a gray PNG and a PGM load as 3 equal channels (CV_LOAD_IMAGE_COLOR)
    pngio.write_png(str(tmp_path / "gray.png"), rgb[..., 0])
    pngio.write_pnm(str(tmp_path / "gray.pgm"), rgb[..., 0])
    for name in ("gray.png", "gray.pgm"):
        subprocess.check_call([io_check, str(tmp_path / name), str(tmp_path / "o.ppm"), str(tmp_path / "o.pgm")], stdout=subprocess.DEVNULL)
        np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "o.pgm")), rgb[..., 0])
    assert subprocess.call([io_check, str(tmp_path / "missing.png"), "a", "b"]) == 3


def test_unknown_flag_is_rejected(io_check, tmp_path):
    src = str(tmp_path / "in.pgm")
    pngio.write_pnm(src, np.zeros((2, 2), np.uint8))
    p = subprocess.run([io_check, src, str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), "--no_such_flag=1"], capture_output=True)
    assert p.returncode == 1 and b"unknown command line flag" in p.stderr


REF_MAIN = "/root/reference/CSPM/main.cc"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference checkout is not on this machine (GPU box)")
def test_reference_main_cc_compiles_and_links_unchanged(tmp_path):
    """the reference's own main.cc (CSPM/main.cc, Windows-style includes and all) builds against host/
    without an edit -- the six backslash-named forwarding headers make `#include"plane_cost\\pre_cs_pc.h"` resolve.  The file
    is copied to a temporary directory for the compile and never enters the repository."""
    import shutil
    pkg = os.path.join(ROOT, "crossscalepatchmatch_amd")
    assert os.path.exists(os.path.join(pkg, "libcspm_hip.so")), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    src = str(tmp_path / "main.cc")
    shutil.copyfile(REF_MAIN, src)
    exe = str(tmp_path / "cspm_ref_main")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", HOST, "-o", exe, src, os.path.join(HOST, "host_impl.cc"),
                           os.path.join(HOST, "image_io.cc"), "-L", pkg, "-lcspm_hip", "-lz", "-Wl,-rpath," + pkg])
    # no GPU here: the binary starts, parses the reference's flags and fails cleanly on the missing image
    p = subprocess.run([exe, "--l_img_file=/nonexistent.png", "--max_dis=16", '--cc_name="GRD"', "--use_cs=true"], input=b"\n",
                       capture_output=True, timeout=60)
    assert p.returncode == 1 and b"can not open image" in p.stdout


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"Pf"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        assert scale < 0  # little-endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w)[::-1]


@pytest.mark.gpu
def test_cli_float_maps_and_batch_list(gpu_ctx, mid_pair, small_pair, tmp_path):
    """--l_disp_pfm / --r_disp_pfm (float32 PFM of the unquantised plane disparities) and --batch_list (one pair per line, one
    device context reused) == the C-ABI path, pair by pair."""
    exe = os.path.join(ROOT, "crossscalepatchmatch_amd", "cspm_main")
    lines = []
    sets = [("a", mid_pair), ("b", mid_pair), ("c", small_pair)]  # two equal sizes (buffers reused), then a different one
    for tag, pr in sets:
        pngio.write_png(str(tmp_path / f"l{tag}.png"), pr["l"][..., ::-1])
        pngio.write_png(str(tmp_path / f"r{tag}.png"), pr["r"][..., ::-1])
        lines.append(" ".join(str(tmp_path / n) for n in (f"l{tag}.png", f"r{tag}.png", f"ld{tag}.png", f"rd{tag}.png", f"l{tag}.pfm", f"r{tag}.pfm")))
    (tmp_path / "list.txt").write_text("# l r ld rd [lpfm rpfm]\n" + "\n".join(lines) + "\n\n")
    out = subprocess.check_output([exe, f"--batch_list={tmp_path}/list.txt", "--max_dis=16", "--dis_scale=4", "--cc_name=GRD", "--use_cs=true",
                                   "--reg_lambda=0.3", "--seed=31"]).decode()
    assert out.count("Total Time:") == 3 and "Batch: 3 pairs" in out
    for tag, pr in sets:
        gpu_ctx.set_images(pr["l"], pr["r"])
        gpu_ctx.build_cost_grd(16, 35, 5, 0.3)
        gpu_ctx.patchmatch(3, seed=31, schedule=0)
        for v, side in ((0, "l"), (1, "r")):
            np.testing.assert_array_equal(pngio.read_png(str(tmp_path / f"{side}d{tag}.png")), gpu_ctx.disparity_u8(v, 4))
            np.testing.assert_array_equal(_read_pfm(str(tmp_path / f"{side}{tag}.pfm")), gpu_ctx.disparity_f64(v).astype(np.float32))
    # single-pair mode with the float flags
    subprocess.check_call([exe, f"--l_img_file={tmp_path}/lc.png", f"--r_img_file={tmp_path}/rc.png", f"--l_dis_file={tmp_path}/x.pgm",
                           f"--r_dis_file={tmp_path}/y.pgm", f"--l_disp_pfm={tmp_path}/x.pfm", "--max_dis=16", "--dis_scale=4", "--cc_name=GRD",
                           "--use_cs=true", "--reg_lambda=0.3", "--seed=31"], stdout=subprocess.DEVNULL)
    np.testing.assert_array_equal(_read_pfm(str(tmp_path / "x.pfm")), gpu_ctx.disparity_f64(0).astype(np.float32))
    assert not os.path.exists(tmp_path / "y.pfm")


@pytest.mark.gpu
@pytest.mark.parametrize("use_cs,use_pp", [(False, False), (True, True)])
def test_cli_matches_the_c_abi_path(gpu_ctx, mid_pair, tmp_path, use_cs, use_pp):
    """cspm_main with the reference's flags (main.cc:23-34) == the same run through the C ABI."""
    exe = os.path.join(ROOT, "crossscalepatchmatch_amd", "cspm_main")
    assert os.path.exists(exe), "build the host layer: python -c 'import __graft_entry__ as g; g.build()'"
    l, r = mid_pair["l"], mid_pair["r"]
    pngio.write_png(str(tmp_path / "l.png"), l[..., ::-1])  # files are RGB, imread returns BGR
    pngio.write_png(str(tmp_path / "r.png"), r[..., ::-1])
    args = [exe, f"--l_img_file={tmp_path}/l.png", f"--r_img_file={tmp_path}/r.png", f"--l_dis_file={tmp_path}/ld.png",
            f"--r_dis_file={tmp_path}/rd.png", "--max_dis=16", "--dis_scale=4", '--cc_name="GRD"', f"--use_cs={'true' if use_cs else 'false'}",
            f"--use_pp={'true' if use_pp else 'false'}", "--reg_lambda=0.3", "--seed=777"]
    out = subprocess.check_output(args).decode()
    assert "Total Time:" in out
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(16, 35, 5 if use_cs else 0, 0.3)
    gpu_ctx.patchmatch(3, seed=777, schedule=0)
    want = gpu_ctx.postprocess(4) if use_pp else (gpu_ctx.disparity_u8(0, 4), gpu_ctx.disparity_u8(1, 4))
    np.testing.assert_array_equal(pngio.read_png(str(tmp_path / "ld.png")), want[0])
    np.testing.assert_array_equal(pngio.read_png(str(tmp_path / "rd.png")), want[1])
    # unknown cost name: error exit instead of the reference's NULL dereference
    assert subprocess.call(args[:7] + ['--cc_name="BSM"'], stdout=subprocess.DEVNULL) == 1
    # the second CCMethod behind the same slot (every line of the reference's input.txt uses it)
    subprocess.check_call(args[:7] + ['--cc_name=CEN', "--use_cs=false", "--seed=5", f"--l_dis_file={tmp_path}/lc.pgm", f"--r_dis_file={tmp_path}/rc.pgm"],
                          stdout=subprocess.DEVNULL)
    gpu_ctx.build_cost_cen(16, 35, 0, 0.0)
    gpu_ctx.patchmatch(3, seed=5, schedule=0)
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "lc.pgm")), gpu_ctx.disparity_u8(0, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("use_cs", [0, 1])
def test_foreign_ccmethod_plugin_through_the_cpp_layer(gpu_ctx, small_pair, tmp_path, use_cs):
    """A CCMethod written by a plugin author (tests/helpers/foreign_cc_check.cc) drives PreSSPC/PreCSPC + CSPatchMatch:
    the host layer calls it level by level on CV_64FC3 Mats and uploads its volumes (pre_cs_pc.cc:57-74)."""
    import ctypes as C
    from oracle import pyoracle as po
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "foreign_cc_check")
    pkg = os.path.join(ROOT, "crossscalepatchmatch_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "helpers", "foreign_cc_check.cc"),
                           os.path.join(HOST, "host_impl.cc"), os.path.join(HOST, "image_io.cc"), "-L", pkg, "-lcspm_hip", "-lz",
                           "-Wl,-rpath," + pkg])
    l, r, D = small_pair["l"], small_pair["r"], small_pair["max_dis"]
    pngio.write_pnm(str(tmp_path / "l.ppm"), l[..., ::-1])
    pngio.write_pnm(str(tmp_path / "r.ppm"), r[..., ::-1])
    txt = subprocess.check_output([exe, str(tmp_path / "l.ppm"), str(tmp_path / "r.ppm"), str(D), str(use_cs), str(tmp_path / "ol.pgm"),
                                   str(tmp_path / "or.pgm")]).decode().split()
    # the same cost function in numpy, fed to the oracle (volumes overwritten) and to the C ABI (slab upload)
    sn = 3 if use_cs else 0
    pc = po.PlaneCost(l, r, D, 35, sn, 0.3 if use_cs else 0.0)
    gpu_ctx.set_images(l, r)
    gpu_ctx.begin_cost(D, 35, sn, 0.3 if use_cs else 0.0)
    for s in range(pc.levels):
        w, h, Ds = pc.dims(s)
        rgb = [pc.image(v, s)[..., ::-1].astype(np.float64) for v in (0, 1)]
        for v in (0, 1):
            vol = pc.volume(v, s)
            for d in range(Ds + 1):
                slab = np.full((h, w), 100.0)
                if d < w:
                    if v == 0:
                        slab[:, d:] = np.abs(rgb[0][:, d:, 0] - rgb[1][:, :w - d, 0]) + 0.5 * np.abs(rgb[0][:, d:, 2] - rgb[1][:, :w - d, 2])
                    else:
                        slab[:, :w - d] = np.abs(rgb[1][:, :w - d, 0] - rgb[0][:, d:, 0]) + 0.5 * np.abs(rgb[1][:, :w - d, 2] - rgb[0][:, d:, 2])
                vol[d] = slab
                gpu_ctx.upload_cost_slab(v, s, d, slab)
    pc.refresh_max_cost()
    gpu_ctx.finish_cost()
    # per-call GetPlaneCost through the C++ virtual == oracle (device order)
    pts = [(0, 0, 0), (small_pair["w"] // 2, small_pair["h"] // 2, 1), (small_pair["w"] - 1, small_pair["h"] - 1, 0)]
    for (x, y, v), got in zip(pts, txt):
        n = np.array([0.1, -0.2, 0.97])
        prm = po.plane_param(n, [x, y, 4.25])
        assert float(got) == pc.cost(x, y, n, prm, v, po.SUM_DEVICE)
    # whole PatchMatch with the plugin's volumes == oracle with the same volumes
    pm = po.PatchMatch(l, r, D, 4)
    pm.run(2, pc, False, seed=99, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "ol.pgm")), pm.dis(0))
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "or.pgm")), pm.dis(1))


@pytest.mark.gpu
@pytest.mark.parametrize("use_cs", [0, 1])
def test_grdpc_cspc_classes_through_the_cpp_layer(gpu_ctx, small_pair, tmp_path, use_cs):
    """`new GrdPC(...)` / `new CSPC(...)` (plane_cost/grd_pc.h:27-29, cspc.h:21-23) with the reference's constructor signatures:
    per-call GetPlaneCost and the whole PatchMatch + post-processing == oracle; the CLI reaches them with --pc_name=IMG."""
    from oracle import pyoracle as po
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "img_pc_check")
    pkg = os.path.join(ROOT, "crossscalepatchmatch_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "helpers", "img_pc_check.cc"),
                           os.path.join(HOST, "host_impl.cc"), os.path.join(HOST, "image_io.cc"), "-L", pkg, "-lcspm_hip", "-lz",
                           "-Wl,-rpath," + pkg])
    l, r, D = small_pair["l"], small_pair["r"], small_pair["max_dis"]
    pngio.write_pnm(str(tmp_path / "l.ppm"), l[..., ::-1])
    pngio.write_pnm(str(tmp_path / "r.ppm"), r[..., ::-1])
    txt = subprocess.check_output([exe, str(tmp_path / "l.ppm"), str(tmp_path / "r.ppm"), str(D), str(use_cs), str(tmp_path / "ol.pgm"),
                                   str(tmp_path / "or.pgm")]).decode().split()
    pc = po.PlaneCost(l, r, D, 35, 3 if use_cs else 0, 0.3 if use_cs else 0.0, cc="IMG")
    pts = [(0, 0, 0), (small_pair["w"] // 2, small_pair["h"] // 2, 1), (small_pair["w"] - 1, small_pair["h"] - 1, 0)]
    for (x, y, v), got in zip(pts, txt):
        n = np.array([0.1, -0.2, 0.97])
        assert float(got) == pc.cost(x, y, n, po.plane_param(n, [x, y, 4.25]), v, po.SUM_DEVICE)
    pm = po.PatchMatch(l, r, D, 4)
    pm.run(2, pc, True, seed=99, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE)
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "ol.pgm")), pm.dis(0))
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "or.pgm")), pm.dis(1))
    # the command line: --pc_name=IMG
    cli = os.path.join(pkg, "cspm_main")
    subprocess.check_call([cli, f"--l_img_file={tmp_path}/l.ppm", f"--r_img_file={tmp_path}/r.ppm", f"--l_dis_file={tmp_path}/cl.pgm",
                           f"--r_dis_file={tmp_path}/cr.pgm", f"--max_dis={D}", "--dis_scale=4", "--pc_name=IMG", "--cc_name=GRD",
                           f"--use_cs={'true' if use_cs else 'false'}", "--reg_lambda=0.3", "--seed=99", "--use_pp=true", "--iters=2"],
                          stdout=subprocess.DEVNULL)
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_img(D, 35, 5 if use_cs else 0, 0.3)
    gpu_ctx.patchmatch(2, seed=99, schedule=0)
    want = gpu_ctx.postprocess(4)
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "cl.pgm")), want[0])
    np.testing.assert_array_equal(pngio.read_pnm(str(tmp_path / "cr.pgm")), want[1])
