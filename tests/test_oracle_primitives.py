"""T0: hand-derived known answers for the primitives the hot path is built on.
These pin the oracle's restatement of commfunc.h / plane.h / pre_cs_pc.cc arithmetic independently of any
reference run (the reference cannot be built here: PARITY UNPINNED, see oracle/cspm_oracle.h)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po


def test_round2int_is_round_half_even():
    # commfunc.h:117-121: d + 6755399441055744.0, low 32 bits
    L = po.lib()
    cases = {0.5: 0, 1.5: 2, 2.5: 2, -0.5: 0, -1.5: -2, 127.5: 128, 3.49999: 3, -3.5: -4, 254.5: 254, 255.5: 256,
             0.0: 0, 1e-12: 0, 59.999999: 60}
    for d, want in cases.items():
        assert L.csor_round2int(d) == want, d
    rng = np.random.default_rng(0)
    xs = rng.uniform(-1000, 1000, 20000)
    got = np.array([L.csor_round2int(float(x)) for x in xs])
    np.testing.assert_array_equal(got, np.rint(xs).astype(int))  # numpy rint is round-half-even too


def test_handle_border_single_wrap():
    L = po.lib()  # commfunc.h:129-145
    assert [L.csor_handle_border(v, 10) for v in (-1, -10, 0, 9, 10, 19, 5)] == [9, 0, 0, 9, 0, 9, 5]
    assert L.csor_handle_border(25, 10) == 15  # only ONE wrap: still outside (the reference would index out of range)


def test_plane_update_param():
    # plane.h:25-34: a=-nx/den, b=-ny/den, c=(n.p)/den, den = sign(nz)*max(|nz|,1e-8), nz==0 -> +1e-8
    p = po.plane_param([0.0, 0.0, 1.0], [3.0, 4.0, 7.5])
    np.testing.assert_array_equal(p, [-0.0, -0.0, 7.5])
    p = po.plane_param([0.6, 0.0, 0.8], [10.0, 20.0, 5.0])
    np.testing.assert_allclose(p, [-0.75, 0.0, (6.0 + 4.0) / 0.8], rtol=1e-15)
    p = po.plane_param([0.6, 0.0, -0.8], [10.0, 20.0, 5.0])
    np.testing.assert_allclose(p, [0.75, 0.0, (6.0 - 4.0) / -0.8], rtol=1e-15)
    p = po.plane_param([1.0, 0.0, 0.0], [2.0, 0.0, 9.0])       # nz == 0 -> denominator +1e-8
    np.testing.assert_allclose(p, [-1e8, 0.0, 2e8], rtol=1e-12)
    p = po.plane_param([1.0, 0.0, -1e-12], [2.0, 0.0, 9.0])    # tiny negative nz -> -1e-8
    np.testing.assert_allclose(p, [1e8, 0.0, (2.0 - 9e-12) / -1e-8], rtol=1e-12)
    # the plane passes through its anchor: a*px + b*py + c == pz
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        if abs(n[2]) < 1e-3:
            continue
        pt = rng.uniform(0, 100, 3)
        a, b, c = po.plane_param(n, pt)
        assert abs(a * pt[0] + b * pt[1] + c - pt[2]) < 1e-8 * (1 + abs(c))


def test_exp_lut():
    lut = np.zeros(1000)
    po.lib().csor_exp_lut(lut.ctypes.data_as(C.POINTER(C.c_double)), 10.0)
    assert lut[0] == 1.0
    np.testing.assert_allclose(lut, np.exp(-np.arange(1000) / 10.0), rtol=4.5e-16)  # libm vs numpy: <= 1 ulp


def test_scale_weights_known_values():
    # pre_cs_pc.cc:86-109; values of SURVEY.md T0 and an independent numpy inverse
    def w(S, lam):
        o = np.zeros(S)
        assert po.lib().csor_scale_weights(S, lam, o.ctypes.data_as(C.POINTER(C.c_double))) == 0
        return o
    np.testing.assert_array_equal(w(5, 0.0), [1, 0, 0, 0, 0])
    np.testing.assert_allclose(w(5, 0.3), [0.80539988076, 0.156732816626, 0.030508474576, 0.005979047781, 0.001379780257], rtol=0, atol=1e-12)
    np.testing.assert_allclose(w(5, 1.0), [0.618181818182, 0.236363636364, 0.090909090909, 0.036363636364, 0.018181818182], rtol=0, atol=1e-12)
    for S in (2, 3, 5, 8):
        for lam in (0.05, 0.3, 1.0, 4.0):
            M = np.zeros((S, S))
            for s in range(S):
                M[s, s] = 1 + lam if s in (0, S - 1) else 1 + 2 * lam
                if s > 0: M[s, s - 1] = -lam
                if s < S - 1: M[s, s + 1] = -lam
            np.testing.assert_allclose(w(S, lam), np.linalg.inv(M)[0], rtol=1e-13)
            assert np.all(w(S, lam) >= 0)  # inverse of an M-matrix: what makes early exit result-preserving


def test_refine_step_counts():
    L = po.lib()  # cs_patchmatch.cc:95,299-301,342: z = max_dis/2; while z >= 0.1: z /= 2
    assert [L.csor_refine_steps(d) for d in (60, 128, 256, 16)] == [9, 10, 11, 7]


def test_pyramid_dims_of_the_baseline_configs():
    # (W,H,D) per level for C2, C3, C5 (pre_cs_pc.cc:43-49)
    want = {(450, 375, 60): [(450, 375, 60), (225, 188, 30), (113, 94, 15), (57, 47, 7), (29, 24, 3)],
            (1242, 375, 128): [(1242, 375, 128), (621, 188, 64), (311, 94, 32), (156, 47, 16), (78, 24, 8)],
            (3000, 2000, 256): [(3000, 2000, 256), (1500, 1000, 128), (750, 500, 64), (375, 250, 32), (188, 125, 16)]}
    for (w, h, d), dims in want.items():
        cur = (w, h, d)
        got = [cur]
        for _ in range(4):
            cur = ((cur[0] + 1) // 2, (cur[1] + 1) // 2, cur[2] // 2)
            got.append(cur)
        assert got == dims
    img = np.zeros((47, 57, 3), np.uint8)
    pc = po.PlaneCost(img, img, 7, 35, 5, 0.3)
    assert [pc.dims(s) for s in range(5)] == [(57, 47, 7), (29, 24, 3), (15, 12, 1), (8, 6, 0), (4, 3, 0)]


def test_rng_is_a_pure_function_of_its_key():
    L = po.lib()
    a = L.csor_rng_u64(12345, 7, 1000, 3)
    assert a == L.csor_rng_u64(12345, 7, 1000, 3)
    keys = {L.csor_rng_u64(12345, s, p, d) for s in range(4) for p in range(50) for d in range(4)}
    assert len(keys) == 4 * 50 * 4
    u = np.array([L.csor_rng_u01(99, 1, p, 0) for p in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # stream ids are unique per (phase, iter, step, view)
    ids = {L.csor_stream_id(ph, it, st, v) for ph in (0, 1) for it in range(16) for st in range(32) for v in (0, 1)}
    assert len(ids) == 2 * 16 * 32 * 2


def test_init_normals_are_unit_and_isotropic(small_pair):
    pc = po.PlaneCost(small_pair["l"], small_pair["r"], small_pair["max_dis"], 3, 0, 0.0)
    pm = po.PatchMatch(small_pair["l"], small_pair["r"], small_pair["max_dis"], 4)
    pm.init(pc, seed=3)
    P = pm.planes(0)
    n = P[..., 0:3].reshape(-1, 3)
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    assert np.all(np.abs(n.mean(0)) < 0.06)            # uniform on the sphere: zero mean
    assert np.all(np.abs((n ** 2).mean(0) - 1 / 3) < 0.03)
    z = P[..., 5]
    assert z.min() >= 1e-8 and z.max() < small_pair["max_dis"]
    np.testing.assert_array_equal(P[..., 3], np.tile(np.arange(small_pair["w"]), (small_pair["h"], 1)))


@pytest.mark.parametrize("scale_num,lam,sum_order", [(0, 0.0, po.SUM_SERIAL), (3, 0.3, po.SUM_DEVICE)])
def test_wavefront_sweep_is_the_serial_sweep(odd_pair, scale_num, lam, sum_order):
    """csor_pm_opts.wavefront walks the raster sweep anti-diagonal by anti-diagonal with OpenMP over a diagonal's pixels (the speed
    knob of the whole-KITTI-pair GPU test): planes, costs and evaluation counts equal the reference's serial double loop
    (cs_patchmatch.cc:163-216) in both sweep directions, over several iterations."""
    l, r, D = odd_pair["l"], odd_pair["r"], odd_pair["max_dis"]
    pc = po.PlaneCost(l, r, D, 9, scale_num, lam)
    out = []
    for wavefront in (False, True):
        pm = po.PatchMatch(l, r, D, 4)
        pm.run(3, pc, False, seed=77, schedule=po.SCHED_RASTER, sum_order=sum_order, threads=4, wavefront=wavefront)
        out.append((pm.planes(0).copy(), pm.planes(1).copy(), pm.min_cost(0).copy(), pm.min_cost(1).copy(), pm.evals()))
    for a, b in zip(out[0][:4], out[1][:4]):
        np.testing.assert_array_equal(a, b)
    assert out[0][4] == out[1][4]


def test_f32_gray_of_every_colour_is_a_multiple_of_2_pow_minus_27():
    """What the packed 8-byte pixels of the raster sweep rest on (csrc/cspm_device.h Pix8): the GRD x-gradient is
    gray[x+1] - gray[x-1] of the f32 gray image (grd_cc.cpp:70-77), so if EVERY f32 gray value of an 8-bit colour is a multiple of
    2^-27 in [0, 256), every gradient is a multiple of 2^-27 in (-256, 256): 36 bits, exactly.  All 2^24 colours, through the
    oracle's own RGB2GRAY restatement; then the encode / decode arithmetic of the device restated in numpy on random gradients."""
    L = po.lib()
    chunk = 1 << 20
    base = np.arange(chunk, dtype=np.int64)
    gmax = 0.0
    for hi in range(16):
        c = base + hi * chunk
        rgb = np.stack([c & 255, (c >> 8) & 255, (c >> 16) & 255], 1).astype(np.float64)
        gray = np.zeros(chunk, np.float32)
        L.csor_rgb2gray_f32(rgb.ctypes.data_as(C.POINTER(C.c_double)), chunk, 1, gray.ctypes.data_as(C.POINTER(C.c_float)))
        g = gray.astype(np.float64)
        s = g * 2.0 ** 27
        assert np.all(s == np.rint(s)) and g.min() >= 0.0 and g.max() < 256.0
        gmax = max(gmax, float(g.max()))
    assert gmax > 254.9
    # encode (cspm_device.h pix8_encode) / decode (cspm_tap.h pix8_x, grd8_cell) on random pairs of such gradients
    rng = np.random.default_rng(1)
    ga = rng.integers(-(2 ** 35) + 1, 2 ** 35, 100000).astype(np.float64) * 2.0 ** -27
    gb = rng.integers(-(2 ** 35) + 1, 2 ** 35, 100000).astype(np.float64) * 2.0 ** -27
    gb[:1000] = ga[:1000] + rng.integers(-3, 4, 1000) * 2.0 ** -27  # near-equal gradients: the untruncated branch
    ua, ub = (ga * 2.0 ** 27 + 2.0 ** 35).astype(np.uint64), (gb * 2.0 ** 27 + 2.0 ** 35).astype(np.uint64)
    assert ua.max() < 2 ** 36 and ub.max() < 2 ** 36
    xa = ((np.uint64(0x43300000) << np.uint64(32)) | ua).view(np.float64)
    xb = ((np.uint64(0x43300000) << np.uint64(32)) | ub).view(np.float64)
    np.testing.assert_array_equal(xa - 2.0 ** 52, ua.astype(np.float64))
    packed = np.minimum(np.abs(xa - xb), 2.0 ** 28)                # min(|dX|, TAU_GRD * 2^27)
    plain = np.minimum(np.abs(ga - gb), 2.0)                       # min(|dG|, TAU_GRD), grd_cc.cpp:14-17
    np.testing.assert_array_equal(packed * 2.0 ** -27, plain)      # the same number; the factor folds into the cell's fma constant
    assert ((1 - 0.1) * 2.0 ** -27) * 2.0 ** 27 == (1 - 0.1)       # (1-ALPHA) * 2^-27 is an exact scaling of (1-ALPHA)
