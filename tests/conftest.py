import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _pair(w, h, max_dis, regions, seed):
    from crossscalepatchmatch_amd import synth
    return synth.make_pair(w, h, max_dis, regions=regions, seed=seed)


@pytest.fixture(scope="session")
def small_pair():
    """64x48, max_dis 16: the T1/T2 fixture size of SURVEY.md section 4."""
    l, r, gl, gr = _pair(64, 48, 16, 3, 11)
    return dict(l=l, r=r, gl=gl, gr=gr, w=64, h=48, max_dis=16)


@pytest.fixture(scope="session")
def mid_pair():
    """96x64, max_dis 16: the T3 whole-pipeline fixture size."""
    l, r, gl, gr = _pair(96, 64, 16, 3, 12)
    return dict(l=l, r=r, gl=gl, gr=gr, w=96, h=64, max_dis=16)


@pytest.fixture(scope="session")
def odd_pair():
    """odd, non-multiple-of-anything dims; exercises pyramid rounding and ragged tails."""
    l, r, gl, gr = _pair(77, 41, 21, 3, 13)
    return dict(l=l, r=r, gl=gl, gr=gr, w=77, h=41, max_dis=21)


@pytest.fixture
def gpu_ctx(_gpu_ctx_session):
    """the session's context with its options back at the library defaults: StereoContext.build_cost_grd leaves an option alone
    unless the test passes it, so a test that switched the table volumes off must not decide what the next one measures"""
    from crossscalepatchmatch_amd import capi
    ctx = _gpu_ctx_session
    for key, value in ((capi.OPT_GRD_VOLUMES, 0), (capi.OPT_SWEEP_PAIRS, 0), (capi.OPT_TABLE_VOLUMES, 1), (capi.OPT_RASTER_LAUNCHES, 0),
                       (capi.OPT_SWEEP_TIMEOUT_MS, 3000), (capi.OPT_SWEEP_PACKED, 0), (capi.OPT_SWEEP_FLOW, 0), (capi.OPT_SWEEP_WG, 0),
                       (capi.OPT_VOLUME_RETRY_PAIRS, 16), (capi.OPT_VIEW_SORT, 1), (capi.OPT_SWEEP_FOLD, 0)):
        ctx.set_option(key, value)
    return ctx


@pytest.fixture(scope="session")
def _gpu_ctx_session():
    # PyTorch bundles its own HIP runtime: when both live in one process, torch has to initialise first (bench.py and
    # batch.py do the same); libcspm_hip.so then binds to the runtime that is already loaded.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import crossscalepatchmatch_amd as cs
    ctx = cs.StereoContext(0)  # raises CspmError when the HIP library / device is missing: no fallback
    yield ctx
    ctx.close()


def random_planes(rng, n, w, h, max_dis):
    """(xy, norm, point, param) tuples incl. corners, near-zero nz, out-of-range disparities."""
    from oracle import pyoracle as po
    xy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.int32)
    corners = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1], [w // 2, 0], [0, h // 2]], np.int32)
    xy[:len(corners)] = corners
    norm = rng.normal(size=(n, 3))
    norm /= np.linalg.norm(norm, axis=1, keepdims=True)
    z = rng.uniform(-0.3 * max_dis, 1.4 * max_dis, n)
    # special cases
    norm[6] = [0.6, 0.8, 1e-9]       # |nz| < kDoubleEps -> clamped denominator, huge slopes
    norm[7] = [0.6, -0.8, -1e-12]    # negative tiny nz
    norm[8] = [0.0, 0.0, 1.0]; z[8] = 5.0    # fronto-parallel, integer disparity (floor_wgt == 1)
    norm[9] = [0.0, 0.0, -1.0]; z[9] = 7.25
    norm[10] = [0.0, 0.0, 1.0]; z[10] = -3.0  # everything invalid
    norm[11] = [0.0, 0.0, 1.0]; z[11] = 10.0 * max_dis
    norm[12] = [0.0, 0.0, 1.0]; z[12] = 0.5   # floor == 0 -> invalid
    norm[13] = [0.0, 0.0, 1.0]; z[13] = max_dis - 0.5  # floor == max_dis-1 -> valid, reads slab max_dis
    point = np.concatenate([xy.astype(np.float64), z[:, None]], 1)
    param = np.stack([po.plane_param(norm[i], point[i]) for i in range(n)])
    return xy, norm, point, param
