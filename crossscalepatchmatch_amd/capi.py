"""ctypes view of the C ABI in include/cspm.h (libcspm_hip.so).  Plumbing only -- all compute is in
the HIP library.  Fails loudly when the library is missing; there is no fallback path."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CSPM_LIB") or os.path.join(_HERE, "libcspm_hip.so")  # CSPM_LIB: an alternative build of the same library (tuning experiments)

SCHED_RASTER, SCHED_REDBLACK = 0, 1
RNG_PER_PIXEL, RNG_ROW_SHARED = 0, 1
K_GRD, K_INIT, K_SPATIAL, K_VIEW, K_REFINE, K_MISC, K_POST = range(7)
K_NAMES = ["grd", "init", "spatial", "view", "refine", "misc", "post"]
MAX_LEVELS = 8
OPT_GRD_VOLUMES = 1
OPT_RASTER_LAUNCHES = 2
OPT_SWEEP_TIMEOUT_MS = 3
OPT_SWEEP_FALLBACKS = 4
OPT_SWEEP_PAIRS = 5          # paired-cell volumes for the raster sweep: 0 never (default), 1 when they fit
OPT_SWEEP_PAIRS_ACTIVE = 6   # read only
OPT_TABLE_VOLUMES = 7        # device-cell volumes for the row kernels' DMA-filled tables: 1 when they fit (default), 0 never
OPT_TABLE_VOLUMES_ACTIVE = 8 # read only
OPT_SWEEP_PACKED = 10        # packed 8-byte pixels for the raster sweep (default 0: measured slower)
OPT_SWEEP_PACKED_ACTIVE = 11 # read only
OPT_SWEEP_PACKED_BAD = 12    # read only, synchronises: pixels the packer could not represent (0 by construction)
OPT_SWEEP_FLOW = 13          # persistent raster sweep scheduled by dataflow (1: measured slower) or by ordered claims (0, default)
OPT_SWEEP_WG = 14            # persistent sweep workgroups per CU (0 = default 2; 1 when three or more pairs are in flight on the GPU)
OPT_VOLUME_FALLBACKS = 9     # read only: hipMalloc failures of an optional volume this context survived
OPT_VOLUME_RETRY_PAIRS = 15  # a cost object that runs without the optional volumes it wanted asks again after this many reuses (default 16, 0 = never)
OPT_VIEW_SORT = 17           # view propagation evaluates a row's proposals in target-column order (default 1; identical planes either way)
OPT_SWEEP_FOLD = 18          # cross-scale sweep workgroups of levels - 1 waves, the coarsest level folded onto them (default 1; identical planes)
OPT_FAULT_VOLUME_ALLOC = 16  # write only, TEST HOOK: the n-th optional-volume allocation from now on fails

# every symbol include/cspm.h declares
SYMBOLS = [
    "cspm_device_count", "cspm_create", "cspm_destroy", "cspm_last_error", "cspm_set_stream", "cspm_get_stream", "cspm_synchronize",
    "cspm_set_images", "cspm_set_images_device", "cspm_build_cost_grd", "cspm_build_cost_cen", "cspm_build_cost_img", "cspm_cen_build_cv_host", "cspm_set_option", "cspm_get_option", "cspm_begin_cost", "cspm_upload_cost_slab",
    "cspm_finish_cost", "cspm_get_levels", "cspm_get_level_dims", "cspm_get_level_image", "cspm_get_cost_slab",
    "cspm_get_max_cost", "cspm_get_scale_weights", "cspm_grd_build_cv_host", "cspm_plane_cost_batch",
    "cspm_pm_default_params", "cspm_patchmatch", "cspm_pm_init", "cspm_pm_spatial", "cspm_pm_view", "cspm_pm_refine",
    "cspm_get_planes", "cspm_set_planes", "cspm_get_disparity_u8", "cspm_get_disparity_f64",
    "cspm_disparity_u8_device", "cspm_postprocess", "cspm_postprocess_device", "cspm_enable_timing", "cspm_reset_timing", "cspm_get_timing",
    "cspm_taps_per_view_pass", "cspm_row_engine_taps_per_view_pass", "cspm_fpm_begin", "cspm_fpm_candidates", "cspm_fpm_commit",
]


class CspmError(RuntimeError):
    pass


class PmParams(C.Structure):
    """struct cspm_pm_params"""
    _fields_ = [("seed", C.c_uint64), ("schedule", C.c_int), ("rb_rounds", C.c_int), ("rb_neighbours", C.c_int),
                ("rng_mode", C.c_int), ("early_exit", C.c_int)]


def library_path():
    return _SO


def build_library(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + [os.path.join(_HERE, "..", "include", "cspm.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise CspmError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(_SO)
    vp, dp, ip, u8p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    llp = C.POINTER(C.c_longlong)
    pp = C.POINTER(PmParams)
    sig = {
        "cspm_device_count": (C.c_int, []),
        "cspm_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "cspm_destroy": (None, [vp]),
        "cspm_last_error": (C.c_char_p, [vp]),
        "cspm_set_stream": (C.c_int, [vp, vp]),
        "cspm_get_stream": (C.c_int, [vp, C.POINTER(vp)]),
        "cspm_synchronize": (C.c_int, [vp]),
        "cspm_set_images": (C.c_int, [vp, u8p, u8p, C.c_int, C.c_int, C.c_size_t]),
        "cspm_set_images_device": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_size_t]),
        "cspm_build_cost_grd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double]),
        "cspm_set_option": (C.c_int, [vp, C.c_int, C.c_longlong]),
        "cspm_get_option": (C.c_int, [vp, C.c_int, llp]),
        "cspm_build_cost_cen": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double]),
        "cspm_build_cost_img": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double]),
        "cspm_cen_build_cv_host": (C.c_int, [C.c_int, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp]),
        "cspm_begin_cost": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double]),
        "cspm_upload_cost_slab": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, dp, C.c_size_t]),
        "cspm_finish_cost": (C.c_int, [vp]),
        "cspm_get_levels": (C.c_int, [vp]),
        "cspm_get_level_dims": (C.c_int, [vp, C.c_int, ip, ip, ip]),
        "cspm_get_level_image": (C.c_int, [vp, C.c_int, C.c_int, u8p]),
        "cspm_get_cost_slab": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, dp]),
        "cspm_get_max_cost": (C.c_int, [vp, C.c_int, C.c_int, dp]),
        "cspm_get_scale_weights": (C.c_int, [vp, dp]),
        "cspm_grd_build_cv_host": (C.c_int, [C.c_int, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp]),
        "cspm_plane_cost_batch": (C.c_int, [vp, C.c_int, C.c_int, ip, dp, dp]),
        "cspm_pm_default_params": (C.c_int, [pp]),
        "cspm_patchmatch": (C.c_int, [vp, C.c_int, pp]),
        "cspm_pm_init": (C.c_int, [vp, pp]),
        "cspm_pm_spatial": (C.c_int, [vp, C.c_int, pp]),
        "cspm_pm_view": (C.c_int, [vp, C.c_int, pp]),
        "cspm_pm_refine": (C.c_int, [vp, C.c_int, pp]),
        "cspm_get_planes": (C.c_int, [vp, C.c_int, dp, dp]),
        "cspm_set_planes": (C.c_int, [vp, C.c_int, dp, dp]),
        "cspm_get_disparity_u8": (C.c_int, [vp, C.c_int, C.c_int, u8p, C.c_size_t]),
        "cspm_get_disparity_f64": (C.c_int, [vp, C.c_int, dp]),
        "cspm_disparity_u8_device": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "cspm_postprocess": (C.c_int, [vp, C.c_int, u8p, u8p, C.c_size_t]),
        "cspm_postprocess_device": (C.c_int, [vp, C.c_int, vp, vp]),
        "cspm_enable_timing": (C.c_int, [vp, C.c_int]),
        "cspm_reset_timing": (C.c_int, [vp]),
        "cspm_get_timing": (C.c_int, [vp, C.c_int, llp, dp, llp]),
        "cspm_fpm_begin": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "cspm_fpm_candidates": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, pp, ip, ip, ip, dp]),
        "cspm_fpm_commit": (C.c_int, [vp, dp]),
        "cspm_taps_per_view_pass": (C.c_longlong, [vp]),
        "cspm_row_engine_taps_per_view_pass": (C.c_longlong, [vp]),
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class StereoContext:
    """One cspm_ctx: one stereo pair on one GPU / stream."""

    def __init__(self, device=0):
        self.L = load_library()
        self.p = C.c_void_p()
        rc = self.L.cspm_create(C.byref(self.p), device)
        if rc != 0:
            raise CspmError(f"cspm_create failed ({rc}): {self.L.cspm_last_error(None).decode()}")
        self.w = self.h = 0

    def close(self):
        if getattr(self, "p", None) and self.p:
            self.L.cspm_destroy(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise CspmError(f"cspm error {rc}: {self.L.cspm_last_error(self.p).decode()}")

    # ---- images / cost ----
    def set_images(self, l_bgr, r_bgr):
        l = np.ascontiguousarray(l_bgr, dtype=np.uint8)
        r = np.ascontiguousarray(r_bgr, dtype=np.uint8)
        assert l.ndim == 3 and l.shape[2] == 3 and l.shape == r.shape
        self.h, self.w = l.shape[:2]
        self._chk(self.L.cspm_set_images(self.p, _u8(l), _u8(r), self.w, self.h, self.w * 3))

    def set_images_device(self, d_l_ptr, d_r_ptr, w, h, stride=None):
        self.h, self.w = h, w
        self._chk(self.L.cspm_set_images_device(self.p, C.c_void_p(d_l_ptr), C.c_void_p(d_r_ptr), w, h, stride or w * 3))

    def set_stream(self, stream_ptr):
        self._chk(self.L.cspm_set_stream(self.p, C.c_void_p(stream_ptr)))

    def stream_ptr(self):
        """the hipStream_t the context enqueues on, as an integer (torch.cuda.ExternalStream(ptr))"""
        p = C.c_void_p()
        self._chk(self.L.cspm_get_stream(self.p, C.byref(p)))
        return p.value or 0

    def synchronize(self):
        self._chk(self.L.cspm_synchronize(self.p))

    def set_option(self, key, value):
        self._chk(self.L.cspm_set_option(self.p, key, value))

    def get_option(self, key):
        v = C.c_longlong()
        self._chk(self.L.cspm_get_option(self.p, key, C.byref(v)))
        return v.value

    def build_cost_grd(self, max_dis, wnd_size=35, scale_num=0, reg_lambda=0.0, volumes=False, sweep_pairs=None, table_volumes=None):
        """volumes=False: fused on-the-fly GRD cells (default); True: materialised f64 cost volumes.
        sweep_pairs (CSPM_OPT_SWEEP_PAIRS): None = leave the context's setting alone (the library's default, off, or what the
        environment variable CSPM_SWEEP_PAIRS / an earlier set_option chose); False = the raster sweep recomputes its cells from image
        gathers like every other kernel; True = paired-cell volumes for the sweep when they fit the context's budget (an option that
        measured no faster).
        table_volumes (CSPM_OPT_TABLE_VOLUMES): None = leave the context's setting alone (default on; CSPM_TABLE_VOLUMES=0 turns it
        off); True = device-cell volumes (when they fit) from which the row kernels fill their cell tables by LDS-DMA; False = the
        tables are computed."""
        self.set_option(OPT_GRD_VOLUMES, int(volumes))
        if sweep_pairs is not None:
            self.set_option(OPT_SWEEP_PAIRS, int(bool(sweep_pairs)))
        if table_volumes is not None:
            self.set_option(OPT_TABLE_VOLUMES, int(bool(table_volumes)))
        self._chk(self.L.cspm_build_cost_grd(self.p, max_dis, wnd_size, scale_num, reg_lambda))

    def build_cost_cen(self, max_dis, wnd_size=35, scale_num=0, reg_lambda=0.0, volumes=False):
        self.set_option(OPT_GRD_VOLUMES, int(volumes))
        self._chk(self.L.cspm_build_cost_cen(self.p, max_dis, wnd_size, scale_num, reg_lambda))

    def build_cost_img(self, max_dis, wnd_size=35, scale_num=0, reg_lambda=0.0):
        """GrdPC (scale_num=0) / CSPC: the volume-free plane costs (plane_cost/grd_pc.cc, cspc.cc)"""
        self._chk(self.L.cspm_build_cost_img(self.p, max_dis, wnd_size, scale_num, reg_lambda))

    def begin_cost(self, max_dis, wnd_size=35, scale_num=0, reg_lambda=0.0):
        self._chk(self.L.cspm_begin_cost(self.p, max_dis, wnd_size, scale_num, reg_lambda))

    def upload_cost_slab(self, view, level, d, slab):
        s = np.ascontiguousarray(slab, dtype=np.float64)
        self._chk(self.L.cspm_upload_cost_slab(self.p, view, level, d, _dp(s), s.shape[1]))

    def finish_cost(self):
        self._chk(self.L.cspm_finish_cost(self.p))

    @property
    def levels(self):
        return self.L.cspm_get_levels(self.p)

    def level_dims(self, s):
        w, h, d = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.cspm_get_level_dims(self.p, s, C.byref(w), C.byref(h), C.byref(d)))
        return w.value, h.value, d.value

    def level_image(self, view, s):
        w, h, _ = self.level_dims(s)
        o = np.zeros((h, w, 3), np.uint8)
        self._chk(self.L.cspm_get_level_image(self.p, view, s, _u8(o)))
        return o

    def cost_slab(self, view, s, d):
        w, h, _ = self.level_dims(s)
        o = np.zeros((h, w))
        self._chk(self.L.cspm_get_cost_slab(self.p, view, s, d, _dp(o)))
        return o

    def cost_volume(self, view, s):
        _, _, D = self.level_dims(s)
        return np.stack([self.cost_slab(view, s, d) for d in range(D + 1)])

    def max_cost(self, view, s):
        o = C.c_double()
        self._chk(self.L.cspm_get_max_cost(self.p, view, s, C.byref(o)))
        return o.value

    def scale_weights(self):
        o = np.zeros(MAX_LEVELS)
        self._chk(self.L.cspm_get_scale_weights(self.p, _dp(o)))
        return o[:self.levels].copy()

    # ---- GetPlaneCost ----
    def plane_cost_batch(self, view, xy, norm_param):
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        npar = np.ascontiguousarray(norm_param, dtype=np.float64).reshape(-1, 6)
        assert len(xy) == len(npar)
        out = np.zeros(len(xy))
        self._chk(self.L.cspm_plane_cost_batch(self.p, view, len(xy), xy.ctypes.data_as(C.POINTER(C.c_int)), _dp(npar), _dp(out)))
        return out

    # ---- PatchMatch ----
    def params(self, seed=12345, schedule=SCHED_RASTER, rb_rounds=1, rb_neighbours=4, rng_mode=RNG_PER_PIXEL, early_exit=1):
        return PmParams(seed, schedule, rb_rounds, rb_neighbours, rng_mode, early_exit)

    def patchmatch(self, iters=3, **kw):
        p = self.params(**kw)
        self._chk(self.L.cspm_patchmatch(self.p, iters, C.byref(p)))

    def pm_init(self, **kw):
        p = self.params(**kw)
        self._chk(self.L.cspm_pm_init(self.p, C.byref(p)))

    def pm_spatial(self, it, **kw):
        p = self.params(**kw)
        self._chk(self.L.cspm_pm_spatial(self.p, it, C.byref(p)))

    def pm_view(self, it, **kw):
        p = self.params(**kw)
        self._chk(self.L.cspm_pm_view(self.p, it, C.byref(p)))

    def pm_refine(self, it, **kw):
        p = self.params(**kw)
        self._chk(self.L.cspm_pm_refine(self.p, it, C.byref(p)))

    def get_planes(self, view):
        npar = np.zeros((self.h, self.w, 6))
        cost = np.zeros((self.h, self.w))
        self._chk(self.L.cspm_get_planes(self.p, view, _dp(npar), _dp(cost)))
        return npar, cost

    def set_planes(self, view, norm_param, min_cost):
        npar = np.ascontiguousarray(norm_param, dtype=np.float64)
        cost = np.ascontiguousarray(min_cost, dtype=np.float64)
        assert npar.shape == (self.h, self.w, 6) and cost.shape == (self.h, self.w)
        self._chk(self.L.cspm_set_planes(self.p, view, _dp(npar), _dp(cost)))

    def disparity_u8(self, view, dis_scale):
        o = np.zeros((self.h, self.w), np.uint8)
        self._chk(self.L.cspm_get_disparity_u8(self.p, view, dis_scale, _u8(o), self.w))
        return o

    def disparity_f64(self, view):
        o = np.zeros((self.h, self.w))
        self._chk(self.L.cspm_get_disparity_f64(self.p, view, _dp(o)))
        return o

    def disparity_u8_device(self, view, dis_scale, d_out_ptr):
        self._chk(self.L.cspm_disparity_u8_device(self.p, view, dis_scale, C.c_void_p(d_out_ptr)))

    def postprocess(self, dis_scale):
        l = np.zeros((self.h, self.w), np.uint8)
        r = np.zeros((self.h, self.w), np.uint8)
        self._chk(self.L.cspm_postprocess(self.p, dis_scale, _u8(l), _u8(r), self.w))
        return l, r

    def postprocess_device(self, dis_scale, d_l_ptr, d_r_ptr):
        """PlaneToDisp + PostProcessing with device-resident outputs (asynchronous on the context's stream)"""
        self._chk(self.L.cspm_postprocess_device(self.p, dis_scale, C.c_void_p(d_l_ptr), C.c_void_p(d_r_ptr)))

    # ---- measurement ----
    def enable_timing(self, on=True):
        self._chk(self.L.cspm_enable_timing(self.p, int(on)))

    def reset_timing(self):
        self._chk(self.L.cspm_reset_timing(self.p))

    def timing(self):
        out = {}
        for k, name in enumerate(K_NAMES):
            n, ms, ev = C.c_longlong(), C.c_double(), C.c_longlong()
            self._chk(self.L.cspm_get_timing(self.p, k, C.byref(n), C.byref(ms), C.byref(ev)))
            out[name] = {"launches": n.value, "ms": ms.value, "evals": ev.value}
        return out

    def taps_per_view_pass(self):
        return self.L.cspm_taps_per_view_pass(self.p)

    def row_engine_taps_per_view_pass(self):
        return self.L.cspm_row_engine_taps_per_view_pass(self.p)
