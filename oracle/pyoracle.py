"""ctypes view of oracle/libcspm_oracle.so -- the CPU parity oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under crossscalepatchmatch_amd/ imports this module.  PARITY UNPINNED (see
oracle/cspm_oracle.h): the reference needs OpenCV + gflags and cannot be built in this image.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcspm_oracle.so")

LEFT, RIGHT = 0, 1
SUM_SERIAL, SUM_DEVICE = 0, 1
SCHED_RASTER, SCHED_REDBLACK = 0, 1
RNG_PER_PIXEL, RNG_ROW_SHARED = 0, 1


def effective_cpus():
    """CPUs this process may actually use: the visible count, cut by the affinity mask and by the cgroup CPU quota (a container
    that shows 256 CPUs with a quota of 16 runs 256 OpenMP threads SLOWER than 16: measured 4.6 s vs 2.1 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    try:  # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            n = min(n, max(1, int(quota / period + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def build(force=False):
    src = os.path.join(_HERE, "cspm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcspm_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class PmOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("rng_mode", C.c_int), ("schedule", C.c_int), ("sum_order", C.c_int),
                ("rb_rounds", C.c_int), ("rb_neighbours", C.c_int), ("threads", C.c_int), ("wavefront", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp, u8p, ip = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    sig = {
        "csor_round2int": (C.c_int, [C.c_double]),
        "csor_handle_border": (C.c_int, [C.c_int, C.c_int]),
        "csor_plane_param": (None, [dp, dp, dp]),
        "csor_exp_lut": (None, [dp, C.c_double]),
        "csor_scale_weights": (C.c_int, [C.c_int, C.c_double, dp]),
        "csor_rng_u64": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]),
        "csor_rng_u01": (C.c_double, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]),
        "csor_stream_id": (C.c_uint32, [C.c_int] * 4),
        "csor_pyrdown_bgr8": (None, [u8p, C.c_int, C.c_int, u8p]),
        "csor_grd_build_cv": (None, [dp, dp, C.c_int, C.c_int, C.c_int, dp]),
        "csor_grd_build_right_cv": (None, [dp, dp, C.c_int, C.c_int, C.c_int, dp]),
        "csor_rgb2gray_f32": (None, [dp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
        "csor_sobel_x_ks1": (None, [C.POINTER(C.c_float), C.c_int, C.c_int, dp]),
        "csor_pc_create": (C.c_void_p, [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
        "csor_pc_create_cc": (C.c_void_p, [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
        "csor_cen_build_cv": (None, [dp, dp, C.c_int, C.c_int, C.c_int, dp]),
        "csor_cen_build_right_cv": (None, [dp, dp, C.c_int, C.c_int, C.c_int, dp]),
        "csor_pc_destroy": (None, [C.c_void_p]),
        "csor_pc_levels": (C.c_int, [C.c_void_p]),
        "csor_pc_level_dims": (None, [C.c_void_p, C.c_int, ip, ip, ip]),
        "csor_pc_image": (u8p, [C.c_void_p, C.c_int, C.c_int]),
        "csor_pc_volume": (dp, [C.c_void_p, C.c_int, C.c_int]),
        "csor_pc_max_cost": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
        "csor_pc_volume_dev": (dp, [C.c_void_p, C.c_int, C.c_int]),
        "csor_pc_max_cost_dev": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
        "csor_pc_refresh_max_cost": (None, [C.c_void_p]),
        "csor_pc_scale_wgt": (dp, [C.c_void_p]),
        "csor_pc_cost": (C.c_double, [C.c_void_p, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int]),
        "csor_pc_cost_thresh": (C.c_double, [C.c_void_p, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, C.c_double,
                                             C.POINTER(C.c_longlong)]),
        "csor_pc_taps": (C.c_longlong, [C.c_void_p, C.c_int, C.c_int]),
        "csor_pc_level_costs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, dp]),
        "csor_pm_create": (C.c_void_p, [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]),
        "csor_pm_destroy": (None, [C.c_void_p]),
        "csor_pm_run": (None, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(PmOpts)]),
        "csor_pm_init": (None, [C.c_void_p, C.c_void_p, C.POINTER(PmOpts)]),
        "csor_pm_spatial": (None, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(PmOpts)]),
        "csor_pm_view": (None, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(PmOpts)]),
        "csor_pm_refine": (None, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(PmOpts)]),
        "csor_pm_plane_to_disp": (None, [C.c_void_p]),
        "csor_pm_postprocess": (None, [C.c_void_p]),
        "csor_pm_dis": (u8p, [C.c_void_p, C.c_int]),
        "csor_pm_planes": (dp, [C.c_void_p, C.c_int]),
        "csor_pm_min_cost": (dp, [C.c_void_p, C.c_int]),
        "csor_pm_disp_f64": (None, [C.c_void_p, C.c_int, dp]),
        "csor_pm_evals": (C.c_longlong, [C.c_void_p]),
        "csor_refine_steps": (C.c_int, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _bgr(img):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    return a


class PlaneCost:
    """PreSSPC (scale_num=0) / PreCSPC (scale_num>=1) with the GRD or CEN cost; cc="IMG": GrdPC (scale_num=0) / CSPC."""

    def __init__(self, l_bgr, r_bgr, max_disp, wnd_size=35, scale_num=0, reg_lambda=0.0, cc="GRD"):
        self.L = lib()
        self.l, self.r = _bgr(l_bgr), _bgr(r_bgr)
        self.h, self.w = self.l.shape[:2]
        self.p = self.L.csor_pc_create_cc(_u8(self.l), _u8(self.r), self.w, self.h, max_disp, wnd_size, scale_num,
                                          reg_lambda, {"GRD": 0, "CEN": 1, "IMG": 2}[cc])
        if not self.p:
            raise ValueError("csor_pc_create failed")
        self.levels = self.L.csor_pc_levels(self.p)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.csor_pc_destroy(self.p)
            self.p = None

    def dims(self, s):
        w, h, d = C.c_int(), C.c_int(), C.c_int()
        self.L.csor_pc_level_dims(self.p, s, C.byref(w), C.byref(h), C.byref(d))
        return w.value, h.value, d.value

    def image(self, view, s):
        w, h, _ = self.dims(s)
        return np.ctypeslib.as_array(self.L.csor_pc_image(self.p, view, s), shape=(h, w, 3))

    def volume(self, view, s):
        w, h, d = self.dims(s)
        return np.ctypeslib.as_array(self.L.csor_pc_volume(self.p, view, s), shape=(d + 1, h, w))

    def max_cost(self, view, s):
        return self.L.csor_pc_max_cost(self.p, view, s)

    def volume_dev(self, view, s):
        """the cells a SUM_DEVICE evaluation reads: GRD cells with the contracted last step; otherwise volume()"""
        w, h, d = self.dims(s)
        return np.ctypeslib.as_array(self.L.csor_pc_volume_dev(self.p, view, s), shape=(d + 1, h, w))

    def max_cost_dev(self, view, s):
        return self.L.csor_pc_max_cost_dev(self.p, view, s)

    def refresh_max_cost(self):
        self.L.csor_pc_refresh_max_cost(self.p)

    def scale_wgt(self):
        return np.ctypeslib.as_array(self.L.csor_pc_scale_wgt(self.p), shape=(self.levels,)).copy()

    def cost(self, x, y, norm, param, view, sum_order=SUM_SERIAL):
        n = np.ascontiguousarray(norm, dtype=np.float64)
        p = np.ascontiguousarray(param, dtype=np.float64)
        return self.L.csor_pc_cost(self.p, int(x), int(y), _dp(n), _dp(p), int(view), int(sum_order))

    def cost_thresh(self, x, y, norm, param, view, sum_order, thresh):
        n = np.ascontiguousarray(norm, dtype=np.float64)
        p = np.ascontiguousarray(param, dtype=np.float64)
        taps = C.c_longlong()
        c = self.L.csor_pc_cost_thresh(self.p, int(x), int(y), _dp(n), _dp(p), int(view), int(sum_order),
                                       float(thresh), C.byref(taps))
        return c, taps.value

    def level_costs(self, x, y, norm, param, view, sum_order=SUM_SERIAL):
        n = np.ascontiguousarray(norm, dtype=np.float64)
        p = np.ascontiguousarray(param, dtype=np.float64)
        o = np.zeros(self.levels)
        self.L.csor_pc_level_costs(self.p, int(x), int(y), _dp(n), _dp(p), int(view), int(sum_order), _dp(o))
        return o

    def taps(self, x, y):
        return self.L.csor_pc_taps(self.p, int(x), int(y))


def plane_param(norm, point):
    n = np.ascontiguousarray(norm, dtype=np.float64)
    p = np.ascontiguousarray(point, dtype=np.float64)
    o = np.zeros(3)
    lib().csor_plane_param(_dp(n), _dp(p), _dp(o))
    return o


class PatchMatch:
    """CSPatchMatch."""

    def __init__(self, l_bgr, r_bgr, max_dis, dis_scale):
        self.L = lib()
        self.l, self.r = _bgr(l_bgr), _bgr(r_bgr)
        self.h, self.w = self.l.shape[:2]
        self.p = self.L.csor_pm_create(_u8(self.l), _u8(self.r), self.w, self.h, max_dis, dis_scale)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.csor_pm_destroy(self.p)
            self.p = None

    @staticmethod
    def opts(seed=12345, rng_mode=RNG_PER_PIXEL, schedule=SCHED_RASTER, sum_order=SUM_SERIAL, rb_rounds=1,
             rb_neighbours=4, threads=0, wavefront=False):
        """wavefront: the raster sweep walked anti-diagonal by anti-diagonal in parallel (identical results, a speed knob for the
        whole-KITTI-pair test); the default is the reference's serial double loop"""
        if not threads:  # OpenMP's own default is every visible CPU, whatever the container may use
            threads = effective_cpus()
        return PmOpts(seed, rng_mode, schedule, sum_order, rb_rounds, rb_neighbours, threads, int(bool(wavefront)))

    def run(self, iters, pc, use_pp=False, **kw):
        o = self.opts(**kw)
        self.L.csor_pm_run(self.p, iters, pc.p, int(use_pp), C.byref(o))

    def init(self, pc, **kw):
        o = self.opts(**kw)
        self.L.csor_pm_init(self.p, pc.p, C.byref(o))

    def spatial(self, it, pc, **kw):
        o = self.opts(**kw)
        self.L.csor_pm_spatial(self.p, it, pc.p, C.byref(o))

    def view(self, it, pc, **kw):
        o = self.opts(**kw)
        self.L.csor_pm_view(self.p, it, pc.p, C.byref(o))

    def refine(self, it, pc, **kw):
        o = self.opts(**kw)
        self.L.csor_pm_refine(self.p, it, pc.p, C.byref(o))

    def plane_to_disp(self):
        self.L.csor_pm_plane_to_disp(self.p)

    def postprocess(self):
        self.L.csor_pm_postprocess(self.p)

    def dis(self, view):
        return np.ctypeslib.as_array(self.L.csor_pm_dis(self.p, view), shape=(self.h, self.w)).copy()

    def planes(self, view):
        """(h, w, 9): norm[3], point[3], param[3] -- a live view of the oracle's state."""
        return np.ctypeslib.as_array(self.L.csor_pm_planes(self.p, view), shape=(self.h, self.w, 9))

    def min_cost(self, view):
        return np.ctypeslib.as_array(self.L.csor_pm_min_cost(self.p, view), shape=(self.h, self.w))

    def disp_f64(self, view):
        o = np.zeros((self.h, self.w))
        self.L.csor_pm_disp_f64(self.p, view, _dp(o))
        return o

    def evals(self):
        return self.L.csor_pm_evals(self.p)
