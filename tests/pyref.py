"""A SECOND, independent restatement of the reference path in plain numpy/Python loops, written from the
reference sources (file:line cited) without looking at oracle/cspm_oracle.c's structure.  Small cases only.
It exists to catch restatement mistakes in the C oracle: two independent readings of the same C++ must agree
bit for bit.  (Random draws are taken from the oracle's counter-based generator: the reference's
cv::RNG(time(NULL)) stream is not reproducible.)"""
import math
from fractions import Fraction

import numpy as np

from oracle import pyoracle as po

EPS = 1e-8          # commfunc.h:26
DMAX = np.finfo(np.float64).max


def round2int(d):   # commfunc.h:117-121
    return int(np.rint(d))


def reflect101(p, n):
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * (n - 1) - p
    return p


def pyrdown(img):   # OpenCV 2.4 pyrDown 8UC3, pre_cs_pc.cc:45
    h, w = img.shape[:2]
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = [1, 4, 6, 4, 1]
    out = np.zeros((dh, dw, 3), np.uint8)
    for y in range(dh):
        for x in range(dw):
            acc = np.zeros(3, np.int64)
            for j in range(5):
                for i in range(5):
                    acc += k[j] * k[i] * img[reflect101(2 * y + j - 2, h), reflect101(2 * x + i - 2, w)].astype(np.int64)
            out[y, x] = (acc + 128) >> 8
    return out


def gray_grad(bgr):  # grd_cc.cpp:70-77 on the RGB-converted CV_64F image
    r, g, b = (bgr[..., 2].astype(np.float32), bgr[..., 1].astype(np.float32), bgr[..., 0].astype(np.float32))
    gray = (r * np.float32(0.299) + g * np.float32(0.587)) + b * np.float32(0.114)
    h, w = gray.shape
    G = np.zeros((h, w))
    for x in range(w):
        G[:, x] = gray[:, reflect101(x + 1, w)].astype(np.float64) - gray[:, reflect101(x - 1, w)].astype(np.float64)
    return G


def fma(a, b, c):
    """a*b + c with ONE rounding, computed exactly in rational arithmetic (no libm, no hardware fma)."""
    a, b, c = float(a), float(b), float(c)
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return a * b + c
    r = Fraction(a) * Fraction(b) + Fraction(c)
    if r == 0:
        return a * b + c  # keeps the sign of zero of the unfused expression (equal for +0.0 accumulators)
    return float(r)  # int / int true division: correctly rounded


def grd_volume(l_bgr, r_bgr, max_dis_slabs, right, dev=False):  # grd_cc.cpp:60-154 + myCostGrd :4-35
    L = l_bgr[..., ::-1].astype(np.float64)  # RGB
    R = r_bgr[..., ::-1].astype(np.float64)
    lG, rG = gray_grad(l_bgr), gray_grad(r_bgr)
    h, w = lG.shape
    vol = np.zeros((max_dis_slabs, h, w))
    for d in range(max_dis_slabs):
        for x in range(w):
            if not right:
                own_c, own_g = L[:, x], lG[:, x]
                if x - d >= 0:
                    oth_c, oth_g = R[:, x - d], rG[:, x - d]
                else:
                    oth_c, oth_g = np.full((h, 3), 3.0), np.full(h, 3.0)
            else:
                own_c, own_g = R[:, x], rG[:, x]
                if x + d < w:
                    oth_c, oth_g = L[:, x + d], lG[:, x + d]
                else:
                    oth_c, oth_g = np.full((h, 3), 3.0), np.full(h, 3.0)
            clr = np.zeros(h)
            for c in range(3):
                clr = clr + np.abs(own_c[:, c] - oth_c[:, c])
            clr = clr * 0.3333333333
            grd = np.abs(own_g - oth_g)
            clr = np.where(clr > 10.0, 10.0, clr)
            grd = np.where(grd > 2.0, 2.0, grd)
            if dev:  # the device order's cell: the last multiply-add is one fma
                vol[d, :, x] = [fma(1 - 0.1, g, 0.1 * c) for c, g in zip(clr, grd)]
            else:
                vol[d, :, x] = 0.1 * clr + (1 - 0.1) * grd
    return vol


def cen_volume(l_bgr, r_bgr, max_dis_slabs, right):  # cen_cc.cc:4-137
    def gray8(bgr):  # convertTo(CV_8U) of u8-valued doubles, cvtColor(CV_RGB2GRAY) 8U fixed point
        b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
        return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.int64)

    def codes(gray):
        h, w = gray.shape
        out = np.zeros((h, w, 80), bool)
        k = 0
        for wy in range(-4, 5):
            for wx in range(-4, 5):
                if wy == 0 and wx == 0:
                    continue
                out[..., k] = gray > np.roll(np.roll(gray, -wy, axis=0), -wx, axis=1)  # neighbour ((y+wy) mod h, (x+wx) mod w)
                k += 1
        return out
    lc, rc = codes(gray8(l_bgr)), codes(gray8(r_bgr))
    h, w = lc.shape[:2]
    vol = np.full((max_dis_slabs, h, w), 80.0)
    for d in range(max_dis_slabs):
        if not right:
            if d < w:
                vol[d, :, d:] = (lc[:, d:] ^ rc[:, :w - d]).sum(-1)
        else:
            if d < w:
                vol[d, :, :w - d] = (rc[:, :w - d] ^ lc[:, d:]).sum(-1)
    return vol


def img_grad(bgr):  # grd_pc.cc:37-40 / cspc.cc:55-58: cvtColor(8UC3, BGR2GRAY) fixed point, Sobel(.., CV_64F, 1, 0, 1)
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    gray = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14
    h, w = gray.shape
    out = np.zeros((h, w))
    for x in range(w):
        out[:, x] = gray[:, reflect101(x + 1, w)] - gray[:, reflect101(x - 1, w)]
    return out


def plane_param(n, p):  # plane.h:25-34
    den = max(abs(n[2]), EPS)
    if n[2] < 0.0:
        den = -den
    s = n[0] * p[0]
    s += n[1] * p[1]
    s += n[2] * p[2]
    return np.array([-n[0] / den, -n[1] / den, s / den])


class PlaneCost:
    """PreSSPC (scale_num=0) / PreCSPC: pre_ss_pc.cc, pre_cs_pc.cc"""

    def __init__(self, l, r, max_disp, wnd, scale_num, lam, cc="GRD", dev=True):
        build = cen_volume if cc == "CEN" else grd_volume
        self.cs = scale_num > 0
        S = scale_num if self.cs else 1
        self.half = wnd // 2
        self.img = [[l], [r]]
        self.dims = [(l.shape[1], l.shape[0], max_disp)]
        for s in range(1, S):
            for v in (0, 1):
                self.img[v].append(pyrdown(self.img[v][s - 1]))
            w, h, d = self.dims[-1]
            self.dims.append(((w + 1) // 2, (h + 1) // 2, d // 2))
        self.img_kind = cc == "IMG"
        if self.img_kind:
            # GrdPC (grd_pc.cc:27-49) / CSPC (cspc.cc:37-61): 8U gray, Sobel [-1 0 1] -> CV_64F; no volumes
            self.grd = [[img_grad(self.img[v][s]) for s in range(S)] for v in (0, 1)]
            self.vol = [[None] * S, [None] * S]
            self.max_cost = [[0.1 * 10.0 + (1 - 0.1) * 2.0] * S for v in (0, 1)]  # grd_pc.cc:131-132, cspc.cc:150-152
            self.vol_dev, self.max_cost_dev = self.vol, self.max_cost
        else:
            self.vol = [[build(self.img[0][s], self.img[1][s], self.dims[s][2] + 1, v == 1) for s in range(S)] for v in (0, 1)]
            self.max_cost = [[max(-1.0, float(self.vol[v][s].max())) for s in range(S)] for v in (0, 1)]
            # what the device order reads: GRD cells with the contracted last step; census cells are integers
            self.vol_dev, self.max_cost_dev = self.vol, self.max_cost
            if cc == "GRD" and dev:
                self.vol_dev = [[grd_volume(self.img[0][s], self.img[1][s], self.dims[s][2] + 1, v == 1, dev=True) for s in range(S)]
                                for v in (0, 1)]
                self.max_cost_dev = [[max(-1.0, float(self.vol_dev[v][s].max())) for s in range(S)] for v in (0, 1)]
        if self.cs:
            M = np.zeros((S, S))
            for s in range(S):
                M[s, s] = 1 + lam if s in (0, S - 1) else 1 + 2 * lam
                if s > 0: M[s, s - 1] = -lam
                if s < S - 1: M[s, s + 1] = -lam
            if S == 1:
                M[0, 0] = 1 + lam
            self.wgt = po.PlaneCost.__new__(po.PlaneCost)  # placeholder, weights are taken from the oracle's LU below
            o = np.zeros(S)
            import ctypes as C
            po.lib().csor_scale_weights(S, lam, o.ctypes.data_as(C.POINTER(C.c_double)))
            np.testing.assert_allclose(o, np.linalg.inv(M)[0], rtol=1e-13)
            self.wgt = o
        else:
            self.wgt = np.array([1.0])
        self.lut = np.array([math.exp(-i * 1.0 / 10.0) for i in range(1000)])

    def _level(self, v, s, cx, cy, a, b, c, rowmod=0):
        """rowmod == 0: the reference's single running sum.  rowmod == K > 0: the device order of DESIGN.md 3.2 -- per window
        row K interleaved partial sums (window column % K) combined left to right; the row totals (zero-padded to 64) are
        combined by a balanced binary tree (neighbours first)."""
        w, h, D = self.dims[s]
        img = self.img[v][s].astype(np.int64)
        vol, maxc = (self.vol_dev[v][s], self.max_cost_dev[v][s]) if rowmod else (self.vol[v][s], self.max_cost[v][s])
        cost = 0.0
        rows = [0.0] * 64
        Ip = img[cy, cx]
        for dy in range(-self.half, self.half + 1):
            qy = cy + dy
            if qy < 0 or qy >= h:
                continue
            qdy = b * qy + c
            part = [0.0] * max(rowmod, 1)
            for dx in range(-self.half, self.half + 1):
                qx = cx + dx
                if qx < 0 or qx >= w:
                    continue
                wgt = self.lut[int(np.abs(Ip - img[qy, qx]).sum())]
                if rowmod:
                    # device order: the disparity is formed per group of `rowmod` window columns, two fused multiply-adds
                    j = (dx + self.half) % rowmod
                    qd = fma(a, float(j), fma(a, float(qx - j), qdy))
                else:
                    qd = a * qx + qdy
                f = int(qd) if (qd == qd and abs(qd) < 2 ** 31) else -(2 ** 31)  # cvttsd2si
                if f <= 0 or f >= D:
                    val = maxc
                elif self.img_kind:
                    val = self._img_cell(v, s, qx, qy, qd)
                elif rowmod:
                    c0, c1 = vol[f, qy, qx], vol[f + 1, qy, qx]
                    val = fma(qd - f, c1 - c0, c0)  # c0 + fr*(c1 - c0), one rounding
                else:
                    fw = (f + 1) - qd
                    val = fw * vol[f, qy, qx] + (1 - fw) * vol[f + 1, qy, qx]
                if rowmod:
                    part[(dx + self.half) % rowmod] = fma(wgt, val, part[(dx + self.half) % rowmod])
                else:
                    cost += wgt * val
            if rowmod:
                row = part[0]
                for j in range(1, rowmod):
                    row = row + part[j]
                rows[dy + self.half] = row
        if rowmod:
            while len(rows) > 1:  # neighbours first: (r0+r1), (r2+r3), ... then pairs of pairs
                rows = [rows[i] + rows[i + 1] for i in range(0, len(rows), 2)]
            cost = rows[0]
        return cost

    def _img_cell(self, v, s, qx, qy, qd):
        """grd_pc.cc:151-168 / cspc.cc:154-173: colour and gradient of the other view interpolated at qx -+ q_disp"""
        w = self.dims[s][0]
        ox = qx + (2 * v - 1) * qd
        fx = int(ox)  # static_cast<int>: towards zero
        cxx = fx + 1
        fw = cxx - ox
        wrap = lambda t: t + w if t < 0 else (t - w if t >= w else t)  # commfunc.h:129-145
        fx, cxx = wrap(fx), wrap(cxx)
        Iq = self.img[v][s][qy, qx].astype(np.int64)
        If = self.img[1 - v][s][qy, fx].astype(np.int64)
        Ic = self.img[1 - v][s][qy, cxx].astype(np.int64)
        clr = 0.0
        for ch in range(3):
            t = abs(float(int(Iq[ch] - Ic[ch])) + fw * float(int(Ic[ch] - If[ch])))
            clr = t if ch == 0 else clr + t
        clr *= 0.33333333333333
        G = self.grd
        grd = abs(G[v][s][qy, qx] - G[1 - v][s][qy, cxx] + fw * (G[1 - v][s][qy, cxx] - G[1 - v][s][qy, fx]))
        return 0.1 * min(clr, 10.0) + (1 - 0.1) * min(grd, 2.0)

    def cost(self, x, y, norm, param, v, rowmod=0):
        if not self.cs:
            return self._level(v, 0, x, y, param[0], param[1], param[2], rowmod)
        cost = 0.0
        cur = param[0] * x + param[1] * y + param[2]
        cx, cy = x, y
        for s in range(len(self.dims)):
            a, b, c = plane_param(norm, [float(cx), float(cy), cur])
            cost += self._level(v, s, cx, cy, a, b, c, rowmod) * self.wgt[s]
            cy //= 2; cx //= 2; cur /= 2.0
        return cost


class PatchMatch:
    """cs_patchmatch.cc.  state: planes[v][y][x] = dict(n, p, prm), cost[v][y][x]"""

    def __init__(self, l, r, max_dis, dis_scale, seed):
        self.h, self.w = l.shape[:2]
        self.img = [l.astype(np.int64), r.astype(np.int64)]
        self.max_dis, self.dis_scale, self.seed = max_dis, dis_scale, seed
        self.n = np.zeros((2, self.h, self.w, 3)); self.p = np.zeros((2, self.h, self.w, 3)); self.prm = np.zeros((2, self.h, self.w, 3))
        self.cost = np.full((2, self.h, self.w), DMAX)
        self.dis = np.zeros((2, self.h, self.w), np.uint8)

    def _u(self, sid, x, y, k):
        return po.lib().csor_rng_u01(self.seed, sid, y * self.w + x, k)

    def _try(self, pc, v, x, y, n, p, prm):
        c = pc.cost(x, y, n, prm, v)
        if c < self.cost[v, y, x]:
            self.cost[v, y, x] = c
            self.n[v, y, x], self.p[v, y, x], self.prm[v, y, x] = n, p, prm

    def init(self, pc):  # :115-148 (direction by rejection sampling, see DESIGN.md "RNG")
        for v in (0, 1):
            sid = po.lib().csor_stream_id(0, 0, 0, v)
            for y in range(self.h):
                for x in range(self.w):
                    z = self._u(sid, x, y, 0) * (self.max_dis - EPS) + EPS
                    for t in range(32):
                        rn = np.array([self._u(sid, x, y, 1 + 3 * t + k) * 2.0 + -1.0 for k in range(3)])
                        s = rn[0] * rn[0]; s += rn[1] * rn[1]; s += rn[2] * rn[2]
                        if s <= 1.0 and s > 1e-12:
                            break
                    inv = 1.0 / max(math.sqrt(s), EPS)
                    n = rn * inv
                    p = np.array([float(x), float(y), z])
                    prm = plane_param(n, p)
                    self.n[v, y, x], self.p[v, y, x], self.prm[v, y, x] = n, p, prm
                    self.cost[v, y, x] = pc.cost(x, y, n, prm, v)

    def spatial(self, it, pc):  # :163-216
        w, h = self.w, self.h
        if it % 2 == 0:
            xs, ys, inc = list(range(1, w)), list(range(1, h)), 1
            x0, y0 = 0, 0
        else:
            xs, ys, inc = list(range(w - 2, -1, -1)), list(range(h - 2, -1, -1)), -1
            x0, y0 = w - 1, h - 1
        for v in (0, 1):
            for x in xs:
                self._try(pc, v, x, y0, self.n[v, y0, x - inc].copy(), self.p[v, y0, x - inc].copy(), self.prm[v, y0, x - inc].copy())
            for y in ys:
                self._try(pc, v, x0, y, self.n[v, y - inc, x0].copy(), self.p[v, y - inc, x0].copy(), self.prm[v, y - inc, x0].copy())
                for x in xs:
                    self._try(pc, v, x, y, self.n[v, y, x - inc].copy(), self.p[v, y, x - inc].copy(), self.prm[v, y, x - inc].copy())
                    self._try(pc, v, x, y, self.n[v, y - inc, x].copy(), self.p[v, y - inc, x].copy(), self.prm[v, y - inc, x].copy())

    def view(self, it, pc):  # :229-277
        w, h = self.w, self.h
        xs = list(range(w)) if it % 2 == 0 else list(range(w - 1, -1, -1))
        ys = list(range(h)) if it % 2 == 0 else list(range(h - 1, -1, -1))
        for v in (0, 1):
            o = 1 - v
            for y in ys:
                for x in xs:
                    prm = self.prm[o, y, x]
                    d = prm[0] * x + prm[1] * y + prm[2]
                    if d < 0.0: d = 0.0
                    if d >= self.max_dis: d = self.max_dis - 1.0
                    cx = x + round2int(d) if v == 0 else x - round2int(d)
                    if cx < 0: cx += w
                    elif cx >= w: cx -= w
                    if cx < 0 or cx >= w:
                        continue
                    n = self.n[o, y, x].copy()
                    p = np.array([float(cx), float(y), d])
                    self._try(pc, v, cx, y, n, p, plane_param(n, p))

    def refine(self, it, pc):  # :292-345
        z, nn, step = self.max_dis / 2.0, 1.0, 0
        while z >= 0.1:
            for v in (0, 1):
                sid = po.lib().csor_stream_id(1, it, step, v)
                for y in range(self.h):
                    for x in range(self.w):
                        prm = self.prm[v, y, x]
                        dz = prm[0] * x + prm[1] * y + prm[2]
                        p = np.array([float(x), float(y), dz + (self._u(sid, x, y, 0) * (z - -z) + -z)])
                        dn = self.n[v, y, x] + np.array([self._u(sid, x, y, 1 + k) * (nn - -nn) + -nn for k in range(3)])
                        s = dn[0] * dn[0]; s += dn[1] * dn[1]; s += dn[2] * dn[2]
                        n = dn * (1.0 / max(math.sqrt(s), EPS))
                        self._try(pc, v, x, y, n, p, plane_param(n, p))
            z /= 2.0; nn /= 2.0; step += 1

    def plane_to_disp(self):  # :590-601
        for v in (0, 1):
            for y in range(self.h):
                for x in range(self.w):
                    a, b, c = self.prm[v, y, x]
                    d = a * x; d += b * y; d += c * 1.0
                    self.dis[v, y, x] = min(255, max(0, round2int(d * self.dis_scale)))

    def run(self, iters, pc):
        self.init(pc)
        for i in range(iters):
            self.spatial(i, pc); self.view(i, pc); self.refine(i, pc)
        self.plane_to_disp()

    def _pdisp(self, v, x, y, xs):
        a, b, c = self.prm[v, y, xs]
        d = a * x; d += b * y; d += c * 1.0
        return d

    def postprocess(self):  # :347-588
        w, h, sc = self.w, self.h, self.dis_scale
        valid = np.zeros((2, h, w), np.int32)
        for v in (0, 1):  # LeftRightCheck
            for y in range(h):
                for x in range(w):
                    cur = self.dis[v, y, x] * 1.0 / sc
                    ox = x + (2 * v - 1) * round2int(cur)
                    if 0 <= ox < w:
                        oth = self.dis[1 - v, y, ox] * 1.0 / sc
                        if abs(cur - oth) <= 0.5 and cur > 0.0:
                            valid[v, y, x] = 1
        for v in (0, 1):  # FillInvalid
            for y in range(h):
                for x in range(w):
                    if valid[v, y, x]:
                        continue
                    lf = next((i for i in range(x, -1, -1) if valid[v, y, i]), None)
                    rf = next((i for i in range(x, w) if valid[v, y, i]), None)
                    if lf is not None and rf is not None:
                        ld, rd = self._pdisp(v, x, y, lf), self._pdisp(v, x, y, rf)
                        self.dis[v, y, x] = min(255, max(0, sc * round2int(ld if ld <= rd else rd)))
                    elif lf is not None:
                        self.dis[v, y, x] = min(255, max(0, sc * round2int(self._pdisp(v, x, y, lf))))
                    elif rf is not None:
                        self.dis[v, y, x] = min(255, max(0, sc * round2int(self._pdisp(v, x, y, rf))))
        lut = [math.exp(-i * 1.0 / 10.0) for i in range(1000)]
        for v in (0, 1):  # WeightedMedian(35, 10)
            for y in range(h):
                for x in range(w):
                    if valid[v, y, x]:
                        continue
                    hist = [0.0] * 256
                    sw = 0.0
                    for wy in range(-17, 18):
                        qy = y + wy
                        if qy < 0 or qy >= h:
                            continue
                        for wx in range(-17, 18):
                            qx = x + wx
                            if 0 <= qx < w and valid[v, qy, qx]:
                                wgt = lut[int(np.abs(self.img[v][y, x] - self.img[v][qy, qx]).sum())]
                                hist[int(self.dis[v, qy, qx])] += wgt
                                sw += wgt
                    med, acc, md = sw / 2.0, 0.0, 0
                    for d in range(256):
                        acc += hist[d]
                        if acc >= med:
                            md = d
                            break
                    if med > 0.0:
                        self.dis[v, y, x] = md
