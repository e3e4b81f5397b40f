// cspm_kernels.h -- the HIP kernels of the PatchMatch-stereo hot path (gfx950, wave64).
//
// Mapping: ONE WAVEFRONT EVALUATES ONE CANDIDATE PLANE.  The 64 lanes stride over the linearised
// (2*half+1)^2 support window (tap t -> lane t%64), so the plane parameters, the centre pixel and all
// branch decisions (early exit, accept/reject) are wave-uniform, and the guide-image / cost-volume
// addresses of a wave are runs along image rows of ONE plane: coalesced regardless of how incoherent
// the plane field of neighbouring pixels is (random init, early refinement steps).
// A workgroup is 4 waves = 4 consecutive candidates; blockIdx is remapped so that each XCD (block b
// runs on XCD b%8) walks one contiguous band of the image and keeps its cost-volume rows in its own L2.
#pragma once
#include "cspm_device.h"

#pragma clang fp contract(off)

namespace cspm {

// ------------------------------------------------------------------------------------------------
// IPlaneCost::GetPlaneCost  (PreSSPC: pre_ss_pc.cc:74-118, PreCSPC: pre_cs_pc.cc:133-188)
// ------------------------------------------------------------------------------------------------

// One level: sum over the window of  w(p,q) * lerp(cost_vol[floor], cost_vol[ceil])  (pre_cs_pc.cc:151-181).
// Returns the level sum (identical in all lanes) or -1.0 when base + partial*mul >= thresh was proven.
__device__ __forceinline__ double level_cost(const Cost &cd, const Level &L, const double *s_lut, int view, int s,
                                             int cx, int cy, double a, double b, double c, double base, double mul,
                                             double thresh, bool use_thresh, int lane, int dy0, int dx0) {
  const int W = L.W, H = L.H, half = cd.half, n = cd.n, T = cd.T, groups = cd.groups;
  const double Dd = (double)L.D;
  const double maxc = cd.max_cost[view * CSPM_MAX_LEVELS + s];
  const uint32_t *__restrict__ img = L.img[view];
  const double *__restrict__ vol = L.vol[view];
  const size_t slab = (size_t)W * (size_t)H;
  const uint32_t Ip = img[(size_t)cy * W + cx];
  const int q64 = kWave / n, r64 = kWave % n;
  double part = 0.0;
  int t = lane, dy = dy0, dx = dx0;
  for (int g = 0; g < groups; ++g) {
    const int qy = cy + dy - half, qx = cx + dx - half;
    const bool ok = (t < T) && ((unsigned)qy < (unsigned)H) && ((unsigned)qx < (unsigned)W);
    if (ok) {
      const size_t o = (size_t)qy * W + qx;
      const uint32_t Iq = img[o];
      const int sum = (int)__builtin_amdgcn_sad_u8(Ip, Iq, 0u);  // |dB|+|dG|+|dR| (pre_cs_pc.cc:161-163)
      const double wgt = s_lut[sum];
      const double q_disp_y = b * (double)qy + c;                 // :155
      const double q_disp = a * (double)qx + q_disp_y;            // :165
      // static_cast<int>(q_disp) in 1..D-1  <=>  1.0 <= q_disp < D ; NaN / out of range -> invalid (:166-169)
      const bool valid = (q_disp >= 1.0) && (q_disp < Dd);
      double tmp = maxc;
      if (valid) {
        const int f = (int)q_disp;
        const double floor_wgt = (double)(f + 1) - q_disp;       // :171-172
        const double *c0 = vol + (size_t)f * slab + o;
        tmp = floor_wgt * c0[0] + (1 - floor_wgt) * c0[slab];    // :173-175
      }
      part += wgt * tmp;                                          // :176 / :169
    }
    t += kWave;
    dx += r64;
    dy += q64;
    if (dx >= n) { dx -= n; ++dy; }
    const bool last = (g == groups - 1);
    if (last || (use_thresh && (g % kCheckEvery) == kCheckEvery - 1)) {
      const double tot = wave_sum(part);
      if (use_thresh && base + tot * mul >= thresh) return -1.0;
      if (last) return tot;
    }
  }
  return 0.0;
}

// Aggregated plane cost at (x,y); +inf when the candidate is proven not to beat `thresh`.
// (nx,ny,nz) = Plane::norm(), (pa,pb,pc) = Plane::param().
template <bool CS>
__device__ __forceinline__ double eval_plane(const Cost &cd, const double *s_lut, int view, int x, int y, double nx,
                                             double ny, double nz, double pa, double pb, double pc, double thresh,
                                             bool use_thresh, int lane) {
  const int dy0 = lane / cd.n, dx0 = lane % cd.n;
  if (!CS) {
    const double r = level_cost(cd, cd.lv[0], s_lut, view, 0, x, y, pa, pb, pc, 0.0, 1.0, thresh, use_thresh, lane, dy0, dx0);
    return r < 0.0 ? __builtin_inf() : r;
  }
  double cost = 0.0;
  double cur_disp = pa * (double)x + pb * (double)y + pc;  // pre_cs_pc.cc:139-140
  int cur_x = x, cur_y = y;
  for (int s = 0; s < cd.levels; ++s) {
    double a, b, c;
    plane_param(nx, ny, nz, (double)cur_x, (double)cur_y, cur_disp, a, b, c);  // :144-149
    const double wgt = cd.lv[s].wgt;
    const double sc = level_cost(cd, cd.lv[s], s_lut, view, s, cur_x, cur_y, a, b, c, cost, wgt, thresh, use_thresh, lane, dy0, dx0);
    if (sc < 0.0) return __builtin_inf();
    cost += sc * wgt;  // :182
    cur_y /= 2;        // :183-185
    cur_x /= 2;
    cur_disp /= 2.0;
  }
  return cost;
}

__device__ __forceinline__ void load_lut(const Cost &cd, double *s_lut) {
  for (int i = threadIdx.x; i < kLutSize; i += blockDim.x) s_lut[i] = cd.lut[i];
  __syncthreads();
}

// Work item (candidate) index of this wave.  Blocks are dealt round-robin to the 8 XCDs; give XCD k
// the k-th contiguous eighth of the index space.
__device__ __forceinline__ long long wave_item(long long n_items) {
  const long long nb = (long long)gridDim.x;  // multiple of 8
  const long long per = nb / 8;
  const long long b = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long e = b * (kEvalBlock / kWave) + wave;
  return e < n_items ? e : -1;
}

// ------------------------------------------------------------------------------------------------
// cspm_plane_cost_batch: batched GetPlaneCost on explicit (x,y,plane) tuples -- the parity hook.
// ------------------------------------------------------------------------------------------------
template <bool CS>
__global__ __launch_bounds__(kEvalBlock) void k_cost_batch(Cost cd, int view, int n, const int *__restrict__ xy,
                                                           const double *__restrict__ np, double *__restrict__ out) {
  __shared__ double s_lut[kLutSize];
  load_lut(cd, s_lut);
  const long long e = wave_item(n);
  if (e < 0) return;
  const int lane = threadIdx.x & 63;
  const int x = xy[2 * e], y = xy[2 * e + 1];
  const double *p = np + 6 * e;
  const double c = eval_plane<CS>(cd, s_lut, view, x, y, p[0], p[1], p[2], p[3], p[4], p[5], kDoubleMax, false, lane);
  if (lane == 0) out[e] = c;
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::InitRandomPlane  (cs_patchmatch.cc:115-148)
// ------------------------------------------------------------------------------------------------
template <bool CS>
__global__ __launch_bounds__(kEvalBlock) void k_init(Cost cd, Pm pm) {
  __shared__ double s_lut[kLutSize];
  load_lut(cd, s_lut);
  const long long npix = (long long)pm.W * pm.H;
  const long long e = wave_item(2 * npix);
  if (e < 0) return;
  const int lane = threadIdx.x & 63;
  const int v = (int)(e / npix);
  const long long i = e - (long long)v * npix;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const Rng rng(pm.seed, stream_id(0, 0, 0, v), pm.rng_row_shared ? (uint64_t)x : (uint64_t)i);
  const double rand_dis = rng.uniform(0, kDoubleEps, (double)pm.max_dis);  // :134-135
  // direction: uniform on the sphere by rejection from the unit ball (DESIGN.md "RNG"); :137-140
  double r0 = 0.0, r1 = 0.0, r2 = 1.0, len = 1.0;
  for (int t = 0; t < 32; ++t) {
    r0 = rng.uniform(1 + 3 * t, -1.0, 1.0);
    r1 = rng.uniform(2 + 3 * t, -1.0, 1.0);
    r2 = rng.uniform(3 + 3 * t, -1.0, 1.0);
    double s = r0 * r0;
    s += r1 * r1;
    s += r2 * r2;
    len = __dsqrt_rn(s);
    if (s <= 1.0 && s > 1e-12) break;
  }
  const double inv = 1. / fmax(len, kDoubleEps);
  const double nx = r0 * inv, ny = r1 * inv, nz = r2 * inv;
  double a, b, c;
  plane_param(nx, ny, nz, (double)x, (double)y, rand_dis, a, b, c);  // :141-142
  const double cost = eval_plane<CS>(cd, s_lut, v, x, y, nx, ny, nz, a, b, c, kDoubleMax, false, lane);  // :143-144
  if (lane == 0) {
    const Field &f = pm.f[v];
    f.nx[i] = nx; f.ny[i] = ny; f.nz[i] = nz;
    f.a[i] = a; f.b[i] = b; f.c[i] = c;
    f.cost[i] = cost;
  }
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::PlaneRefinement, one halving step  (cs_patchmatch.cc:303-344)
// ------------------------------------------------------------------------------------------------
template <bool CS>
__global__ __launch_bounds__(kEvalBlock) void k_refine(Cost cd, Pm pm, int iter, int step, double z_iter, double n_iter) {
  __shared__ double s_lut[kLutSize];
  load_lut(cd, s_lut);
  const long long npix = (long long)pm.W * pm.H;
  const long long e = wave_item(2 * npix);
  if (e < 0) return;
  const int lane = threadIdx.x & 63;
  const int v = (int)(e / npix);
  const long long i = e - (long long)v * npix;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const Field &f = pm.f[v];
  const double cnx = f.nx[i], cny = f.ny[i], cnz = f.nz[i], ca = f.a[i], cb = f.b[i], cc = f.c[i];
  const double cur_min = f.cost[i];
  const Rng rng(pm.seed, stream_id(1, iter, step, v), pm.rng_row_shared ? (uint64_t)x : (uint64_t)i);
  const double disturb_z = ca * (double)x + cb * (double)y + cc;             // :317-319
  const double pz = disturb_z + rng.uniform(0, -z_iter, z_iter);             // :320-322
  const double d0 = cnx + rng.uniform(1, -n_iter, n_iter);                   // :324-325
  const double d1 = cny + rng.uniform(2, -n_iter, n_iter);
  const double d2 = cnz + rng.uniform(3, -n_iter, n_iter);
  double s = d0 * d0;
  s += d1 * d1;
  s += d2 * d2;
  const double inv = 1. / fmax(__dsqrt_rn(s), kDoubleEps);                   // :326-328
  const double nx = d0 * inv, ny = d1 * inv, nz = d2 * inv;
  double a, b, c;
  plane_param(nx, ny, nz, (double)x, (double)y, pz, a, b, c);                // :330
  const double cost = eval_plane<CS>(cd, s_lut, v, x, y, nx, ny, nz, a, b, c, cur_min, pm.use_thresh != 0, lane);
  if (cost < cur_min && lane == 0) {                                         // :335-338
    f.nx[i] = nx; f.ny[i] = ny; f.nz[i] = nz;
    f.a[i] = a; f.b[i] = b; f.c[i] = c;
    f.cost[i] = cost;
  }
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::SpatialPropagation.
// k_spatial_rb: red-black half-step (fast path).  k_spatial_diag: one anti-diagonal of the reference's
// in-place raster sweep (cs_patchmatch.cc:163-216); pixels on a diagonal are mutually independent.
// ------------------------------------------------------------------------------------------------
struct Cand { double nx, ny, nz, a, b, c; };

template <bool CS>
__device__ __forceinline__ void try_neighbour(const Cost &cd, const double *s_lut, const Field &f, int v, int x, int y,
                                              long long j, Cand &best, double &best_cost, bool &changed, bool use_thresh,
                                              int lane) {
  const double nx = f.nx[j], ny = f.ny[j], nz = f.nz[j], a = f.a[j], b = f.b[j], c = f.c[j];
  const double cost = eval_plane<CS>(cd, s_lut, v, x, y, nx, ny, nz, a, b, c, best_cost, use_thresh, lane);
  if (cost < best_cost) {
    best_cost = cost;
    best = Cand{nx, ny, nz, a, b, c};
    changed = true;
  }
}

template <bool CS>
__global__ __launch_bounds__(kEvalBlock) void k_spatial_rb(Cost cd, Pm pm, int colour, int inc, int nb) {
  __shared__ double s_lut[kLutSize];
  load_lut(cd, s_lut);
  const int halfW = (pm.W + 1) / 2;
  const long long per_view = (long long)halfW * pm.H;
  const long long e = wave_item(2 * per_view);
  if (e < 0) return;
  const int lane = threadIdx.x & 63;
  const int v = (int)(e / per_view);
  const long long r = e - (long long)v * per_view;
  const int y = (int)(r / halfW);
  const int x = 2 * (int)(r - (long long)y * halfW) + ((y + colour) & 1);
  if (x >= pm.W) return;
  const Field &f = pm.f[v];
  const long long i = (long long)y * pm.W + x;
  Cand best{};
  double best_cost = f.cost[i];
  bool changed = false;
  const int nxs[4] = {x - inc, x, x + inc, x}, nys[4] = {y, y - inc, y, y + inc};
  for (int k = 0; k < nb; ++k) {
    if (nxs[k] < 0 || nxs[k] >= pm.W || nys[k] < 0 || nys[k] >= pm.H) continue;
    try_neighbour<CS>(cd, s_lut, f, v, x, y, (long long)nys[k] * pm.W + nxs[k], best, best_cost, changed, pm.use_thresh != 0, lane);
  }
  if (changed && lane == 0) {
    f.nx[i] = best.nx; f.ny[i] = best.ny; f.nz[i] = best.nz;
    f.a[i] = best.a; f.b[i] = best.b; f.c[i] = best.c;
    f.cost[i] = best_cost;
  }
}

// diagonal k of the sweep: sweep coordinates (xs,ys), xs+ys == k; image x = inc>0 ? xs : W-1-xs.
// A pixel first tries the plane of its x-predecessor, then of its y-predecessor (:198-212); the first
// sweep row has only the former (:178-186), the first sweep column only the latter (:189-195).
template <bool CS>
__global__ __launch_bounds__(kEvalBlock) void k_spatial_diag(Cost cd, Pm pm, int k, int inc) {
  __shared__ double s_lut[kLutSize];
  load_lut(cd, s_lut);
  const int ys_lo = max(0, k - (pm.W - 1)), ys_hi = min(pm.H - 1, k);
  const int cnt = ys_hi - ys_lo + 1;
  const long long e = wave_item(2LL * cnt);
  if (e < 0) return;
  const int lane = threadIdx.x & 63;
  const int v = (int)(e / cnt);
  const int ys = ys_lo + (int)(e - (long long)v * cnt), xs = k - ys;
  const int x = inc > 0 ? xs : pm.W - 1 - xs, y = inc > 0 ? ys : pm.H - 1 - ys;
  const Field &f = pm.f[v];
  const long long i = (long long)y * pm.W + x;
  Cand best{};
  double best_cost = f.cost[i];
  bool changed = false;
  if (xs > 0) try_neighbour<CS>(cd, s_lut, f, v, x, y, i - inc, best, best_cost, changed, pm.use_thresh != 0, lane);
  if (ys > 0) try_neighbour<CS>(cd, s_lut, f, v, x, y, i - (long long)inc * pm.W, best, best_cost, changed, pm.use_thresh != 0, lane);
  if (changed && lane == 0) {
    f.nx[i] = best.nx; f.ny[i] = best.ny; f.nz[i] = best.nz;
    f.a[i] = best.a; f.b[i] = best.b; f.c[i] = best.c;
    f.cost[i] = best_cost;
  }
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::ViewPropagation  (cs_patchmatch.cc:229-277), target view v.
// Phase 1 (k_view_eval): every pixel (x,y) of the OTHER view proposes its plane to pixel (cor_x,y) of
// view v and evaluates it there -- independent, because the pass only reads the other view's planes
// and the candidates of a pass do not depend on each other.
// Phase 2 (k_view_resolve): the serial loop keeps, per target pixel, the candidate with the smallest
// cost that is < the pixel's current cost, the earliest in traversal order among equal costs.  One
// workgroup per row (cor_x stays in row y) reproduces exactly that with LDS atomics.
// ------------------------------------------------------------------------------------------------
struct ViewCand {
  double *cost; // candidate cost, +inf = rejected / none
  double *c;    // candidate param c (a, b follow from the source normal)
  int *cx;      // target column
};

template <bool CS>
__global__ __launch_bounds__(kEvalBlock) void k_view_eval(Cost cd, Pm pm, int v, ViewCand vc) {
  __shared__ double s_lut[kLutSize];
  load_lut(cd, s_lut);
  const long long npix = (long long)pm.W * pm.H;
  const long long i = wave_item(npix);
  if (i < 0) return;
  const int lane = threadIdx.x & 63;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const Field &src = pm.f[1 - v];
  const Field &dst = pm.f[v];
  const double nx = src.nx[i], ny = src.ny[i], nz = src.nz[i];
  double disp = src.a[i] * (double)x + src.b[i] * (double)y + src.c[i];  // :245-246
  if (disp < 0.0) disp = 0.0;                                             // :247-252
  if (disp >= (double)pm.max_dis) disp = (double)pm.max_dis - 1.0;
  const int r = round2int(disp);
  const int cor_x = handle_border(v == 0 ? x + r : x - r, pm.W);          // :255-261
  double cost = __builtin_inf(), a = 0.0, b = 0.0, c = 0.0;
  if (cor_x >= 0 && cor_x < pm.W) {
    plane_param(nx, ny, nz, (double)cor_x, (double)y, disp, a, b, c);     // :263-265
    const double thr = dst.cost[(long long)y * pm.W + cor_x];
    cost = eval_plane<CS>(cd, s_lut, v, cor_x, y, nx, ny, nz, a, b, c, thr, pm.use_thresh != 0, lane);  // :266-267
  }
  if (lane == 0) {
    vc.cost[i] = cost;
    vc.c[i] = c;
    vc.cx[i] = cor_x;
  }
}

__global__ __launch_bounds__(256) void k_view_resolve(Pm pm, int v, int reverse, ViewCand vc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *s_key = (unsigned long long *)smem;          // W entries
  unsigned int *s_rank = (unsigned int *)(s_key + pm.W);           // W entries
  const int y = blockIdx.x, W = pm.W;
  const Field &src = pm.f[1 - v];
  const Field &dst = pm.f[v];
  const long long row = (long long)y * W;
  for (int t = threadIdx.x; t < W; t += blockDim.x) {
    s_key[t] = f64_key(dst.cost[row + t]);
    s_rank[t] = 0xFFFFFFFFu;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const int cx = vc.cx[row + x];
    const double c = vc.cost[row + x];
    if (cx >= 0 && cx < W && c < dst.cost[row + cx]) atomicMin(&s_key[cx], f64_key(c));
  }
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const int cx = vc.cx[row + x];
    const double c = vc.cost[row + x];
    if (cx >= 0 && cx < W && c < dst.cost[row + cx] && f64_key(c) == s_key[cx])
      atomicMin(&s_rank[cx], (unsigned)(reverse ? W - 1 - x : x));
  }
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const int cx = vc.cx[row + x];
    const double c = vc.cost[row + x];
    if (cx >= 0 && cx < W && c < dst.cost[row + cx] && f64_key(c) == s_key[cx] &&
        s_rank[cx] == (unsigned)(reverse ? W - 1 - x : x)) {
      const double nx = src.nx[row + x], ny = src.ny[row + x], nz = src.nz[row + x];
      double denom = fmax(fabs(nz), kDoubleEps);  // a, b of cor_plane.update_param() (plane.h:27-32)
      if (nz < 0.0) denom = -denom;
      // NOTE: dst.cost[row+cx] is written below; every other thread reading it for the same cx lost
      // the rank test above, and __syncthreads() separates the passes.
      dst.nx[row + cx] = nx; dst.ny[row + cx] = ny; dst.nz[row + cx] = nz;
      dst.a[row + cx] = -nx / denom; dst.b[row + cx] = -ny / denom; dst.c[row + cx] = vc.c[row + x];
    }
  }
  __syncthreads();
  // costs last, so the `c < dst.cost` tests of the pass above saw the pre-pass values
  for (int t = threadIdx.x; t < W; t += blockDim.x)
    if (s_rank[t] != 0xFFFFFFFFu) dst.cost[row + t] = key_f64(s_key[t]);
}

// ------------------------------------------------------------------------------------------------
// PlaneToDisp (cs_patchmatch.cc:590-601)
// ------------------------------------------------------------------------------------------------
__global__ void k_plane_to_disp_u8(Pm pm, int v, int dis_scale, uint8_t *__restrict__ out, size_t stride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)pm.W * pm.H) return;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const Field &f = pm.f[v];
  double d = f.a[i] * (double)x;  // param().dot(Vec3d(x, y, 1.0))
  d += f.b[i] * (double)y;
  d += f.c[i] * 1.0;
  int q = round2int(d * (double)dis_scale);
  q = q < 0 ? 0 : (q > 255 ? 255 : q);  // saturate_cast<uchar>
  out[(size_t)y * stride + x] = (uint8_t)q;
}
__global__ void k_plane_to_disp_f64(Pm pm, int v, double *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)pm.W * pm.H) return;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const Field &f = pm.f[v];
  double d = f.a[i] * (double)x;
  d += f.b[i] * (double)y;
  d += f.c[i] * 1.0;
  out[i] = d;
}

// ------------------------------------------------------------------------------------------------
// Image preparation: BGR8 -> packed u32, pyrDown (pre_cs_pc.cc:45), gray + x-gradient (grd_cc.cpp:70-77)
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_bgr(const uint8_t *__restrict__ src, size_t stride, int W, int H, uint32_t *__restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const uint8_t *p = src + (size_t)y * stride + 3 * (size_t)x;
  dst[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
__global__ void k_unpack_bgr(const uint32_t *__restrict__ src, int W, int H, uint8_t *__restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const uint32_t p = src[i];
  dst[3 * i] = (uint8_t)p; dst[3 * i + 1] = (uint8_t)(p >> 8); dst[3 * i + 2] = (uint8_t)(p >> 16);
}

__device__ __forceinline__ int reflect101(int p, int len) {  // cv::borderInterpolate(BORDER_REFLECT_101)
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// OpenCV 2.4 pyrDown on 8UC3: separable [1 4 6 4 1], integer accumulate, (v+128)>>8, REFLECT_101,
// dst = ((W+1)/2, (H+1)/2).  One thread per destination pixel (25 taps; the pyramid is <0.1 % of the work).
__global__ void k_pyrdown(const uint32_t *__restrict__ src, int W, int H, uint32_t *__restrict__ dst, int dW, int dH) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)dW * dH) return;
  const int y = (int)(i / dW), x = (int)(i - (long long)y * dW);
  const int kw[5] = {1, 4, 6, 4, 1};
  int acc[3] = {0, 0, 0};
  for (int ky = 0; ky < 5; ++ky) {
    const int sy = reflect101(2 * y + ky - 2, H);
    int row[3] = {0, 0, 0};
    for (int kx = 0; kx < 5; ++kx) {
      const uint32_t p = src[(size_t)sy * W + reflect101(2 * x + kx - 2, W)];
      row[0] += kw[kx] * (int)(p & 255u);
      row[1] += kw[kx] * (int)((p >> 8) & 255u);
      row[2] += kw[kx] * (int)((p >> 16) & 255u);
    }
    acc[0] += kw[ky] * row[0]; acc[1] += kw[ky] * row[1]; acc[2] += kw[ky] * row[2];
  }
  dst[i] = (uint32_t)((acc[0] + 128) >> 8) | ((uint32_t)((acc[1] + 128) >> 8) << 8) | ((uint32_t)((acc[2] + 128) >> 8) << 16);
}

// pixel sources for the GRD kernels: packed u8 image of the ctx, or a CV_64FC3 RGB host volume
struct SrcU32 {
  const uint32_t *p;
  __device__ __forceinline__ void rgb(size_t i, double &r, double &g, double &b) const {
    const uint32_t q = p[i];
    b = (double)(q & 255u); g = (double)((q >> 8) & 255u); r = (double)((q >> 16) & 255u);
  }
};
struct SrcF64 {
  const double *p;
  __device__ __forceinline__ void rgb(size_t i, double &r, double &g, double &b) const { r = p[3 * i]; g = p[3 * i + 1]; b = p[3 * i + 2]; }
};

// grd_cc.cpp:70-73: convertTo(CV_32F); cvtColor(CV_RGB2GRAY): gray = R*0.299f + G*0.587f + B*0.114f in float
template <class Src>
__device__ __forceinline__ float gray_at(const Src &s, size_t i) {
  double r, g, b;
  s.rgb(i, r, g, b);
  float t = (float)r * 0.299f;
  t = t + (float)g * 0.587f;
  t = t + (float)b * 0.114f;
  return t;
}
// grd_cc.cpp:76-77: Sobel(gray, CV_64F, 1, 0, ksize=1) = gray[x+1]-gray[x-1] in double, REFLECT_101
template <class Src>
__global__ void k_gradient(Src s, int W, int H, double *__restrict__ grd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const size_t row = (size_t)y * W;
  grd[i] = (double)gray_at(s, row + reflect101(x + 1, W)) - (double)gray_at(s, row + reflect101(x - 1, W));
}

// GrdCC::buildCV / buildRightCV (cc/grd_cc.cpp:60-154) with myCostGrd (:4-35); one thread per cell,
// slabs d-major as Mat costVol[d].  Also reduces max over the volume (pre_cs_pc.cc:75-82).
//   left  view: other = right image at x-d, border branch when x-d < 0     (:88-100)
//   right view: other = left image at x+d, border branch when x+d >= wid   (:134-147)
template <class Src>
__global__ __launch_bounds__(256) void k_grd_volume(Src l, Src r, const double *__restrict__ lG, const double *__restrict__ rG,
                                                    int W, int H, int maxDis, int right_view, double *__restrict__ vol,
                                                    unsigned long long *max_key) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long cells = (long long)W * H * maxDis;
  double cost = -1.7976931348623157e308;
  if (i < cells) {
    const long long slab = (long long)W * H;
    const int d = (int)(i / slab);
    const long long o = i - (long long)d * slab;
    const int y = (int)(o / W), x = (int)(o - (long long)y * W);
    const int xo = right_view ? x + d : x - d;
    const bool inside = right_view ? (xo < W) : (xo >= 0);
    double c0, c1, c2, g0;  // own pixel
    double o0, o1, o2, og;  // other-view pixel, or BORDER_THRES
    if (right_view) { r.rgb((size_t)o, c0, c1, c2); g0 = rG[o]; } else { l.rgb((size_t)o, c0, c1, c2); g0 = lG[o]; }
    if (inside) {
      const size_t j = (size_t)y * W + xo;
      if (right_view) { l.rgb(j, o0, o1, o2); og = lG[j]; } else { r.rgb(j, o0, o1, o2); og = rG[j]; }
    } else {
      o0 = o1 = o2 = 3.0; og = 3.0;  // BORDER_THRES (grd_cc.h:6)
    }
    // myCostGrd(lC, rC, lG, rG): differences are always left - right; fabs makes the sign irrelevant.
    double clrDiff = 0;
    clrDiff += fabs(c0 - o0);
    clrDiff += fabs(c1 - o1);
    clrDiff += fabs(c2 - o2);
    clrDiff *= 0.3333333333;
    double grdDiff = fabs(g0 - og);
    clrDiff = clrDiff > 10.0 ? 10.0 : clrDiff;  // TAU_CLR
    grdDiff = grdDiff > 2.0 ? 2.0 : grdDiff;    // TAU_GRD
    cost = 0.1 * clrDiff + (1 - 0.1) * grdDiff; // ALPHA
    vol[i] = cost;
  }
  // block max -> one atomic per wave
  unsigned long long key = f64_key(cost);
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long other = __shfl_xor(key, off, kWave);
    key = other > key ? other : key;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(max_key, key);
}

// max over an uploaded (foreign CCMethod) volume
__global__ __launch_bounds__(256) void k_volume_max(const double *__restrict__ vol, long long cells, unsigned long long *max_key) {
  unsigned long long key = f64_key(-1.7976931348623157e308);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long k = f64_key(vol[i]);
    key = k > key ? k : key;
  }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long other = __shfl_xor(key, off, kWave);
    key = other > key ? other : key;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(max_key, key);
}
__global__ void k_keys_to_f64(const unsigned long long *keys, double *out, int n, double floor_val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double v = key_f64(keys[i]);
    out[i] = v > floor_val ? v : floor_val;  // the reference starts the max at -1.0 (pre_cs_pc.cc:75)
  }
}

}  // namespace cspm
