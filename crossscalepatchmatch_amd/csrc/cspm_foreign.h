// cspm_foreign.h -- CSPatchMatch driving a FOREIGN IPlaneCost (plane_cost/i_plane_cost.h:28-33: any object with a
// GetPlaneCost(x, y, plane, view), not one of this library's device costs).  The plugin contract is a per-call virtual function
// on the host, so the cost evaluations stay where the plugin lives; everything else of CSPatchMatch::PatchMatch
// (cs_patchmatch.cc:51-345) -- the plane field, the random streams, which candidates are tried where, and every accept
// decision -- is the same device code the fused paths use:
//   k_fpm_*_cand   one thread per candidate: where it is evaluated (view, x, y; x = -1: nowhere) and its plane (norm, param)
//   (host)         cost[i] = plugin->GetPlaneCost(x_i, y_i, Plane(norm_i, param_i), view_i)
//   k_fpm_*_commit the reference's accept rules on the returned costs
// Batches: InitRandomPlane and one halving step of PlaneRefinement = every pixel of both views; ViewPropagation = one view;
// the raster sweep = one anti-diagonal (its pixels depend only on earlier diagonals).
#pragma once
#include "cspm_rows.h"

namespace cspm {

struct FpmCand {
  int *xy;        // 2 ints per candidate: evaluation pixel (x = -1: no evaluation)
  int *view;      // target view of the evaluation
  double *plane;  // 6 doubles per candidate: norm, param
  double *cost;   // filled by the host between *_cand and *_commit
};
__device__ __forceinline__ void put_cand(const FpmCand &fc, long long i, int x, int y, int v, const RowPlane &p) {
  fc.xy[2 * i] = x; fc.xy[2 * i + 1] = y; fc.view[i] = v;
  double *o = fc.plane + 6 * i;
  o[0] = p.nx; o[1] = p.ny; o[2] = p.nz; o[3] = p.a; o[4] = p.b; o[5] = p.c;
}

// InitRandomPlane (cs_patchmatch.cc:115-148) / one halving step of PlaneRefinement (:292-345): candidate i = (view, y, x)
__global__ void k_fpm_point_cand(Pm pm, FpmCand fc, int refine, int iter, int step, double z_iter, double n_iter) {
  const long long n = (long long)pm.W * pm.H;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const int v = (int)(i / n);
  const long long r = i - (long long)v * n;
  const int y = (int)(r / pm.W), x = (int)(r - (long long)y * pm.W);
  const RowPlane p = refine ? refine_plane(pm, v, x, y, iter, step, z_iter, n_iter) : init_plane(pm, v, x, y);
  put_cand(fc, i, x, y, v, p);
}
__global__ void k_fpm_point_commit(Pm pm, FpmCand fc, int refine) {
  const long long n = (long long)pm.W * pm.H;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const int v = (int)(i / n);
  const long long r = i - (long long)v * n;
  const double cost = fc.cost[i];
  const Field &f = pm.f[v];
  if (!refine || cost < f.cost[r]) {  // :143-146 unconditional; :335-338 `<`
    const double *p = fc.plane + 6 * i;
    store_plane(f, r, p[0], p[1], p[2], p[3], p[4], p[5], cost);
  }
}

// ViewPropagation towards view v (:229-277): candidate i = source pixel (y, x) of view 1-v, evaluated at (cor_x, y) of view v
__global__ void k_fpm_view_cand(Pm pm, FpmCand fc, int v) {
  const long long n = (long long)pm.W * pm.H;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const ViewProposal q = view_proposal(pm, v, x, y);
  const bool inside = q.cor_x >= 0 && q.cor_x < pm.W;
  put_cand(fc, i, inside ? q.cor_x : -1, y, v, q.p);
}
__global__ void k_fpm_view_commit(Pm pm, FpmCand fc, ViewCand vc) {
  const long long n = (long long)pm.W * pm.H;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cx = fc.xy[2 * i];
  vc.cost[i] = cx >= 0 ? fc.cost[i] : __builtin_inf();
  vc.c[i] = fc.plane[6 * i + 5];
  vc.cx[i] = cx;  // k_view_resolve applies the serial loop's rule (smallest cost below the current one, earliest in traversal order)
}

// SpatialPropagation, anti-diagonal k of the raster sweep (:163-216): candidates 2j, 2j+1 = the x- and the y-predecessor's plane
// for the j-th pixel of the diagonal (both views stacked: cnt pixels of view 0, then cnt of view 1)
__device__ __forceinline__ bool fpm_diag_pixel(const Pm &pm, int k, int inc, long long j, int &v, int &x, int &y, int &xs, int &ys) {
  const int ys_lo = max(0, k - (pm.W - 1)), ys_hi = min(pm.H - 1, k);
  const int cnt = ys_hi - ys_lo + 1;
  if (j >= 2LL * cnt) return false;
  v = (int)(j / cnt);
  ys = ys_lo + (int)(j - (long long)v * cnt);
  xs = k - ys;
  x = inc > 0 ? xs : pm.W - 1 - xs;
  y = inc > 0 ? ys : pm.H - 1 - ys;
  return true;
}
__global__ void k_fpm_diag_cand(Pm pm, FpmCand fc, int k, int inc) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int v, x, y, xs, ys;
  if (!fpm_diag_pixel(pm, k, inc, j, v, x, y, xs, ys)) return;
  const Field &f = pm.f[v];
  const long long i = (long long)y * pm.W + x;
  const long long pred[2] = {i - inc, i - (long long)inc * pm.W};
  const bool have[2] = {xs > 0, ys > 0};
  for (int c = 0; c < 2; ++c) {
    RowPlane p{};
    if (have[c]) { const long long q = pred[c]; p = RowPlane{f.nx[q], f.ny[q], f.nz[q], f.a[q], f.b[q], f.c[q]}; }
    put_cand(fc, 2 * j + c, have[c] ? x : -1, y, v, p);
  }
}
__global__ void k_fpm_diag_commit(Pm pm, FpmCand fc, int k, int inc) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int v, x, y, xs, ys;
  if (!fpm_diag_pixel(pm, k, inc, j, v, x, y, xs, ys)) return;
  const Field &f = pm.f[v];
  const long long i = (long long)y * pm.W + x;
  double best = f.cost[i];
  int pick = -1;
  if (xs > 0 && fc.cost[2 * j] < best) { best = fc.cost[2 * j]; pick = 0; }          // x-predecessor first (:198-204)
  if (ys > 0 && fc.cost[2 * j + 1] < best) { best = fc.cost[2 * j + 1]; pick = 1; }  // then the y-predecessor against the updated minimum (:206-212)
  if (pick >= 0) {
    const double *p = fc.plane + 6 * (2 * j + pick);
    store_plane(f, i, p[0], p[1], p[2], p[3], p[4], p[5], best);
  }
}

}  // namespace cspm
