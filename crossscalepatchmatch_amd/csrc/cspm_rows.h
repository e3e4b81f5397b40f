// cspm_rows.h -- the ROW ENGINE: one LANE per pixel.  A wavefront owns a run of 64 x-adjacent pixels of one image row
// and every lane evaluates ITS OWN candidate plane at ITS OWN pixel, walking the support window serially
// (rows outer, columns inner) -- the loop nest of PreCSPC::GetPlaneCost (pre_cs_pc.cc:151-181) itself.
// Used by InitRandomPlane, PlaneRefinement and ViewPropagation (cs_patchmatch.cc:115-148, 229-277, 292-345):
// per-pixel independent work, > 95 % of all window taps of a run.
//
// Why lanes = pixels.  With lanes = taps (cspm_chain.h) every tap costs three gathers through the CU's L1 address path
// (own element, two of the other view) and that path, not arithmetic, bounds the kernel.  Here
//   * the own-view element of tap dx is ONE coalesced 768-byte row run for the whole wave (lane L reads column
//     x0+L+dx): scalar base + lane offset + an immediate, no address arithmetic;
//   * the other view's row is staged once per window row into a wave-private LDS strip ([cx_min-half-D, cx_max+half]
//     for the left view): every tap of every lane reads its two neighbouring cells from the strip with one address
//     computation, however incoherent the 64 planes are (random initialisation, early refinement steps);
//   * rows and columns outside the image cost nothing (rows are skipped by scalar control flow; only waves that touch
//     the image border carry the per-tap column mask);
//   * the running sums are registers of the lane: no cross-lane reduction, no per-level table set-up.
// Per tap and lane: ~30 VALU instructions (19 of them f64), 1 global load, 2 LDS strip reads, 3 LDS table reads.
#pragma once
#include "cspm_tap.h"

#pragma clang fp contract(off)

namespace cspm {

constexpr int kRowWaves = 8;                  // waves per workgroup (they share the two lookup tables)
constexpr int kRowBlock = kRowWaves * kWave;

// element of the LDS strip: what a tap needs of the other view, 16 bytes so that one ds_read_b128 fetches it
//   GRD: {g.lo, g.hi, pix, -}   census: {code0, code1, code2, pix}
// The strip of a wave holds `cap` elements; rows_shared_bytes() sizes the dynamic LDS of a launch.
struct RowShared {
  LutMem lut;
};
constexpr int kStripRegs = 6;                 // strip elements a lane carries from global memory to LDS: strips of <= 384 elements
__host__ __device__ inline int strip_capacity(int max_dis, int half) {
  const int want = kWave + 2 * half + max_dis + 2;  // 64 centres + window + disparity range (level 0 is the widest)
  return want <= kStripRegs * kWave ? want : kStripRegs * kWave;  // wider than that: the level reads the other view from global memory
}

// per-lane state of one level
struct RowLevel {
  int W, H, n, half, Dm1;
  int Wp, pad;
  bool has_valid;
  double maxc, wgt;
  const char *px, *opx;
  const double *vol;
  size_t slab;
};

// binary-counter stack for the row tree: pend[k] holds the sum of a completed block of 2^k rows
struct RowTree {
  double pend[6];
  // add row total R as row number `idx` (0-based, wave-uniform): blocks of equal size merge, earlier rows on the left
  __device__ __forceinline__ void push(int idx, double R) {
    double t = R;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (((idx >> k) & 1) == 0) { pend[k] = t; return; }
      t = pend[k] + t;
    }
    // idx = 63 (all ones): t is the complete 64-row tree; n <= 45, not reached
  }
  // the tree over 64 leaves whose rows >= count are +0.0: fold the pending blocks, small (late) ones first
  __device__ __forceinline__ double total(int count) const {
    double t = 0.0;
    bool have = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if ((count >> k) & 1) {
        t = have ? pend[k] + t : pend[k];
        have = true;
      }
    }
    return t;
  }
};

// One window row of one level for all 64 lanes.  EDGE: some lane's window leaves the image in x (per-tap mask).
// STAGED: the other view's cells come from the LDS strip (else straight from global memory: volumes, or a wave whose
// centres are too spread out for the strip).
//   own_row   : byte address of element (row qy, padded column 0) of the own view
//   lane_off  : byte offset of the lane's first tap (padded column pad + cx - half) * E
//   strip_adr : LDS byte address of the strip element that holds other-view column (cx - half) -/+ 0 for this lane, i.e.
//               tap dx at disparity f is at strip_adr + 16*dx + dirS*f  (dirS = -16 left view, +16 right view)
template <int SRC, bool EDGE, bool STAGED>
__device__ __forceinline__ double row_taps(const RowLevel &A, const Luts &lut, const char *own_row, const char *oth_row, int lane_off,
                                           const char *strip, int strip_adr, int dirS, int dirE, uint32_t Ip, double pa, double rowterm,
                                           double qx0_d, int e_lo, int e_span, int qy, int cx_lane) {
  constexpr int E = elem_size<SRC>();
  const int lutzero = kLutZero;
  double S[kRowMod];
#pragma unroll
  for (int j = 0; j < kRowMod; ++j) S[j] = 0.0;
  double qx_d = qx0_d;
  for (int g0 = 0; g0 < A.n; g0 += kRowMod) {
#pragma unroll
    for (int j = 0; j < kRowMod; ++j) {
      const int dx = g0 + j;
      if (dx < A.n) {  // wave-uniform; always true for the usual 35 = 5 x 7
        const uint4 P = ld_elem<SRC>(own_row, lane_off + dx * E);
        int sad = (int)__builtin_amdgcn_sad_u8(Ip, pix_of<SRC>(P), 0u);
        if (EDGE) sad = ((unsigned)(dx - e_lo) <= (unsigned)e_span) ? sad : lutzero;  // outside the image: weight entry 0.0
        const double wgt = lut.w[sad];                                                // :161-164
        const double q_disp = pa * qx_d + rowterm;                                    // :165
        const DispSplit d = split_disp(q_disp, A.Dm1, A.has_valid);
        double c0, c1;
        if (SRC == kSrcVolume) {
          int qx = cx_lane - A.half + dx;
          if (EDGE) qx = ((unsigned)(dx - e_lo) <= (unsigned)e_span) ? qx : cx_lane;
          const double *v = A.vol + (size_t)d.f * A.slab + (size_t)qy * A.W + qx;
          c0 = v[0];
          c1 = v[A.slab];
        } else if (STAGED) {
          const int adr = strip_adr + __mul24(dirS, d.f) + dx * 16;
          const uint4 o0 = *reinterpret_cast<const uint4 *>(strip + adr);
          const uint4 o1 = *reinterpret_cast<const uint4 *>(strip + adr + dirS);
          c0 = cell_of<SRC>(lut.a, P, o0);
          c1 = cell_of<SRC>(lut.a, P, o1);
        } else {
          const int of = lane_off + dx * E + __mul24(dirE, d.f);
          c0 = cell_of<SRC>(lut.a, P, ld_elem<SRC>(oth_row, of));
          c1 = cell_of<SRC>(lut.a, P, ld_elem<SRC>(oth_row, of + dirE));
        }
        S[j] += tap_value(d, c0, c1, A.maxc, wgt);
        qx_d += 1.0;  // exact: small integers
      }
    }
  }
  double R = S[0];
#pragma unroll
  for (int j = 1; j < kRowMod; ++j) R = R + S[j];
  return R;
}

// Per-wave description of the 64 evaluation centres of one pass (wave-uniform unless noted)
struct RowCtx {
  int view, y;       // all lanes evaluate in row y of `view`
  int lane;
  char *strip;       // this wave's LDS strip, `cap` elements of 16 bytes
  int cap;
};

// Aggregated plane cost of 64 candidates, one per lane, each at its own centre column `x` (per lane, inside the image)
// of row ctx.y.  (nx,ny,nz) = Plane::norm(), (pa,pb,pc) = Plane::param() -- per lane.  Returns the cost per lane;
// lanes whose candidate is proven not to beat `thresh` (per lane) may return +inf instead (checked at level ends, only
// when use_thresh; the wave leaves early once every lane is rejected).
template <bool CS, int SRC>
__device__ __forceinline__ double eval_rows(const Cost &cd, const Luts &lut, const RowCtx &ctx, int x, double nx, double ny, double nz,
                                            double pa, double pb, double pc, double thresh, bool use_thresh) {
  constexpr int E = elem_size<SRC>();
  const int lane = ctx.lane, view = ctx.view;
  double cost = 0.0;
  bool dead = false;
  double cur_disp = pa * (double)x + pb * (double)ctx.y + pc;  // pre_cs_pc.cc:139-140
  int cur_x = x, cur_y = ctx.y;
  // Plane(org_norm, Point3d(cur_x,cur_y,cur_disp)).param() (:144-149): a and b depend on the normal only, so
  // they are the same bits at every level; c is re-derived per level
  double denom = fmax(fabs(nz), kDoubleEps);
  if (nz < 0.0) denom = -denom;
  const double a = CS ? -nx / denom : pa, b = CS ? -ny / denom : pb;
  const int levels = CS ? cd.levels : 1;
  for (int s = 0; s < levels; ++s) {
    double c = pc;
    if (CS) {
      double dot = nx * (double)cur_x;
      dot += ny * (double)cur_y;
      dot += nz * cur_disp;
      c = dot / denom;
    }
    const Level &L = cd.lv[s];
    RowLevel A;
    A.W = L.W; A.H = L.H; A.n = cd.n; A.half = cd.half; A.Dm1 = L.D - 1;
    A.Wp = L.Wp; A.pad = L.pad;
    A.has_valid = L.D >= 2;
    A.maxc = cd.max_cost[view * CSPM_MAX_LEVELS + s];
    A.vol = L.vol[view];
    A.slab = (size_t)L.W * (size_t)L.H;
    if (SRC == kSrcCen) {
      A.px = reinterpret_cast<const char *>(L.pc[view]); A.opx = reinterpret_cast<const char *>(L.pc[1 - view]);
    } else {
      A.px = reinterpret_cast<const char *>(L.px[view]); A.opx = reinterpret_cast<const char *>(L.px[1 - view]);
    }
    const int cy = cur_y;  // wave-uniform
    const int cx = cur_x;  // per lane
    // the wave's span of centres: decides the strip window and whether any lane needs the column mask
    int cmin = cx, cmax = cx;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      cmin = min(cmin, __shfl_xor(cmin, off, kWave));
      cmax = max(cmax, __shfl_xor(cmax, off, kWave));
    }
    cmin = __builtin_amdgcn_readfirstlane(cmin);
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    const bool edge = (cmin - A.half < 0) | (cmax + A.half >= A.W);
    // strip window in padded columns: left view reads x-f-1 .. x-1, right view x+1 .. x+f+1  (f in [1, D-1])
    const int D = L.D;
    const int s_lo = view == 0 ? A.pad + cmin - A.half - D : A.pad + cmin - A.half;
    const int s_hi = view == 0 ? A.pad + cmax + A.half : A.pad + cmax + A.half + D + 1;
    const int s_len = s_hi - s_lo + 1;
    const bool staged = SRC != kSrcVolume && s_len <= ctx.cap;  // wave-uniform
    const int dirS = view == 0 ? -16 : 16, dirE = view == 0 ? -E : E;
    const int lane_off = (A.pad + cx - A.half) * E;
    const int strip_adr = (A.pad + cx - A.half - s_lo) * 16 + (view == 0 ? 0 : 0);
    const uint32_t Ip = SRC == kSrcCen ? L.pc[view][cy * L.Wp + L.pad + cx].pix : L.px[view][cy * L.Wp + L.pad + cx].pix;
    const double qx0_d = (double)(cx - A.half);
    const int e_lo = max(0, A.half - cx);                               // first window column inside the image
    const int e_span = min(A.n - 1, A.W - 1 - cx + A.half) - e_lo;      // last one, relative
    RowTree tree;
    const int dy_lo = max(0, A.half - cy), dy_hi = min(A.n - 1, A.H - 1 - cy + A.half);
    for (int dy = 0; dy < dy_lo; ++dy) tree.push(dy, 0.0);
    // The strip of window row dy+1 is fetched into registers while row dy is being evaluated and written to LDS afterwards:
    // the LDS queue of a wave is in order, so one strip suffices (reads of row dy precede the writes of row dy+1).
    uint4 pre[kStripRegs];
    const size_t row_stride = (size_t)A.Wp * E;
    const char *own_row = A.px + (size_t)(cy - A.half + dy_lo) * row_stride, *oth_row = A.opx + (size_t)(cy - A.half + dy_lo) * row_stride;
    if (staged) {
#pragma unroll
      for (int k = 0; k < kStripRegs; ++k)
        if (lane + k * kWave < s_len) pre[k] = ld_elem<SRC>(oth_row, (s_lo + lane + k * kWave) * E);
      wave_lds_fence();  // the previous level's strip reads are done
#pragma unroll
      for (int k = 0; k < kStripRegs; ++k)
        if (lane + k * kWave < s_len) *reinterpret_cast<uint4 *>(ctx.strip + (lane + k * kWave) * 16) = pre[k];
    }
    for (int dy = dy_lo; dy <= dy_hi; ++dy) {
      const int qy = cy - A.half + dy;
      const bool more = dy < dy_hi;
      if (staged) {
        wave_lds_fence();
        if (more) {
#pragma unroll
          for (int k = 0; k < kStripRegs; ++k)
            if (lane + k * kWave < s_len) pre[k] = ld_elem<SRC>(oth_row + row_stride, (s_lo + lane + k * kWave) * E);
        }
      }
      const double rowterm = b * (double)qy + c;  // q_disp_y, :155
      double R;
      if (staged) {
        R = edge ? row_taps<SRC, true, true>(A, lut, own_row, oth_row, lane_off, ctx.strip, strip_adr, dirS, dirE, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx)
                 : row_taps<SRC, false, true>(A, lut, own_row, oth_row, lane_off, ctx.strip, strip_adr, dirS, dirE, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
      } else {
        R = row_taps<SRC, true, false>(A, lut, own_row, oth_row, lane_off, ctx.strip, strip_adr, dirS, dirE, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
      }
      tree.push(dy, R);
      if (staged && more) {
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < kStripRegs; ++k)
          if (lane + k * kWave < s_len) *reinterpret_cast<uint4 *>(ctx.strip + (lane + k * kWave) * 16) = pre[k];
      }
      own_row += row_stride;
      oth_row += row_stride;
    }
    const double sc = tree.total(dy_hi + 1);
    if (CS) cost += sc * L.wgt;  // :182
    else cost = sc;
    if (use_thresh) {
      dead = dead | (cost >= thresh);
      if (__builtin_amdgcn_ballot_w64(!dead) == 0ull) break;  // every lane is rejected
    }
    cur_y /= 2;  // :183-185
    cur_x /= 2;
    cur_disp /= 2.0;
  }
  return dead ? __builtin_inf() : cost;
}

// pixel run of this wave: item = (view, y, 64-pixel segment), XCD-banded.  Returns false past the end.
struct RowItem {
  int v, y, x0;
};
__device__ __forceinline__ bool row_item(int W, int H, int views, RowItem &it) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int segs = (W + kWave - 1) / kWave;
  const long long e = xcd_block() * kRowWaves + wave;
  if (e >= (long long)views * H * segs) return false;
  const int per_view = H * segs;
  it.v = (int)(e / per_view);
  const int r = (int)(e - (long long)it.v * per_view);
  it.y = r / segs;
  it.x0 = (r - it.y * segs) * kWave;
  return true;
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::InitRandomPlane  (cs_patchmatch.cc:115-148)
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kRowBlock) void k_init(Cost cd, Pm pm, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LutMem &s_lut = *reinterpret_cast<LutMem *>(smem);
  const Luts lut = load_luts(cd, s_lut);
  RowItem it;
  if (!row_item(pm.W, pm.H, 2, it)) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const RowCtx ctx{it.v, it.y, lane, reinterpret_cast<char *>(smem + sizeof(LutMem)) + (size_t)wave * cap * 16, cap};
  const bool live = it.x0 + lane < pm.W;
  const int x = live ? it.x0 + lane : pm.W - 1;  // tail lanes shadow the last pixel
  const long long i = (long long)it.y * pm.W + x;
  const Rng rng(pm.seed, stream_id(0, 0, 0, it.v), pm.rng_row_shared ? (uint64_t)x : (uint64_t)i);
  const double rand_dis = rng.uniform(0, kDoubleEps, (double)pm.max_dis);  // :134-135
  // direction: uniform on the sphere by rejection from the unit ball (DESIGN.md "RNG"); :137-140
  double r0 = 0.0, r1 = 0.0, r2 = 1.0, len = 1.0;
  bool found = false;
  for (int t = 0; t < 32; ++t) {
    if (!found) {
      r0 = rng.uniform(1 + 3 * t, -1.0, 1.0);
      r1 = rng.uniform(2 + 3 * t, -1.0, 1.0);
      r2 = rng.uniform(3 + 3 * t, -1.0, 1.0);
      double s = r0 * r0;
      s += r1 * r1;
      s += r2 * r2;
      len = __dsqrt_rn(s);
      found = s <= 1.0 && s > 1e-12;
    }
    if (__builtin_amdgcn_ballot_w64(!found) == 0ull) break;
  }
  const double inv = 1. / fmax(len, kDoubleEps);
  const double nx = r0 * inv, ny = r1 * inv, nz = r2 * inv;
  double a, b, c;
  plane_param(nx, ny, nz, (double)x, (double)it.y, rand_dis, a, b, c);  // :141-142
  const double cost = eval_rows<CS, SRC>(cd, lut, ctx, x, nx, ny, nz, a, b, c, kDoubleMax, false);  // :143-144
  if (live) store_plane(pm.f[it.v], i, nx, ny, nz, a, b, c, cost);
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::PlaneRefinement  (cs_patchmatch.cc:292-345): ALL halving steps of one iteration in one launch.  A
// pixel's steps depend only on that pixel's own earlier steps, so the lane keeps its plane in registers across them.
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kRowBlock) void k_refine(Cost cd, Pm pm, int iter, int first_step, int nsteps, double z_iter, double n_iter, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LutMem &s_lut = *reinterpret_cast<LutMem *>(smem);
  const Luts lut = load_luts(cd, s_lut);
  RowItem it;
  if (!row_item(pm.W, pm.H, 2, it)) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const RowCtx ctx{it.v, it.y, lane, reinterpret_cast<char *>(smem + sizeof(LutMem)) + (size_t)wave * cap * 16, cap};
  const bool live = it.x0 + lane < pm.W;
  const int x = live ? it.x0 + lane : pm.W - 1;
  const long long i = (long long)it.y * pm.W + x;
  const Field &f = pm.f[it.v];
  double cnx = f.nx[i], cny = f.ny[i], cnz = f.nz[i], ca = f.a[i], cb = f.b[i], cc = f.c[i];
  double cur_min = f.cost[i];
  const bool use_thresh = pm.use_thresh != 0 && *cd.early_ok != 0;
  bool changed = false;
  for (int step = first_step; step < first_step + nsteps; ++step) {
    const Rng rng(pm.seed, stream_id(1, iter, step, it.v), pm.rng_row_shared ? (uint64_t)x : (uint64_t)i);
    const double disturb_z = ca * (double)x + cb * (double)it.y + cc;          // :317-319
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
    plane_param(nx, ny, nz, (double)x, (double)it.y, pz, a, b, c);             // :330
    const double cost = eval_rows<CS, SRC>(cd, lut, ctx, x, nx, ny, nz, a, b, c, cur_min, use_thresh);
    if (cost < cur_min) {                                                      // :335-338
      cnx = nx; cny = ny; cnz = nz; ca = a; cb = b; cc = c;
      cur_min = cost;
      changed = true;
    }
    z_iter /= 2.0;  // :342-343
    n_iter /= 2.0;
  }
  if (live && changed) store_plane(f, i, cnx, cny, cnz, ca, cb, cc, cur_min);
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::ViewPropagation  (cs_patchmatch.cc:229-277), target view v.
// Phase 1 (k_view_eval): every pixel (x,y) of the OTHER view proposes its plane to pixel (cor_x,y) of
// view v and evaluates it there -- independent, because the pass only reads the other view's planes
// and the candidates of a pass do not depend on each other.  Lane = source pixel; its evaluation centre
// cor_x = x +- disparity moves with the disparity field: coherent where the field is smooth (the strip
// still covers the wave), anything else falls back to global gathers for that level.
// Phase 2 (k_view_resolve): the serial loop keeps, per target pixel, the candidate with the smallest
// cost that is < the pixel's current cost, the earliest in traversal order among equal costs.  One
// workgroup per row (cor_x stays in row y) reproduces exactly that with LDS atomics.
// ------------------------------------------------------------------------------------------------
struct ViewCand {
  double *cost; // candidate cost, +inf = rejected / none
  double *c;    // candidate param c (a, b follow from the source normal)
  int *cx;      // target column
};

template <bool CS, int SRC>
__global__ __launch_bounds__(kRowBlock) void k_view_eval(Cost cd, Pm pm, int v, ViewCand vc, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LutMem &s_lut = *reinterpret_cast<LutMem *>(smem);
  const Luts lut = load_luts(cd, s_lut);
  RowItem it;
  if (!row_item(pm.W, pm.H, 1, it)) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const RowCtx ctx{v, it.y, lane, reinterpret_cast<char *>(smem + sizeof(LutMem)) + (size_t)wave * cap * 16, cap};
  const bool live = it.x0 + lane < pm.W;
  const int x = live ? it.x0 + lane : pm.W - 1;
  const int y = it.y;
  const long long i = (long long)y * pm.W + x;
  const Field &src = pm.f[1 - v];
  const Field &dst = pm.f[v];
  const double nx = src.nx[i], ny = src.ny[i], nz = src.nz[i];
  double disp = src.a[i] * (double)x + src.b[i] * (double)y + src.c[i];  // :245-246
  if (disp < 0.0) disp = 0.0;                                             // :247-252
  if (disp >= (double)pm.max_dis) disp = (double)pm.max_dis - 1.0;
  const int r = round2int(disp);
  const int cor_x = handle_border(v == 0 ? x + r : x - r, pm.W);          // :255-261
  const bool inside = cor_x >= 0 && cor_x < pm.W;
  const int ex = inside ? cor_x : x;
  double a, b, c;
  plane_param(nx, ny, nz, (double)ex, (double)y, disp, a, b, c);          // :263-265
  const double thr = dst.cost[(long long)y * pm.W + ex];
  const bool use_thresh = pm.use_thresh != 0 && *cd.early_ok != 0;
  double cost = eval_rows<CS, SRC>(cd, lut, ctx, ex, nx, ny, nz, a, b, c, use_thresh ? thr : kDoubleMax, use_thresh);  // :266-267
  if (!inside) cost = __builtin_inf();
  if (live) {
    vc.cost[i] = cost;
    vc.c[i] = c;
    vc.cx[i] = cor_x;
  }
}

}  // namespace cspm
