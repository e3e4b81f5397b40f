// cspm_kernels.h -- the HIP kernels of the PatchMatch-stereo hot path (gfx950, wave64).
//
// Mapping: ONE WAVEFRONT EVALUATES ONE CANDIDATE PLANE (k_init / k_refine / k_spatial_rb / k_view_eval /
// k_cost_batch).  The 64 lanes stride over the linearised (2*half+1)^2 support window (tap t -> lane
// t%64), so the plane parameters, the centre pixel and all branch decisions (early exit, accept/reject)
// are wave-uniform, and the image / cost-volume addresses of a wave are runs along image rows of ONE
// plane: coalesced however incoherent the plane field of neighbouring pixels is (random init, early
// refinement steps).  A workgroup is 4 waves = 4 consecutive candidates; blockIdx is remapped so that
// each XCD (block b runs on XCD b%8) walks one contiguous band of the image and keeps that band's image
// rows in its own L2.  The reference's raster sweep is run as anti-diagonals with 8 cooperating waves
// per pixel (k_spatial_diag), because a diagonal has only <= min(W,H) independent pixels.
//
// Cell costs come from one of two sources, selected at compile time:
//   SRC = kSrcGrd: GRD cell cost computed on the fly from the padded images + gradients (bit-identical
//                  to reading GrdCC's volume, cc/grd_cc.cpp:4-35,60-154); nothing but ~12 B/pixel per
//                  view and level is ever read, so the working set stays in L2 / Infinity Cache.
//   SRC = kSrcCen: census / Hamming cell cost computed on the fly from 80-bit codes (cc/cen_cc.cc:47-66).
//   SRC = kSrcVolume: cost volumes in HBM (any CCMethod plugin; what the reference's PreSSPC/PreCSPC do).
//
// Summation order ("SLOT256", mirrored by the oracle): tap t is accumulated in t order into slot
// t%256 (= accumulator (t/64)%4 of lane t%64); slots are reduced as (p0+p1)+(p2+p3) per lane, then an
// xor butterfly with offsets 1..32.
#pragma once
#include "cspm_device.h"

// minimum waves per SIMD the register allocator leaves room for in the sweep kernel (2nd __launch_bounds__ argument)
#ifndef CSPM_SWEEP_MINW
#define CSPM_SWEEP_MINW 4
#endif

#pragma clang fp contract(off)

namespace cspm {

// ------------------------------------------------------------------------------------------------
// IPlaneCost::GetPlaneCost  (PreSSPC: pre_ss_pc.cc:74-118, PreCSPC: pre_cs_pc.cc:133-188)
// ------------------------------------------------------------------------------------------------
struct Luts {
  const double *w;      // exp(-i/10), entry kLutZero = 0                 (pre_cs_pc.cc:111-114)
  const double *a;      // ALPHA*min(i*0.3333333333,TAU_CLR)               (grd_cc.cpp:8-18), fused path only
  double *tab;          // this wave's tables: tab[dx] = a*qx, tab[kTabSize+dy] = b*qy+c
  const uint32_t *dec;  // tap t -> dx | dy<<8 | (t>=T)<<31
};

// everything one level needs, wave-uniform
struct LevelArgs {
  int W, H, ox0, oy0;
  int row12, obase12, ocen12, dir12;  // BYTE offsets into the element arrays (12-byte PixG / 16-byte PixC): row stride,
                                      // window tap (0,0), centre, +- one element
  double Dd, maxc;
  const char *px, *opx;               // own / other view elements
  const double *vol;
  size_t slab;
  uint32_t Ip;
};

// LDS written by some lanes of a wave and read by others of the SAME wave: the LDS queue of a wave is
// in order, so only the compiler has to be kept from reordering.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
// One element with ONE global load and a 32-bit byte offset (saddr + voffset addressing, no 64-bit address math):
// PixG = dwordx3 {g.lo, g.hi, pix}, PixC = dwordx4 {code0, code1, code2, pix}.  Held as 4 dwords either way.
template <int SRC>
__device__ __forceinline__ uint4 ld_elem(const char *base, int byte_off) {
  if (SRC == kSrcCen) return *reinterpret_cast<const uint4 *>(base + (size_t)(unsigned)byte_off);
  const u32x3 v = *reinterpret_cast<const u32x3_a4 *>(base + (size_t)(unsigned)byte_off);
  return uint4{v.x, v.y, v.z, 0u};
}
template <int SRC> constexpr int elem_size() { return SRC == kSrcCen ? 16 : 12; }
template <int SRC>
__device__ __forceinline__ uint32_t pix_of(const uint4 &v) { return SRC == kSrcCen ? v.w : v.z; }
__device__ __forceinline__ double g_of(const uint4 &v) { return __hiloint2double((int)v.y, (int)v.x); }

__device__ __forceinline__ void fill_tab(const Cost &cd, double *tab, int ox0, int oy0, double a, double b, double c, int lane) {
  wave_lds_fence();  // earlier reads of this table are done
  for (int l = lane; l < cd.n; l += kWave) {
    tab[l] = a * (double)(ox0 + l);
    tab[kTabSize + l] = b * (double)(oy0 + l) + c;
  }
  wave_lds_fence();
}

// Prepare one level for this wave: uniform arguments plus the two per-wave tables
//   tab[dx]          = plane_a * q_x            (the product of pre_cs_pc.cc:165)
//   tab[kTabSize+dy] = plane_b * q_y + plane_c  (q_disp_y, pre_cs_pc.cc:155)
// so that a tap's q_disp is one add of two LDS reads instead of two int->f64 converts, two multiplies
// and two adds -- bit-identical, each table entry is rounded exactly like the expression it replaces.
template <int SRC>
__device__ __forceinline__ LevelArgs make_level(const Cost &cd, const Luts &lut, int s, int view, int cx, int cy, double a,
                                                double b, double c, int lane) {
  const Level &L = cd.lv[s];
  constexpr int E = elem_size<SRC>();
  LevelArgs A;
  A.W = L.W; A.H = L.H;
  A.ox0 = cx - cd.half; A.oy0 = cy - cd.half;
  A.row12 = L.Wp * E;
  A.obase12 = (A.oy0 * L.Wp + L.pad + A.ox0) * E;  // window tap (0,0); may be negative, used masked
  A.ocen12 = (cy * L.Wp + L.pad + cx) * E;
  A.dir12 = view == 0 ? -E : E;  // left view looks at x-d in the right image, right view at x+d in the left
  A.Dd = (double)L.D;
  A.maxc = cd.max_cost[view * CSPM_MAX_LEVELS + s];
  if (SRC == kSrcCen) {
    A.px = reinterpret_cast<const char *>(L.pc[view]); A.opx = reinterpret_cast<const char *>(L.pc[1 - view]);
    A.Ip = L.pc[view][cy * L.Wp + L.pad + cx].pix;
  } else {
    A.px = reinterpret_cast<const char *>(L.px[view]); A.opx = reinterpret_cast<const char *>(L.px[1 - view]);
    A.Ip = L.px[view][cy * L.Wp + L.pad + cx].pix;
  }
  A.vol = L.vol[view];
  A.slab = (size_t)L.W * (size_t)L.H;
  fill_tab(cd, lut.tab, A.ox0, A.oy0, a, b, c, lane);
  return A;
}

// myCostGrd (cc/grd_cc.cpp:4-35) on one (own pixel, other pixel) pair; the border variant is the same
// arithmetic on the pad cells.  |dR|+|dG|+|dB| is an exact small integer, so ALPHA*min(sum*0.3333333333,
// TAU_CLR) is a table of the SAD; min(.,TAU_GRD) on finite values is v_min_f64.
__device__ __forceinline__ double grd_cell(const Luts &lut, uint32_t Iq, double Gq, const uint4 &o) {
  const int sad = (int)__builtin_amdgcn_sad_u8(Iq, pix_of<kSrcGrd>(o), 0u);
  const double grdDiff = __builtin_fmin(fabs(Gq - g_of(o)), 2.0);  // TAU_GRD
  return lut.a[sad] + (1 - 0.1) * grdDiff;                         // ALPHA*clrDiff + (1-ALPHA)*grdDiff
}
// CenCC cell (cc/cen_cc.cc:54-62): Hamming distance of the two 80-bit codes, CENCUS_BIT = 80 when the other view's
// pixel is outside the image (pad cells carry bit 31 in `pix`)
__device__ __forceinline__ double cen_cell(const uint4 &q, const uint4 &o) {
  const int eighty = 80;
  const int ham = __popc(q.x ^ o.x) + __popc(q.y ^ o.y) + __popc(q.z ^ o.z);
  const int cnt = ((int)o.w < 0) ? eighty : ham;
  return (double)cnt;
}

// v_cvt_i32_f64 saturates and maps NaN to 0; written as asm because (int)double is undefined out of range.
__device__ __forceinline__ int cvt_i32_sat(double x) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// One window tap t (pre_cs_pc.cc:157-179) in two parts, so that several candidate planes evaluated at the SAME
// pixel share the plane-independent half (tap decode, bounds, own pixel, guide weight).
// NB: every `c ? x : y` has plain locals on both sides.  clang emits a real branch for a conditional
// operator with a member access in an arm, and LLVM then sinks all loads of the tap into that branch,
// which serialises the taps of a round behind s_waitcnt vmcnt(0).
struct TapOwn {
  int o, dx, dy;
  bool ok;
  uint4 P;     // own element: gradient + colour (GRD / volume) or census code + colour
  double wgt;  // lookup_exp_[|dB|+|dG|+|dR|] (:161-164); 0 for taps outside the window / image
};

template <int SRC>
__device__ __forceinline__ TapOwn tap_own(const LevelArgs &A, const Luts &lut, int t) {
  const int ocen = A.ocen12, lutzero = kLutZero;
  TapOwn w;
  const int dec = (int)lut.dec[t];
  w.dx = dec & 255;
  w.dy = (dec >> 8) & 255;
  w.ok = (dec >= 0) & ((unsigned)(A.oy0 + w.dy) < (unsigned)A.H) & ((unsigned)(A.ox0 + w.dx) < (unsigned)A.W);
  // byte offset of the tap's element: two 24-bit multiply-adds (full rate; v_mul_lo_u32 / v_mad_u64_u32 are not)
  const int o0 = __mul24(w.dx, elem_size<SRC>()) + (__mul24(w.dy, A.row12) + A.obase12);
  w.o = w.ok ? o0 : ocen;                               // masked taps read the centre pixel ...
  w.P = ld_elem<SRC>(A.px, w.o);
  const int sum0 = (int)__builtin_amdgcn_sad_u8(A.Ip, pix_of<SRC>(w.P), 0u);
  const int sum = w.ok ? sum0 : lutzero;                // ... with weight entry kLutZero = 0.0, so they add +0.0
  w.wgt = lut.w[sum];
  return w;
}

// plane-dependent half: tab = the candidate's tables (tab[dx] = a*qx, tab[kTabSize+dy] = b*qy+c)
template <int SRC>
__device__ __forceinline__ double tap_plane(const Cost &cd, const LevelArgs &A, const Luts &lut, const double *tab, const TapOwn &w) {
  const int one = 1;
  const double maxc = A.maxc;
  const double q_disp = tab[w.dx] + tab[kTabSize + w.dy];        // :155,165 (masked taps: any finite or NaN value)
  // static_cast<int>(q_disp) in [1, D-1]  <=>  1.0 <= q_disp < D; NaN / out of int range -> the
  // "impossible disparity" branch (:166-169), as x86 cvttsd2si (INT_MIN) takes it.
  const bool valid = (q_disp >= 1.0) & (q_disp < A.Dd);
  const int f0 = cvt_i32_sat(q_disp);
  const int f = valid ? f0 : one;
  const double floor_wgt = (double)(f + 1) - q_disp;             // :171-172
  double c0, c1;
  if (SRC == kSrcGrd) {
    const double Gq = g_of(w.P);
    const int of = w.o + __mul24(A.dir12, f);
    c0 = grd_cell(lut, pix_of<SRC>(w.P), Gq, ld_elem<SRC>(A.opx, of));
    c1 = grd_cell(lut, pix_of<SRC>(w.P), Gq, ld_elem<SRC>(A.opx, of + A.dir12));
  } else if (SRC == kSrcCen) {
    const int of = w.o + __mul24(A.dir12, f);
    c0 = cen_cell(w.P, ld_elem<SRC>(A.opx, of));
    c1 = cen_cell(w.P, ld_elem<SRC>(A.opx, of + A.dir12));
  } else {
    const int hh = cd.half;
    const int dyc = w.ok ? w.dy : hh, dxc = w.ok ? w.dx : hh;
    const double *p = A.vol + (size_t)f * A.slab + (size_t)(A.oy0 + dyc) * A.W + (A.ox0 + dxc);
    c0 = p[0];
    c1 = p[A.slab];
  }
  double tmp = floor_wgt * c0 + (1 - floor_wgt) * c1;            // :173-175
  tmp = valid ? tmp : maxc;                                      // :169
  return w.wgt * tmp;                                            // :176
}

template <int SRC>
__device__ __forceinline__ double tap_term(const Cost &cd, const LevelArgs &A, const Luts &lut, int t) {
  const TapOwn w = tap_own<SRC>(A, lut, t);
  return tap_plane<SRC>(cd, A, lut, lut.tab, w);
}

// Cheap wave-wide LOWER-BOUND sum for the early-exit test: f32 DPP reduction (6 VALU instructions, no
// LDS).  The per-lane f64 partial is rounded toward zero to f32 and the f32 sum is scaled by (1 - 2^-16)
// (64 round-to-nearest additions inflate it by < 64*2^-24), so the returned value never exceeds the
// exact SLOT256 sum of the same partials: a candidate rejected on it would have been rejected anyway.
__device__ __forceinline__ float wave_lower_bound(double part) {
  float v = __double2float_rz(part);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));  // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31
  const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return tot * 0.99998474f;  // 1 - 2^-16
}

// One level, one wave: returns the level sum (identical in all lanes) or -1.0 once
// base + partial*mul >= thresh is proven (all terms are >= 0: monotone, so the candidate is rejected).
// The proof uses the cheap lower bound after every round of 256 taps and the exact sum at the level end.
template <int SRC>
__device__ __forceinline__ double level_cost(const Cost &cd, const LevelArgs &A, const Luts &lut, double base, double mul,
                                             double thresh, bool use_thresh, int lane) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int rounds = cd.rounds;
  // exit when base + S*mul >= thresh, S >= lb: implied by lb > (thresh-base)/mul * (1+1e-6); the margin
  // covers the roundings of this expression and of the float conversion (mul == 0 gives inf/NaN: no exit)
  const float need = use_thresh ? (float)(((thresh - base) / mul) * 1.000001) : 0.0f;
  for (int i = 0; i < rounds; ++i) {
    const int t = i * 256 + lane;
    const double t0 = tap_term<SRC>(cd, A, lut, t);
    const double t1 = tap_term<SRC>(cd, A, lut, t + 64);
    const double t2 = tap_term<SRC>(cd, A, lut, t + 128);
    const double t3 = tap_term<SRC>(cd, A, lut, t + 192);
    a0 += t0; a1 += t1; a2 += t2; a3 += t3;
    if (i == rounds - 1) break;
    if (use_thresh && wave_lower_bound((a0 + a1) + (a2 + a3)) > need) return -1.0;
  }
  const double tot = wave_sum((a0 + a1) + (a2 + a3));
  if (use_thresh && base + tot * mul >= thresh) return -1.0;
  return tot;
}

// Aggregated plane cost at (x,y); +inf when the candidate is proven not to beat `thresh`.
// (nx,ny,nz) = Plane::norm(), (pa,pb,pc) = Plane::param().
template <bool CS, int SRC>
__device__ __forceinline__ double eval_plane(const Cost &cd, const Luts &lut, int view, int x, int y, double nx, double ny,
                                             double nz, double pa, double pb, double pc, double thresh, bool use_thresh,
                                             int lane) {
  if (!CS) {
    const LevelArgs A = make_level<SRC>(cd, lut, 0, view, x, y, pa, pb, pc, lane);
    const double r = level_cost<SRC>(cd, A, lut, 0.0, 1.0, thresh, use_thresh, lane);
    return r < 0.0 ? __builtin_inf() : r;
  }
  double cost = 0.0;
  double cur_disp = pa * (double)x + pb * (double)y + pc;  // pre_cs_pc.cc:139-140
  int cur_x = x, cur_y = y;
  // Plane(org_norm, Point3d(cur_x,cur_y,cur_disp)).param() (:144-149): a and b depend on the normal only, so
  // they are the same bits at every level; c is re-derived per level
  double denom = fmax(fabs(nz), kDoubleEps);
  if (nz < 0.0) denom = -denom;
  const double a = -nx / denom, b = -ny / denom;
  for (int s = 0; s < cd.levels; ++s) {
    double dot = nx * (double)cur_x;
    dot += ny * (double)cur_y;
    dot += nz * cur_disp;
    const double c = dot / denom;
    const double wgt = cd.lv[s].wgt;
    const LevelArgs A = make_level<SRC>(cd, lut, s, view, cur_x, cur_y, a, b, c, lane);
    const double sc = level_cost<SRC>(cd, A, lut, cost, wgt, thresh, use_thresh, lane);
    if (sc < 0.0) return __builtin_inf();
    cost += sc * wgt;  // :182
    cur_y /= 2;        // :183-185
    cur_x /= 2;
    cur_disp /= 2.0;
  }
  return cost;
}

template <int WAVES, int TABS = 1>
struct LutMem {
  double w[kLutSize];
  double a[kLutSize];
  double tab[WAVES][TABS * 2 * kTabSize];
  uint32_t dec[kMaxRounds * 256];
};
template <int WAVES, int TABS>
__device__ __forceinline__ Luts load_luts(const Cost &cd, LutMem<WAVES, TABS> &m) {
  for (int i = threadIdx.x; i < kLutSize; i += blockDim.x) {
    m.w[i] = i == kLutZero ? 0.0 : cd.lut[i];
    m.a[i] = cd.lut_a[i];
  }
  for (int i = threadIdx.x; i < WAVES * TABS * 2 * kTabSize; i += blockDim.x) (&m.tab[0][0])[i] = 0.0;
  for (int i = threadIdx.x; i < cd.rounds * 256; i += blockDim.x) m.dec[i] = cd.dec[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  return Luts{m.w, m.a, m.tab[wave], m.dec};
}

// Work item (candidate) index of this wave.  Blocks are dealt round-robin to the 8 XCDs; give XCD k
// the k-th contiguous eighth of the index space.
__device__ __forceinline__ long long xcd_block() {
  const long long per = (long long)gridDim.x / 8;  // gridDim.x is a multiple of 8
  return (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
}
__device__ __forceinline__ long long wave_item(long long n_items) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long e = xcd_block() * (kEvalBlock / kWave) + wave;
  return e < n_items ? e : -1;
}

__device__ __forceinline__ void store_plane(const Field &f, long long i, double nx, double ny, double nz, double a, double b,
                                            double c, double cost) {
  f.nx[i] = nx; f.ny[i] = ny; f.nz[i] = nz;
  f.a[i] = a; f.b[i] = b; f.c[i] = c;
  f.cost[i] = cost;
}

// ------------------------------------------------------------------------------------------------
// cspm_plane_cost_batch: batched GetPlaneCost on explicit (x,y,plane) tuples -- the parity hook.
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_cost_batch(Cost cd, int view, int n, const int *__restrict__ xy,
                                                           const double *__restrict__ np, double *__restrict__ out) {
  __shared__ LutMem<kEvalBlock / kWave> s_lut;
  const Luts lut = load_luts(cd, s_lut);
  const long long e = wave_item(n);
  if (e < 0) return;
  const int lane = threadIdx.x & 63;
  const int x = xy[2 * e], y = xy[2 * e + 1];
  const double *p = np + 6 * e;
  const double c = eval_plane<CS, SRC>(cd, lut, view, x, y, p[0], p[1], p[2], p[3], p[4], p[5], kDoubleMax, false, lane);
  if (lane == 0) out[e] = c;
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::InitRandomPlane  (cs_patchmatch.cc:115-148)
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_init(Cost cd, Pm pm) {
  __shared__ LutMem<kEvalBlock / kWave> s_lut;
  const Luts lut = load_luts(cd, s_lut);
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
  const double cost = eval_plane<CS, SRC>(cd, lut, v, x, y, nx, ny, nz, a, b, c, kDoubleMax, false, lane);  // :143-144
  if (lane == 0) store_plane(pm.f[v], i, nx, ny, nz, a, b, c, cost);
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::PlaneRefinement, one halving step  (cs_patchmatch.cc:303-344)
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_refine(Cost cd, Pm pm, int iter, int step, double z_iter, double n_iter) {
  __shared__ LutMem<kEvalBlock / kWave> s_lut;
  const Luts lut = load_luts(cd, s_lut);
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
  const double cost = eval_plane<CS, SRC>(cd, lut, v, x, y, nx, ny, nz, a, b, c, cur_min, (pm.use_thresh != 0 && *cd.early_ok != 0), lane);
  if (cost < cur_min && lane == 0) store_plane(f, i, nx, ny, nz, a, b, c, cost);  // :335-338
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::SpatialPropagation, red-black half-step (fast, lower-quality schedule).
// ------------------------------------------------------------------------------------------------
struct Cand { double nx, ny, nz, a, b, c; };

template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_spatial_rb(Cost cd, Pm pm, int colour, int inc, int nb) {
  __shared__ LutMem<kEvalBlock / kWave> s_lut;
  const Luts lut = load_luts(cd, s_lut);
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
    const long long j = (long long)nys[k] * pm.W + nxs[k];
    const Cand cand{f.nx[j], f.ny[j], f.nz[j], f.a[j], f.b[j], f.c[j]};
    const double cost = eval_plane<CS, SRC>(cd, lut, v, x, y, cand.nx, cand.ny, cand.nz, cand.a, cand.b, cand.c, best_cost,
                                              (pm.use_thresh != 0 && *cd.early_ok != 0), lane);
    if (cost < best_cost) { best_cost = cost; best = cand; changed = true; }
  }
  if (changed && lane == 0) store_plane(f, i, best.nx, best.ny, best.nz, best.a, best.b, best.c, best_cost);
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::SpatialPropagation in the reference's order (cs_patchmatch.cc:163-216): the in-place
// raster sweep makes pixel (x,y) depend on (x-inc,y) and (x,y-inc) only, so all pixels of one
// anti-diagonal are independent.  One launch per diagonal k (sweep coordinates xs+ys == k, image
// x = inc>0 ? xs : W-1-xs); one 8-wave workgroup per pixel: waves 0-3 evaluate the x-predecessor's
// plane, waves 4-7 the y-predecessor's, each wave one of the four SLOT256 accumulator blocks of every
// level.  A pixel tries the x-predecessor first, then the y-predecessor against the updated minimum
// (:198-212); the first sweep row has only the former (:178-186), the first column only the latter
// (:189-195).  No early exit: both candidate costs are needed in full when accepted.
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kDiagBlock) void k_spatial_diag(Cost cd, Pm pm, int k, int inc) {
  __shared__ LutMem<kDiagBlock / kWave> s_lut;
  __shared__ double s_part[2][CSPM_MAX_LEVELS][4][kWave];
  __shared__ double s_cost[2];
  const Luts lut = load_luts(cd, s_lut);
  const int ys_lo = max(0, k - (pm.W - 1)), ys_hi = min(pm.H - 1, k);
  const int cnt = ys_hi - ys_lo + 1;
  const int b = (int)blockIdx.x;
  const int v = b / cnt;  // grid = 2*cnt
  const int ys = ys_lo + (b - v * cnt), xs = k - ys;
  const int x = inc > 0 ? xs : pm.W - 1 - xs, y = inc > 0 ? ys : pm.H - 1 - ys;
  const Field &f = pm.f[v];
  const long long i = (long long)y * pm.W + x;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cand = wave >> 2, blk = wave & 3;
  const bool have = cand == 0 ? (xs > 0) : (ys > 0);
  const long long j = cand == 0 ? i - inc : i - (long long)inc * pm.W;
  const int levels = CS ? cd.levels : 1;
  Cand c{};
  if (have) {
    c = Cand{f.nx[j], f.ny[j], f.nz[j], f.a[j], f.b[j], f.c[j]};
    double cur_disp = c.a * (double)x + c.b * (double)y + c.c;  // pre_cs_pc.cc:139-140
    int cur_x = x, cur_y = y;
    for (int s = 0; s < levels; ++s) {
      double pa = c.a, pb = c.b, pc = c.c;
      if (CS) plane_param(c.nx, c.ny, c.nz, (double)cur_x, (double)cur_y, cur_disp, pa, pb, pc);
      const LevelArgs A = make_level<SRC>(cd, lut, s, v, cur_x, cur_y, pa, pb, pc, lane);
      double acc = 0.0;
      const int rounds = cd.rounds;
      int i = 0;
      for (; i + 5 <= rounds; i += 5) {  // 5 independent taps per trip: their loads are issued together
        const int t = i * 256 + blk * 64 + lane;
        const double t0 = tap_term<SRC>(cd, A, lut, t);
        const double t1 = tap_term<SRC>(cd, A, lut, t + 256);
        const double t2 = tap_term<SRC>(cd, A, lut, t + 512);
        const double t3 = tap_term<SRC>(cd, A, lut, t + 768);
        const double t4 = tap_term<SRC>(cd, A, lut, t + 1024);
        acc += t0; acc += t1; acc += t2; acc += t3; acc += t4;
      }
      for (; i < rounds; ++i) acc += tap_term<SRC>(cd, A, lut, i * 256 + blk * 64 + lane);
      s_part[cand][s][blk][lane] = acc;
      cur_y /= 2; cur_x /= 2; cur_disp /= 2.0;
    }
  }
  __syncthreads();
  if (blk == 0 && have) {
    double cost = 0.0;
    for (int s = 0; s < levels; ++s) {
      const double sc = wave_sum((s_part[cand][s][0][lane] + s_part[cand][s][1][lane]) +
                                 (s_part[cand][s][2][lane] + s_part[cand][s][3][lane]));
      if (CS) cost += sc * cd.lv[s].wgt;  // :182
      else cost = sc;
    }
    if (lane == 0) s_cost[cand] = cost;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double best_cost = f.cost[i];
    int pick = -1;
    if (xs > 0 && s_cost[0] < best_cost) { best_cost = s_cost[0]; pick = 0; }
    if (ys > 0 && s_cost[1] < best_cost) { best_cost = s_cost[1]; pick = 1; }
    if (pick >= 0) {
      const long long q = pick == 0 ? i - inc : i - (long long)inc * pm.W;
      store_plane(f, i, f.nx[q], f.ny[q], f.nz[q], f.a[q], f.b[q], f.c[q], best_cost);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same raster sweep as ONE persistent launch (default): workgroups pull pixels in diagonal-major
// order from a device-wide counter and wait, per pixel, for the "done" flags of its two predecessors
// instead of for a kernel boundary.  Dataflow instead of 1615 launches per sweep: a pixel starts as
// soon as ITS predecessors are final, diagonals overlap, and the ~15 us fixed cost per launch is gone.
//
// Inter-workgroup protocol (MI355X: 8 XCDs with private, mutually non-coherent L2s; per-CU L1 never
// refreshed by other CUs' stores): every word another workgroup may read -- the 7 doubles of a plane
// and the done flag -- is written with 8-byte / 4-byte AGENT-scope atomic stores (write-through) and
// read with agent-scope atomic loads (L1 bypass), both sides; the producer drains its stores
// (s_waitcnt vmcnt(0)) before it stores the flag.  No fences, no reliance on placement or dispatch
// order.  Deadlock freedom: pixels are claimed in an order in which predecessors come first, so every
// flag a workgroup waits for belongs to a pixel already claimed by a running workgroup.  Every spin is
// bounded (wall clock); a timeout raises ctrl[1] and all workgroups drain.
// ------------------------------------------------------------------------------------------------
struct Sweep {
  unsigned int *ctrl;         // [0] next item, [1] error
  unsigned int *done[2];      // per view, per pixel: epoch of the last sweep that finalised the pixel
  const unsigned int *start;  // start[k] = items (both views) on diagonals < k; W+H entries
  unsigned int epoch, total;
  long long *trace;  // debug (-DCSPM_SWEEP_TRACE): 8 wall-clock stamps per item
};
#ifdef CSPM_SWEEP_TRACE
#define SWEEP_STAMP(slot) do { if (threadIdx.x == 0 && sw.trace) sw.trace[(size_t)item * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define SWEEP_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ double ld_agent(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wait_done(const unsigned int *flag, unsigned int epoch, unsigned int *err) {
  const long long t0 = wall_clock64();
  for (unsigned spins = 1;; ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 255u) == 0u) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (wall_clock64() - t0 > 300000000LL) {  // 3 s of the 100 MHz constant clock
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// Both candidates of a sweep pixel (the planes of its x- and y-predecessor) are evaluated at the same pixel,
// i.e. over the same window: one pass computes the plane-independent half of every tap once and the
// plane-dependent half twice.  NC = number of candidates present (2 except on the first sweep row / column).
// Returns exact SLOT256 sums, identical in all lanes.
template <int SRC, int NC>
__device__ __forceinline__ void level_cost_pair(const Cost &cd, const LevelArgs &A, const Luts &lut, int t_first, int t_step,
                                                int t_end, const double *tab0, const double *tab1, double acc0[4], double acc1[4]) {
  if (t_end == 1280 && t_step == 256) {
    // the usual 35x35 window: 5 rounds, fully unrolled so that the loads of later rounds are issued while earlier
    // rounds compute -- a sweep pixel is latency-bound, and its latency is the length of the dependency chain
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const TapOwn w = tap_own<SRC>(A, lut, t_first + i * 256 + 64 * u);
        acc0[u] += tap_plane<SRC>(cd, A, lut, tab0, w);
        if (NC == 2) acc1[u] += tap_plane<SRC>(cd, A, lut, tab1, w);
      }
    }
    return;
  }
  for (int t = t_first; t < t_end; t += t_step) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const TapOwn w = tap_own<SRC>(A, lut, t + 64 * u);
      acc0[u] += tap_plane<SRC>(cd, A, lut, tab0, w);
      if (NC == 2) acc1[u] += tap_plane<SRC>(cd, A, lut, tab1, w);
    }
  }
}

// Work split inside a sweep workgroup: cross-scale -> one wave per pyramid level (`levels` waves); single-scale
// -> 4 waves, one per SLOT256 accumulator block.  Every wave handles both candidates.
constexpr int kSweepMaxWaves = 8;
// the second candidate's tables live in the gaps of the first one's (entries 48..95 and kTabSize+48..): a window has
// at most 45 columns / rows, so one 2 KB table block per wave serves both and a fourth workgroup fits in the CU's LDS
constexpr int kTab1 = 48;

template <bool CS, int SRC>
__global__ __launch_bounds__(kSweepMaxWaves * kWave, CSPM_SWEEP_MINW) void k_spatial_sweep(Cost cd, Pm pm, Sweep sw, int inc) {
  __shared__ LutMem<kSweepMaxWaves, 1> s_lut;
  __shared__ double s_part[2][4][kWave];        // single-scale: per-lane partials of the 4 accumulator blocks
  __shared__ double s_lvl[2][CSPM_MAX_LEVELS];  // cross-scale: exact level sums
  __shared__ double s_plane[2][6];
  __shared__ unsigned int s_item;
  __shared__ int s_ok;
  const Luts lut = load_luts(cd, s_lut);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const double *tab0 = lut.tab, *tab1 = lut.tab + kTab1;
  const int ndiag = pm.W + pm.H - 1;
  int k = 0;
  unsigned int next_item = 0;
  if (threadIdx.x == 0) next_item = __hip_atomic_fetch_add(&sw.ctrl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    if (threadIdx.x == 0) {
      s_item = next_item;
      s_ok = 1;
      // claim the following item now: the atomic's latency hides behind this item's work.  Claims of a
      // workgroup stay increasing, which is all the deadlock argument needs.
      if (next_item < sw.total) next_item = __hip_atomic_fetch_add(&sw.ctrl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned int item = s_item;
    if (item >= sw.total) return;
    SWEEP_STAMP(0);
    while (k + 1 < ndiag && item >= sw.start[k + 1]) ++k;  // items of one workgroup only increase
    const int ys_lo = max(0, k - (pm.W - 1)), ys_hi = min(pm.H - 1, k);
    const int cnt = ys_hi - ys_lo + 1;
    const int r = (int)(item - sw.start[k]);
    const int v = r / cnt;
    const int ys = ys_lo + (r - v * cnt), xs = k - ys;
    const int x = inc > 0 ? xs : pm.W - 1 - xs, y = inc > 0 ? ys : pm.H - 1 - ys;
    const Field &f = pm.f[v];
    const long long i = (long long)y * pm.W + x;
    const long long jx = i - inc, jy = i - (long long)inc * pm.W;
    const bool have0 = xs > 0, have1 = ys > 0;
    SWEEP_STAMP(1);
    // 1. wait for the predecessors: lanes 0 and 1 of wave 0 poll one flag each
    if (wave == 0 && lane < 2) {
      const bool need = lane == 0 ? have0 : have1;
      if (need && !wait_done(sw.done[v] + (lane == 0 ? jx : jy), sw.epoch, &sw.ctrl[1])) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) return;
    asm volatile("" ::: "memory");
    SWEEP_STAMP(2);
    // 2. both candidate costs in one pass over the window
    if (have0 || have1) {
      // with one candidate missing, both slots hold the existing one (the duplicate is computed once, NC = 1)
      const long long j0 = have0 ? jx : jy, j1 = have1 ? jy : jx;
      const Cand c0{ld_agent(f.nx + j0), ld_agent(f.ny + j0), ld_agent(f.nz + j0), ld_agent(f.a + j0), ld_agent(f.b + j0), ld_agent(f.c + j0)};
      const Cand c1{ld_agent(f.nx + j1), ld_agent(f.ny + j1), ld_agent(f.nz + j1), ld_agent(f.a + j1), ld_agent(f.b + j1), ld_agent(f.c + j1)};
      if (wave == 0 && lane == 0) {
        s_plane[0][0] = c0.nx; s_plane[0][1] = c0.ny; s_plane[0][2] = c0.nz; s_plane[0][3] = c0.a; s_plane[0][4] = c0.b; s_plane[0][5] = c0.c;
        s_plane[1][0] = c1.nx; s_plane[1][1] = c1.ny; s_plane[1][2] = c1.nz; s_plane[1][3] = c1.a; s_plane[1][4] = c1.b; s_plane[1][5] = c1.c;
      }
      // Result-preserving shortcut: once the sweep has passed over them, both predecessors very often hold bitwise
      // the same plane (98 / 87 / 49 % of neighbours after sweeps 0 / 1 / 2 on the C3 pair).  The second evaluation
      // would return the same bits as the first and `cost1 < min(cur, cost0)` would fail, so it is not computed.
      const bool same01 = c0.nx == c1.nx && c0.ny == c1.ny && c0.nz == c1.nz && c0.a == c1.a && c0.b == c1.b && c0.c == c1.c;
      const bool both = have0 && have1 && !same01;
      SWEEP_STAMP(3);
      double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
      if (CS) {
        // this wave's level `wave`: (cur_x, cur_y, cur_disp) after `wave` halvings (pre_cs_pc.cc:139-140,183-185)
        double d0 = c0.a * (double)x + c0.b * (double)y + c0.c, d1 = c1.a * (double)x + c1.b * (double)y + c1.c;
        int cur_x = x, cur_y = y;
        for (int s = 0; s < wave; ++s) { cur_y /= 2; cur_x /= 2; d0 /= 2.0; d1 /= 2.0; }
        double pa, pb, pc;
        plane_param(c0.nx, c0.ny, c0.nz, (double)cur_x, (double)cur_y, d0, pa, pb, pc);  // :144-149
        const LevelArgs A = make_level<SRC>(cd, lut, wave, v, cur_x, cur_y, pa, pb, pc, lane);
        if (both) {
          plane_param(c1.nx, c1.ny, c1.nz, (double)cur_x, (double)cur_y, d1, pa, pb, pc);
          fill_tab(cd, lut.tab + kTab1, A.ox0, A.oy0, pa, pb, pc, lane);
          level_cost_pair<SRC, 2>(cd, A, lut, lane, 256, cd.rounds * 256, tab0, tab1, a0, a1);
        } else {
          level_cost_pair<SRC, 1>(cd, A, lut, lane, 256, cd.rounds * 256, tab0, tab1, a0, a1);
        }
        const double s0 = wave_sum((a0[0] + a0[1]) + (a0[2] + a0[3]));
        const double s1 = both ? wave_sum((a1[0] + a1[1]) + (a1[2] + a1[3])) : s0;
        if (lane == 0) { s_lvl[0][wave] = s0; s_lvl[1][wave] = s1; }
      } else {
        // single scale: wave = accumulator block; tap t = q*256 + wave*64 + lane, one accumulator per lane
        const LevelArgs A = make_level<SRC>(cd, lut, 0, v, x, y, c0.a, c0.b, c0.c, lane);
        if (both) fill_tab(cd, lut.tab + kTab1, A.ox0, A.oy0, c1.a, c1.b, c1.c, lane);
        double p0 = 0.0, p1 = 0.0;
        if (cd.rounds == 5) {  // the usual window: unrolled, all loads of the five taps in flight together
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const TapOwn w = tap_own<SRC>(A, lut, q * 256 + wave * 64 + lane);
            p0 += tap_plane<SRC>(cd, A, lut, tab0, w);
            if (both) p1 += tap_plane<SRC>(cd, A, lut, tab1, w);
          }
        } else {
          for (int q = 0; q < cd.rounds; ++q) {
            const TapOwn w = tap_own<SRC>(A, lut, q * 256 + wave * 64 + lane);
            p0 += tap_plane<SRC>(cd, A, lut, tab0, w);
            if (both) p1 += tap_plane<SRC>(cd, A, lut, tab1, w);
          }
        }
        s_part[0][wave][lane] = p0;
        s_part[1][wave][lane] = both ? p1 : p0;
      }
    }
    SWEEP_STAMP(4);
    __syncthreads();
    SWEEP_STAMP(5);
    // 3. accept (x-predecessor first, then y-predecessor against the updated minimum), publish, raise the flag
    if (wave == 0) {
      double cost0 = 0.0, cost1 = 0.0;
      if (have0 || have1) {
        if (CS) {
          for (int s = 0; s < cd.levels; ++s) {  // :182, levels in order
            cost0 += s_lvl[0][s] * cd.lv[s].wgt;
            cost1 += s_lvl[1][s] * cd.lv[s].wgt;
          }
        } else {
          cost0 = wave_sum((s_part[0][0][lane] + s_part[0][1][lane]) + (s_part[0][2][lane] + s_part[0][3][lane]));
          cost1 = wave_sum((s_part[1][0][lane] + s_part[1][1][lane]) + (s_part[1][2][lane] + s_part[1][3][lane]));
        }
      }
      if (lane == 0) {
        double best_cost = f.cost[i];  // own pixel: nobody else writes it during the sweep
        int pick = -1;
        if (have0 && cost0 < best_cost) { best_cost = cost0; pick = 0; }
        if (have1 && cost1 < best_cost) { best_cost = cost1; pick = 1; }
        if (pick >= 0) {
          st_agent(f.nx + i, s_plane[pick][0]); st_agent(f.ny + i, s_plane[pick][1]); st_agent(f.nz + i, s_plane[pick][2]);
          st_agent(f.a + i, s_plane[pick][3]); st_agent(f.b + i, s_plane[pick][4]); st_agent(f.c + i, s_plane[pick][5]);
          st_agent(f.cost + i, best_cost);
        }
        SWEEP_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the plane is in memory before the flag can be seen
        __hip_atomic_store(sw.done[v] + i, sw.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SWEEP_STAMP(7);
      }
    }
    __syncthreads();  // s_item / s_plane / s_lvl are reused by the next item
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

template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_view_eval(Cost cd, Pm pm, int v, ViewCand vc) {
  __shared__ LutMem<kEvalBlock / kWave> s_lut;
  const Luts lut = load_luts(cd, s_lut);
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
    cost = eval_plane<CS, SRC>(cd, lut, v, cor_x, y, nx, ny, nz, a, b, c, thr, (pm.use_thresh != 0 && *cd.early_ok != 0), lane);  // :266-267
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
      dst.nx[row + cx] = nx; dst.ny[row + cx] = ny; dst.nz[row + cx] = nz;
      dst.a[row + cx] = -nx / denom; dst.b[row + cx] = -ny / denom; dst.c[row + cx] = vc.c[row + x];
    }
  }
  __syncthreads();
  // costs last, so the `c < dst.cost` tests of the passes above saw the pre-pass values
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
// CSPatchMatch::PostProcessing (cs_patchmatch.cc:508-588) on the 8-bit maps.  Deterministic integer /
// f64 work, < 1 % of a run: one thread per pixel, loops exactly in the reference's order so that the
// weighted-median histogram sums round identically.
// ------------------------------------------------------------------------------------------------
// LeftRightCheck (:347-369)
__global__ void k_lr_check(const uint8_t *__restrict__ dis, const uint8_t *__restrict__ other, int W, int H, int v, int dis_scale,
                           int *__restrict__ valid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  int ok = 0;
  const double cur_dis = dis[i] * 1.0 / dis_scale;
  const int other_x = x + (2 * v - 1) * round2int(cur_dis);
  if (other_x >= 0 && other_x < W) {
    const double other_dis = other[(size_t)y * W + other_x] * 1.0 / dis_scale;
    if (fabs(cur_dis - other_dis) <= 0.5 && cur_dis > 0.0) ok = 1;
  }
  valid[i] = ok;
}

__device__ __forceinline__ uint8_t sat_u8(int q) { return (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
__device__ __forceinline__ double plane_disp_at(const Field &f, long long j, int x, int y) {
  double d = f.a[j] * (double)x;  // param().dot(Vec3d(x, y, 1.0))
  d += f.b[j] * (double)y;
  d += f.c[j] * 1.0;
  return d;
}

// FillInvalid (:370-428): nearest valid pixel to the left / right on the row, their planes evaluated at x
__global__ void k_fill_invalid(Pm pm, int v, int dis_scale, const int *__restrict__ valid, uint8_t *__restrict__ dis) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)pm.W * pm.H) return;
  if (valid[i]) return;
  const int y = (int)(i / pm.W), x = (int)(i - (long long)y * pm.W);
  const long long row = (long long)y * pm.W;
  int l_first = x, r_first = x;
  while (l_first >= 0 && !valid[row + l_first]) --l_first;
  while (r_first < pm.W && !valid[row + r_first]) ++r_first;
  const bool l_find = l_first >= 0, r_find = r_first < pm.W;
  const Field &f = pm.f[v];
  if (l_find && r_find) {
    const double l_d = plane_disp_at(f, row + l_first, x, y), r_d = plane_disp_at(f, row + r_first, x, y);
    dis[i] = sat_u8(dis_scale * round2int(l_d <= r_d ? l_d : r_d));
  } else if (l_find) {
    dis[i] = sat_u8(dis_scale * round2int(plane_disp_at(f, row + l_first, x, y)));
  } else if (r_find) {
    dis[i] = sat_u8(dis_scale * round2int(plane_disp_at(f, row + r_first, x, y)));
  }
}

// WeightedMedian(valid, 35, WMF_GAMMA) (:430-506): invalid pixels only, valid neighbours only
__global__ __launch_bounds__(64) void k_weighted_median(const uint32_t *__restrict__ pix, int Wp, int pad, int W, int H,
                                                        const int *__restrict__ valid, const double *__restrict__ lut,
                                                        uint8_t *__restrict__ dis, int half_wnd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  if (valid[i]) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  double disp_hist[256];
  for (int d = 0; d < 256; ++d) disp_hist[d] = 0.0;
  const uint32_t p = pix[(size_t)y * Wp + pad + x];
  double sum_wgt = 0.0;
  for (int wy = -half_wnd; wy <= half_wnd; ++wy) {
    const int qy = y + wy;
    if (qy < 0 || qy >= H) continue;
    for (int wx = -half_wnd; wx <= half_wnd; ++wx) {
      const int qx = x + wx;
      if (qx < 0 || qx >= W) continue;
      const size_t q = (size_t)qy * W + qx;
      if (!valid[q]) continue;
      const int clr_diff = (int)__builtin_amdgcn_sad_u8(p, pix[(size_t)qy * Wp + pad + qx], 0u);
      const double wgt = lut[clr_diff];
      disp_hist[dis[q]] += wgt;
      sum_wgt += wgt;
    }
  }
  const double median_wgt = sum_wgt / 2.0;
  sum_wgt = 0.0;
  int median_disp = 0;
  for (int d = 0; d < 256; ++d) {
    sum_wgt += disp_hist[d];
    if (sum_wgt >= median_wgt) { median_disp = d; break; }
  }
  if (median_wgt > 0.0) dis[i] = (uint8_t)median_disp;
}

// ------------------------------------------------------------------------------------------------
// Image preparation: BGR8 -> packed u32 (padded), pyrDown (pre_cs_pc.cc:45), gray + x-gradient
// (grd_cc.cpp:70-77).  All of it is < 0.1 % of the work: one thread per pixel, nothing clever.
// ------------------------------------------------------------------------------------------------
// fills a whole padded level: interior from packed BGR rows, pad cells with the border constant
__global__ void k_pack_bgr(const uint8_t *__restrict__ src, size_t stride, int W, int H, int Wp, int pad, uint32_t *__restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Wp * H) return;
  const int y = (int)(i / Wp), xp = (int)(i - (long long)y * Wp), x = xp - pad;
  uint32_t v = kBorderPix;
  if (x >= 0 && x < W) {
    const uint8_t *p = src + (size_t)y * stride + 3 * (size_t)x;
    v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  }
  dst[i] = v;
}
// interleave packed colour and gradient into the 16-byte elements the PatchMatch kernels read
__global__ void k_make_aos(const uint32_t *__restrict__ pix, const double *__restrict__ grd, long long n, PixG *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PixG e;
  e.pix = pix[i];
  e.g = grd ? grd[i] : 0.0;
  out[i] = e;
}
// census elements: code of the pixel (unpadded W*H x 3 words) + colour; pad cells are flagged in bit 31 of pix
__global__ void k_make_aos_cen(const uint32_t *__restrict__ pix, const uint32_t *__restrict__ code, int W, int H, int Wp, int pad,
                               PixC *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Wp * H) return;
  const int y = (int)(i / Wp), x = (int)(i - (long long)y * Wp) - pad;
  PixC e;
  if (x >= 0 && x < W) {
    const uint32_t *c = code + 3 * ((size_t)y * W + x);
    e.code[0] = c[0]; e.code[1] = c[1]; e.code[2] = c[2];
    e.pix = pix[i];
  } else {
    e.code[0] = e.code[1] = e.code[2] = 0u;
    e.pix = pix[i] | 0x80000000u;
  }
  out[i] = e;
}
// unpadded W*H packed pixels -> padded level (pad cells = border constant)
__global__ void k_pad_u32(const uint32_t *__restrict__ src, int W, int H, int Wp, int pad, uint32_t *__restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Wp * H) return;
  const int y = (int)(i / Wp), x = (int)(i - (long long)y * Wp) - pad;
  dst[i] = (x >= 0 && x < W) ? src[(size_t)y * W + x] : kBorderPix;
}
__global__ void k_unpack_bgr(const uint32_t *__restrict__ src, int W, int H, int Wp, int pad, uint8_t *__restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const uint32_t p = src[(size_t)y * Wp + pad + x];
  dst[3 * i] = (uint8_t)p; dst[3 * i + 1] = (uint8_t)(p >> 8); dst[3 * i + 2] = (uint8_t)(p >> 16);
}

__device__ __forceinline__ int reflect101(int p, int len) {  // cv::borderInterpolate(BORDER_REFLECT_101)
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// OpenCV 2.4 pyrDown on 8UC3: separable [1 4 6 4 1], integer accumulate, (v+128)>>8, REFLECT_101,
// dst = ((W+1)/2, (H+1)/2).  Writes the whole padded destination level.
__global__ void k_pyrdown(const uint32_t *__restrict__ src, int W, int H, int sWp, int spad, uint32_t *__restrict__ dst, int dW,
                          int dH, int dWp, int dpad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)dWp * dH) return;
  const int y = (int)(i / dWp), xp = (int)(i - (long long)y * dWp), x = xp - dpad;
  if (x < 0 || x >= dW) { dst[i] = kBorderPix; return; }
  const int kw[5] = {1, 4, 6, 4, 1};
  int acc[3] = {0, 0, 0};
  for (int ky = 0; ky < 5; ++ky) {
    const int sy = reflect101(2 * y + ky - 2, H);
    int row[3] = {0, 0, 0};
    for (int kx = 0; kx < 5; ++kx) {
      const uint32_t p = src[(size_t)sy * sWp + spad + reflect101(2 * x + kx - 2, W)];
      row[0] += kw[kx] * (int)(p & 255u);
      row[1] += kw[kx] * (int)((p >> 8) & 255u);
      row[2] += kw[kx] * (int)((p >> 16) & 255u);
    }
    acc[0] += kw[ky] * row[0]; acc[1] += kw[ky] * row[1]; acc[2] += kw[ky] * row[2];
  }
  dst[i] = (uint32_t)((acc[0] + 128) >> 8) | ((uint32_t)((acc[1] + 128) >> 8) << 8) | ((uint32_t)((acc[2] + 128) >> 8) << 16);
}

// pixel sources for the GRD kernels: padded packed u8 image of the ctx, or a CV_64FC3 RGB host volume
struct SrcU32 {
  const uint32_t *p;
  int Wp, pad;
  __device__ __forceinline__ void rgb(int x, int y, double &r, double &g, double &b) const {
    const uint32_t q = p[(size_t)y * Wp + pad + x];
    b = (double)(q & 255u); g = (double)((q >> 8) & 255u); r = (double)((q >> 16) & 255u);
  }
};
struct SrcF64 {
  const double *p;
  int W;
  __device__ __forceinline__ void rgb(int x, int y, double &r, double &g, double &b) const {
    const size_t i = (size_t)y * W + x;
    r = p[3 * i]; g = p[3 * i + 1]; b = p[3 * i + 2];
  }
};

// grd_cc.cpp:70-73: convertTo(CV_32F); cvtColor(CV_RGB2GRAY): gray = R*0.299f + G*0.587f + B*0.114f in float
template <class Src>
__device__ __forceinline__ float gray_at(const Src &s, int x, int y) {
  double r, g, b;
  s.rgb(x, y, r, g, b);
  float t = (float)r * 0.299f;
  t = t + (float)g * 0.587f;
  t = t + (float)b * 0.114f;
  return t;
}
// grd_cc.cpp:76-77: Sobel(gray, CV_64F, 1, 0, ksize=1) = gray[x+1]-gray[x-1] in double, REFLECT_101.
// Output is a padded level (pad cells = BORDER_THRES) when gWp > W, else packed.
template <class Src>
__global__ void k_gradient(Src s, int W, int H, int gWp, int gpad, double *__restrict__ grd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)gWp * H) return;
  const int y = (int)(i / gWp), xp = (int)(i - (long long)y * gWp), x = xp - gpad;
  double g = 3.0;  // BORDER_THRES (grd_cc.h:6)
  if (x >= 0 && x < W) g = (double)gray_at(s, reflect101(x + 1, W), y) - (double)gray_at(s, reflect101(x - 1, W), y);
  grd[i] = g;
}

// GrdCC::buildCV / buildRightCV (cc/grd_cc.cpp:60-154) with myCostGrd (:4-35); one thread per cell,
// slabs d-major as Mat costVol[d] (vol == nullptr: only the max is wanted).  Also reduces the max over
// the volume (pre_cs_pc.cc:75-82).  d0/nd select a slab range (cspm_get_cost_slab in fused mode).
//   left  view: other = right image at x-d, border branch when x-d < 0     (:88-100)
//   right view: other = left image at x+d, border branch when x+d >= wid   (:134-147)
template <class Src>
__global__ __launch_bounds__(256) void k_grd_volume(Src l, Src r, const double *__restrict__ lG, const double *__restrict__ rG,
                                                    int gWp, int gpad, int W, int H, int d0, int nd, int right_view,
                                                    double *__restrict__ vol, unsigned long long *max_key) {
  const long long slab = (long long)W * H;
  const long long cells = slab * nd;
  double best = -1.7976931348623157e308;
  // grid-stride: a bounded grid keeps the number of max-atomics (one per wave) small
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
    double cost;
    const int d = d0 + (int)(i / slab);
    const long long o = i - (long long)(d - d0) * slab;
    const int y = (int)(o / W), x = (int)(o - (long long)y * W);
    const int xo = right_view ? x + d : x - d;
    const bool inside = right_view ? (xo < W) : (xo >= 0);
    double c0, c1, c2, g0;  // own pixel
    double o0, o1, o2, og;  // other-view pixel, or BORDER_THRES
    const size_t grow = (size_t)y * gWp + gpad;
    if (right_view) { r.rgb(x, y, c0, c1, c2); g0 = rG[grow + x]; } else { l.rgb(x, y, c0, c1, c2); g0 = lG[grow + x]; }
    if (inside) {
      if (right_view) { l.rgb(xo, y, o0, o1, o2); og = lG[grow + xo]; } else { r.rgb(xo, y, o0, o1, o2); og = rG[grow + xo]; }
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
    if (vol) vol[i] = cost;
    best = cost > best ? cost : best;
  }
  // wave max -> one atomic per wave
  unsigned long long key = f64_key(best);
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long other = __shfl_xor(key, off, kWave);
    key = other > key ? other : key;
  }
  if (max_key && (threadIdx.x & 63) == 0) atomicMax(max_key, key);
}

// ------------------------------------------------------------------------------------------------
// CenCC (cc/cen_cc.cc:4-137): 8-bit gray, 9x9 census code (80 bits, wrap-around border), Hamming volume.
// ------------------------------------------------------------------------------------------------
// cen_cc.cc:13-16: convertTo(CV_8U) = saturate(round half even), cvtColor(CV_RGB2GRAY) on 8U =
// (R*4899 + G*9617 + B*1868 + (1<<13)) >> 14  (OpenCV 2.4 fixed point)
template <class Src>
__global__ void k_gray8(Src s, int W, int H, uint8_t *__restrict__ gray) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  double r, g, b;
  s.rgb(x, y, r, g, b);
  const double v[3] = {r, g, b};
  int c[3];
  for (int k = 0; k < 3; ++k) {
    const int q = v[k] >= 2147483647.0 ? 2147483647 : (v[k] <= -2147483648.0 ? -2147483647 - 1 : __double2int_rn(v[k]));
    c[k] = q < 0 ? 0 : (q > 255 ? 255 : q);
  }
  gray[i] = (uint8_t)((c[0] * 4899 + c[1] * 9617 + c[2] * 1868 + (1 << 13)) >> 14);
}
__device__ __forceinline__ int wrap_mod(int v, int n) {  // (v + n) % n of cen_cc.cc:30,34, kept non-negative for n < 4
  const int r = v % n;
  return r < 0 ? r + n : r;
}
// cen_cc.cc:19-45: bit k = centre > k-th neighbour of the 9x9 window (row-major, centre skipped), LSB first
__global__ void k_census(const uint8_t *__restrict__ gray, int W, int H, uint32_t *__restrict__ code) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const int c = gray[i];
  uint32_t w[3] = {0u, 0u, 0u};
  int bit = 0;
  for (int wy = -4; wy <= 4; ++wy) {
    const uint8_t *row = gray + (size_t)wrap_mod(y + wy, H) * W;
    for (int wx = -4; wx <= 4; ++wx) {
      if (wy == 0 && wx == 0) continue;
      if (c > row[wrap_mod(x + wx, W)]) w[bit >> 5] |= 1u << (bit & 31);
      ++bit;
    }
  }
  code[3 * i] = w[0]; code[3 * i + 1] = w[1]; code[3 * i + 2] = w[2];
}
// cen_cc.cc:47-66 / 114-133: Hamming distance of the two codes, CENCUS_BIT = 80 where the other view is outside
__global__ __launch_bounds__(256) void k_cen_volume(const uint32_t *__restrict__ lc, const uint32_t *__restrict__ rc, int W, int H, int d0,
                                                    int nd, int right_view, double *__restrict__ vol, unsigned long long *max_key) {
  const long long slab = (long long)W * H, cells = slab * nd;
  double best = -1.7976931348623157e308;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
    const int d = d0 + (int)(i / slab);
    const long long o = i - (long long)(d - d0) * slab;
    const int y = (int)(o / W), x = (int)(o - (long long)y * W);
    const int xo = right_view ? x + d : x - d;
    double cost = 80.0;
    if (right_view ? (xo < W) : (xo >= 0)) {
      const uint32_t *a = (right_view ? rc : lc) + 3 * o, *b = (right_view ? lc : rc) + 3 * ((long long)y * W + xo);
      cost = (double)(__popc(a[0] ^ b[0]) + __popc(a[1] ^ b[1]) + __popc(a[2] ^ b[2]));
    }
    if (vol) vol[i] = cost;
    best = cost > best ? cost : best;
  }
  unsigned long long key = f64_key(best);
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long other = __shfl_xor(key, off, kWave);
    key = other > key ? other : key;
  }
  if (max_key && (threadIdx.x & 63) == 0) atomicMax(max_key, key);
}

// max and min over an uploaded (foreign CCMethod) volume
__global__ __launch_bounds__(256) void k_volume_max(const double *__restrict__ vol, long long cells, unsigned long long *max_key,
                                                    unsigned long long *min_key) {
  unsigned long long key = f64_key(-1.7976931348623157e308), lo = f64_key(1.7976931348623157e308);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long k = f64_key(vol[i]);
    key = k > key ? k : key;
    lo = k < lo ? k : lo;
  }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long other = __shfl_xor(key, off, kWave), olo = __shfl_xor(lo, off, kWave);
    key = other > key ? other : key;
    lo = olo < lo ? olo : lo;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(max_key, key);
    atomicMin(min_key, lo);
  }
}
// keys -> max_cost[view][level] (the reference starts the max at -1.0, pre_cs_pc.cc:75) and the early-exit licence:
// every scale weight (host-checked: wgt_ok), every max_cost and -- for uploaded volumes -- every cell minimum is >= 0
__global__ void k_finish_cost(const unsigned long long *keys, double *out, int n, int levels, double floor_val, int wgt_ok, int check_min,
                              int *early_ok) {
  const int i = threadIdx.x;
  int ok = 1;
  if (i < n) {
    const double v = key_f64(keys[i]);
    const double m = v > floor_val ? v : floor_val;
    out[i] = m;
    const int s = i % CSPM_MAX_LEVELS;
    if (s < levels) {
      if (!(m >= 0.0)) ok = 0;
      if (check_min && !(key_f64(keys[n + i]) >= 0.0)) ok = 0;
    }
  }
  ok = __all(ok);
  if (i == 0) *early_ok = (ok && wgt_ok) ? 1 : 0;
}

}  // namespace cspm
