// cspm_tap.h -- the arithmetic of ONE window tap of IPlaneCost::GetPlaneCost (pre_ss_pc.cc:94-110, pre_cs_pc.cc:157-179)
// and the summation order shared by both tap engines (gfx950, wave64).
//
// Two engines evaluate plane costs:
//   * cspm_rows.h  -- ONE LANE PER PIXEL: a wavefront owns 64 x-adjacent pixels of an image row and walks the window
//                     serially; the other view's rows are staged through LDS.  Used where many pixels are evaluated
//                     independently (InitRandomPlane, PlaneRefinement, ViewPropagation): >= 95 % of all taps.
//   * cspm_chain.h -- ONE WAVEFRONT PER PIXEL: the window is spread over the lanes.  Used where the pixel order is
//                     serial (the raster sweep of SpatialPropagation) and for single evaluations (cspm_plane_cost_batch).
// Both produce the same bits for the same (pixel, plane) because they add the same terms in the same order:
//
// Summation order "ROWTREE7" (the test oracle restates it as its CSOR_SUM_DEVICE order):
//   - within window row dy, tap dx (0-based window column) is accumulated in dx order into partial sum S[dx % 7];
//     row total R[dy] = (((((S0+S1)+S2)+S3)+S4)+S5)+S6;
//   - level sum = balanced binary tree over R[0..63] (rows beyond the window and rows outside the image are +0.0;
//     neighbours first).  An xor butterfly over 64 lanes computes it; so does one lane with six pending partial sums.
//   Taps outside the image contribute nothing (the reference `continue`s; adding +0.0 is the same thing).
// Contracted multiply-adds (one rounding instead of two; the oracle's CSOR_SUM_DEVICE order does the same):
//   - the tap's disparity (group_disp / tap_disp below), the last step of a GRD cell (grd_cell), the interpolation between
//     the two cells (lerp_cells) and the accumulation S[j] = fma(wgt, value, S[j]).  Everything else is compiled with
//     -ffp-contract=off and rounds like the reference's SSE2 arithmetic.
#pragma once
#include "cspm_device.h"

#pragma clang fp contract(off)

namespace cspm {

constexpr int kRowMod = 7;     // interleaved partial sums per window row
constexpr int kChainRows = 9;  // chain engine: 9 window rows x 7 chains = 63 lanes per pass
constexpr int kMaxWnd = 45;    // window sizes up to 45: <= 64 rows for the row tree, <= 5 chain passes
constexpr int kMaxPasses = (kMaxWnd + kChainRows - 1) / kChainRows;

struct Luts {
  const double *w;  // exp(-i/10), entry kLutZero = 0                 (pre_cs_pc.cc:111-114)
  const double *a;  // ALPHA*min(i*0.3333333333,TAU_CLR)               (grd_cc.cpp:8-18), fused GRD path only
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
template <int SRC> constexpr int elem_size() { return SRC == kSrcCen ? 16 : SRC == kSrcGrd8 ? 8 : 12; }
template <int SRC>
__device__ __forceinline__ uint32_t pix_of(const uint4 &v) { return SRC == kSrcCen ? v.w : v.z; }
// kSrcGrd8 (cspm_device.h Pix8): one element = one dwordx2 load; TWO adjacent elements = one dwordx4 load (8-byte aligned)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
__device__ __forceinline__ uint2 ld_pix8(const char *base, int byte_off) {
  const u32x2 v = *reinterpret_cast<const u32x2 *>(base + (size_t)(unsigned)byte_off);
  return uint2{v.x, v.y};
}
__device__ __forceinline__ uint4 ld_pix8_pair(const char *base, int byte_off) {
  const u32x4 v = *reinterpret_cast<const u32x4_a8 *>(base + (size_t)(unsigned)byte_off);
  return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ uint32_t pix8_colour(uint32_t hi) { return hi >> 8; }
// X = 2^52 + u: the biased, scaled gradient as an exact double (differences of two X are exact multiples of the gradient difference)
__device__ __forceinline__ double pix8_x(uint32_t lo, uint32_t hi) { return __hiloint2double((int)((hi & 0xFu) | 0x43300000u), (int)lo); }
__device__ __forceinline__ double g_of(const uint4 &v) { return __hiloint2double((int)v.y, (int)v.x); }

// myCostGrd (cc/grd_cc.cpp:4-35) on one (own pixel, other pixel) pair; the border variant is the same
// arithmetic on the pad cells.  |dR|+|dG|+|dB| is an exact small integer, so ALPHA*min(sum*0.3333333333,
// TAU_CLR) is a table of the SAD; min(.,TAU_GRD) on finite values is v_min_f64.
// The colour term saturates at SAD = 31 (31 * 0.3333333333 > TAU_CLR = 10): the table is read through min(SAD, 31), so
// its 32 live entries occupy 32 distinct LDS bank pairs and the gather is conflict-free whatever the 64 SADs are.
constexpr unsigned kClrSat = 31;
__device__ __forceinline__ double grd_cell(const double *lut_a, uint32_t Iq, double Gq, uint32_t Io, double Go) {
  const unsigned sad = min(__builtin_amdgcn_sad_u8(Iq, Io, 0u), kClrSat);
  const double grdDiff = __builtin_fmin(fabs(Gq - Go), 2.0);  // TAU_GRD
  return __builtin_fma(1 - 0.1, grdDiff, lut_a[sad]);         // ALPHA*clrDiff + (1-ALPHA)*grdDiff, contracted (device order)
}
// the same cell from packed elements: fabs(Gq - Go) = |Xq - Xo| * 2^-27 exactly, so min(|dG|, TAU_GRD) = 2^-27 * min(|dX|, 2^28) and
// fma(1-ALPHA, grdDiff, clr) = fma((1-ALPHA) * 2^-27, min(|dX|, 2^28), clr): the product is the same real number, rounded once -- the bits of grd_cell()
__device__ __forceinline__ double grd8_cell(const double *lut_a, uint32_t Iq, double Xq, uint32_t Io, double Xo) {
  const unsigned sad = min(__builtin_amdgcn_sad_u8(Iq, Io, 0u), kClrSat);
  const double dX = __builtin_fmin(fabs(Xq - Xo), 268435456.0);        // TAU_GRD * 2^27
  return __builtin_fma((1 - 0.1) * 0x1p-27, dX, lut_a[sad]);
}
// CenCC cell (cc/cen_cc.cc:54-62): Hamming distance of the two 80-bit codes, CENCUS_BIT = 80 when the other view's
// pixel is outside the image (pad cells carry bit 31 in `pix`)
__device__ __forceinline__ double cen_cell(const uint4 &q, const uint4 &o) {
  const int eighty = 80;
  const int ham = __popc(q.x ^ o.x) + __popc(q.y ^ o.y) + __popc(q.z ^ o.z);
  const int cnt = ((int)o.w < 0) ? eighty : ham;
  return (double)cnt;
}
template <int SRC>
__device__ __forceinline__ double cell_of(const double *lut_a, const uint4 &own, const uint4 &other) {
  if (SRC == kSrcCen) return cen_cell(own, other);
  return grd_cell(lut_a, pix_of<SRC>(own), g_of(own), pix_of<SRC>(other), g_of(other));
}

// LDS is addressed with plain 32-bit byte addresses (lds_ld): a tap's address is then ONE integer operation on top of the
// per-row lane constant, every other displacement is an instruction immediate (through generic pointers the compiler re-adds
// the strip base per tap and cannot fold negative displacements).
typedef __attribute__((address_space(3))) const char lds_cchar;
__device__ __forceinline__ int lds_addr(const void *p) { return (int)(uintptr_t)(lds_cchar *)p; }
template <class T> struct LdsVec { typedef T type; };
template <> struct LdsVec<uint2> { typedef u32x2 type; };
template <> struct LdsVec<uint4> { typedef u32x4 type; };
template <class T>
__device__ __forceinline__ T lds_ld(int adr) {
  typedef typename LdsVec<T>::type V;
  const V v = *(__attribute__((address_space(3))) const V *)(uintptr_t)(unsigned)adr;
  if constexpr (sizeof(T) == 16) return T{v[0], v[1], v[2], v[3]};
  else if constexpr (__is_floating_point(T)) return v;
  else if constexpr (sizeof(T) == 8) return T{v[0], v[1]};
  else return v;
}
// The two cells of a tap (own pixel against the other view's pixels at f and f + 1) with the colour look-ups formed together: both SADs
// in one register (v_sad_u8 / v_sad_hi_u8: <= 765 each, 16 bits apart), ONE packed minimum against kClrSat, and the LDS addresses of the two
// table entries with one instruction each (v_mad_u32_u16: a 16-bit half * 8 + table) -- five instructions where two grd_cell() take six.
// The same table entries, the same arithmetic after them: the bits of grd_cell().  `lut_a` MUST point into LDS (Luts::a of load_luts():
// every kernel of the two tap engines; the volume builders, whose table is in global memory, call grd_cell()).
#ifndef CSPM_PAIR_SAD
#define CSPM_PAIR_SAD 1
#endif
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
template <int SRC>
__device__ __forceinline__ void cell_pair_of(const double *lut_a, const uint4 &own, const uint4 &o0, const uint4 &o1, double &c0, double &c1) {
  if constexpr (SRC == kSrcCen || !CSPM_PAIR_SAD) {
    c0 = cell_of<SRC>(lut_a, own, o0);
    c1 = cell_of<SRC>(lut_a, own, o1);
  } else {
    const uint32_t Iq = pix_of<SRC>(own);
    const double Gq = g_of(own);
    uint32_t t = __builtin_amdgcn_sad_u8(Iq, pix_of<SRC>(o0), 0u);
    t = __builtin_amdgcn_sad_hi_u8(Iq, pix_of<SRC>(o1), t);
    const u16x2_t lim = {(unsigned short)kClrSat, (unsigned short)kClrSat};
    t = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, t), lim));
    const int base = lds_addr(lut_a);
    int adr0;
    asm("v_mad_u32_u16 %0, %1, 8, %2" : "=v"(adr0) : "v"(t), "s"(base));  // (t & 0xffff) * 8 + table
    int adr1;
    asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(adr1) : "v"(t), "s"(base));  // (t >> 16) * 8 + table
    const double a0 = lds_ld<double>(adr0), a1 = lds_ld<double>(adr1);
    c0 = __builtin_fma(1 - 0.1, __builtin_fmin(fabs(Gq - g_of(o0)), 2.0), a0);  // grd_cell()'s last two steps
    c1 = __builtin_fma(1 - 0.1, __builtin_fmin(fabs(Gq - g_of(o1)), 2.0), a1);
  }
}
// v_cvt_i32_f64 saturates and maps NaN to 0; written as asm because (int)double is undefined out of range.
__device__ __forceinline__ int cvt_i32_sat(double x) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ int med3_i32(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}

// (The select of the "impossible disparity" branch is left to the compiler.  In isolation `v_cndmask_b32 .., vcc` (VOP2) occupies
// the SIMD for ~23 cycles against 4 for the VOP3 form with the mask in an SGPR pair -- tools/ubench/valu_issue.hip -- but inside
// the tap loop forcing the VOP3 form made k_refine 10 % slower: the masks of a tap batch then live in SGPR pairs across the
// stages and spill; profiles/README.md, round 3.)
// The disparity of a tap split for the interpolation between two integer disparities (:166-175).
//   valid  <=> static_cast<int>(q_disp) in [1, D-1]  <=> 1.0 <= q_disp < D        (else the "impossible disparity" branch)
//   f      = that integer (clamped into [1, D-1] for invalid taps, so addresses stay inside the padded rows)
//   fr     = 1 - floor_wgt = q_disp - f (the reference's ceil weight; floor_wgt = (f+1) - q_disp = 1 - fr)
// For a valid tap q_disp - f is the exact fraction (v_fract_f64) and (f+1) - q_disp is exact too (Sterbenz: f >= 1).
// NaN / out-of-range q_disp saturate in v_cvt_i32_f64 (NaN -> 0) and fail `clamped == raw`, as x86 cvttsd2si's INT_MIN does.
struct DispSplit {
  bool valid;
  int f;
  double fr;
};
__device__ __forceinline__ DispSplit split_disp(double q_disp, int Dm1, bool level_has_valid) {
  DispSplit s;
  const int f0 = cvt_i32_sat(q_disp);
  s.f = med3_i32(f0, 1, Dm1);
  s.valid = (s.f == f0) & level_has_valid;  // level_has_valid: D >= 2 (otherwise [1, D-1] is empty)
  s.fr = __builtin_amdgcn_fract(q_disp);
  return s;
}
// the same for a tap whose q_disp is known to lie in [1, D): no clamp, no test
__device__ __forceinline__ DispSplit split_disp_valid(double q_disp) {
  DispSplit s;
  s.f = cvt_i32_sat(q_disp);
  s.valid = true;
  s.fr = __builtin_amdgcn_fract(q_disp);
  return s;
}
// interpolated cost of the tap (:173-175; the "impossible disparity" branch :166-169).  Device order: floor_wgt*c0 +
// (1-floor_wgt)*c1 is formed as c0 + fr*(c1-c0) with ONE fma (floor_wgt = 1-fr and 1-floor_wgt = fr exactly for a valid tap).
__device__ __forceinline__ double lerp_cells(double fr, double c0, double c1) { return __builtin_fma(fr, c1 - c0, c0); }
__device__ __forceinline__ double tap_value(const DispSplit &s, double c0, double c1, double maxc) {
  const double tmp = lerp_cells(s.fr, c0, c1);
  return s.valid ? tmp : maxc;
}
// A tap's disparity in the device order: formed per group of kRowMod window columns -- q_disp(dx) = fma(a, dx % 7, G) with
// G = fma(a, q_x of the group's first column, q_disp_y)   (the reference: a*q_x + q_disp_y, pre_cs_pc.cc:155,165).  Both
// engines get the multiplier dx % 7 for free (row engine: a compile-time constant; chain engine: a lane constant).
__device__ __forceinline__ double group_disp(double a, double qx_group, double q_disp_y) { return __builtin_fma(a, qx_group, q_disp_y); }
__device__ __forceinline__ double tap_disp(double a, double j, double G) { return __builtin_fma(a, j, G); }

// ---- GrdPC / CSPC tap (plane_cost/grd_pc.cc:125-169 without USE_INTER, cspc.cc:145-174) ----
//   valid  <=> static_cast<int>(q_disp) in [1, D-1], as above
//   fx     = static_cast<int>(other_x), other_x = q_x + (2*view-1)*q_disp (`signed_disp` = that product: exact, +-q_disp)
//   fw     = floor_wgt = (fx+1) - other_x   (truncation, not floor: other_x < 0 near the left border gives weights > 1,
//            exactly as the reference computes them)
struct ImgSplit {
  bool valid;
  int fx;
  double fw;
};
__device__ __forceinline__ ImgSplit split_img(double q_disp, double signed_disp, double qx_d, int Dm1, bool level_has_valid) {
  ImgSplit s;
  const int f0 = cvt_i32_sat(q_disp);
  s.valid = (med3_i32(f0, 1, Dm1) == f0) & level_has_valid;
  const double other_x = qx_d + signed_disp;
  s.fx = cvt_i32_sat(other_x);
  s.fw = (double)(int)((unsigned)s.fx + 1u) - other_x;
  return s;
}
// colour (B, G, R in the reference's channel order) and gradient of the other view interpolated between columns fx and fx+1,
// truncated absolute differences: COST_ALPHA*min(clr,TAU_CLR) + (1-COST_ALPHA)*min(grd,TAU_GRD)
__device__ __forceinline__ double img_cell(uint32_t Iq, double Gq, uint32_t If, double Gf, uint32_t Ic, double Gc, double fw) {
  double t[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = (int)((Iq >> (8 * k)) & 255u), c = (int)((Ic >> (8 * k)) & 255u), f = (int)((If >> (8 * k)) & 255u);
    t[k] = fabs((double)(q - c) + fw * (double)(c - f));  // I_q - I_ceil + floor_wgt * (I_ceil - I_floor)
  }
  double clr = t[0] + t[1];
  clr = clr + t[2];
  clr *= 0.33333333333333;
  const double grd = fabs((Gq - Gc) + fw * (Gc - Gf));
  return 0.1 * __builtin_fmin(clr, 10.0) + (1 - 0.1) * __builtin_fmin(grd, 2.0);
}

// balanced binary tree over the 64 lanes (neighbours first): every lane ends with the same bits because a+b == b+a.
// Round 5: the exchanges are DPP moves and v_readlane instead of six dependent rounds of ds_bpermute (each an LDS-pipe round trip; the
// tree closes every level of every sweep pixel, 1.4 us of a 10 us pixel).  Level g of the tree adds, in every lane, the sum of the lane's
// group of 2^g lanes and the sum of the neighbouring group; after level g all lanes of a group hold the same value, so ANY lane of the
// neighbouring group is a valid source: xor 1 and xor 2 are quad permutes, the 4-lane groups meet by row_half_mirror (i <-> 7-i), the
// 8-lane groups by row_mirror (i <-> 15-i), and the four 16-lane rows are read out with v_readlane: (r0+r1) + (r2+r3) is what the lanes
// of rows 0-1 computed with __shfl_xor, (r2+r3) + (r0+r1) what rows 2-3 did -- the same bits.
// PRECONDITION of wave_tree_sum / wave_min_i32 / wave_max_i32: ALL 64 LANES ACTIVE (EXEC == ~0).  A disabled source lane leaves the DPP
// destination unchanged and v_readlane of a disabled lane returns stale register contents -- neither faults, both give a wrong sum.
// Every caller keeps the whole wave alive (tail lanes shadow a real pixel and are masked at the store); builds with -DCSPM_DEBUG_EXEC trap
// when that is ever not so.
__device__ __forceinline__ void require_full_exec() {
#ifdef CSPM_DEBUG_EXEC
  if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_tree_sum(double v) {
  require_full_exec();
  v = v + dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v = v + dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v = v + dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the 8
  v = v + dpp_f64<0x140>(v);  // row_mirror: the other half of the 16
  const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
  const double a = r0 + r1, b = r2 + r3;
  return a + b;
}

// wave-wide integer min / max, the result in every lane (wave-uniform, in a scalar register): DPP exchanges within the 16-lane rows,
// v_readlane across them -- instead of six dependent ds_bpermute round trips (the row engine reduces its lanes' disparity intervals
// several times per level pass)
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_min_i32(int v) {
  require_full_exec();
  v = min(v, dpp_i32<0xB1>(v));
  v = min(v, dpp_i32<0x4E>(v));
  v = min(v, dpp_i32<0x141>(v));
  v = min(v, dpp_i32<0x140>(v));
  return __builtin_amdgcn_readfirstlane(min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48))));
}
__device__ __forceinline__ int wave_max_i32(int v) {
  require_full_exec();
  v = max(v, dpp_i32<0xB1>(v));
  v = max(v, dpp_i32<0x4E>(v));
  v = max(v, dpp_i32<0x141>(v));
  v = max(v, dpp_i32<0x140>(v));
  return __builtin_amdgcn_readfirstlane(max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48))));
}

// LDS copies of the two lookup tables, one per workgroup
struct LutMem {
  double w[kLutSize];
  double a[kClrSat + 1];  // 256 bytes = one LDS row: entry i lives in banks 2i, 2i+1
};
__device__ __forceinline__ Luts load_luts(const Cost &cd, LutMem &m) {
  for (int i = threadIdx.x; i < kLutSize; i += blockDim.x) m.w[i] = i == kLutZero ? 0.0 : cd.lut[i];
  if (threadIdx.x <= kClrSat) m.a[threadIdx.x] = cd.lut_a[threadIdx.x];
  __syncthreads();
  return Luts{m.w, m.a};
}

// Work item index of this block.  Blocks are dealt round-robin to the 8 XCDs; give XCD k the k-th contiguous
// eighth of the index space, so that each XCD's L2 holds one band of the image.
__device__ __forceinline__ long long xcd_block() {
  const long long per = (long long)gridDim.x / 8;  // gridDim.x is a multiple of 8
  return (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
}

__device__ __forceinline__ void store_plane(const Field &f, long long i, double nx, double ny, double nz, double a, double b,
                                            double c, double cost) {
  f.nx[i] = nx; f.ny[i] = ny; f.nz[i] = nz;
  f.a[i] = a; f.b[i] = b; f.c[i] = c;
  f.cost[i] = cost;
}

struct Cand { double nx, ny, nz, a, b, c; };

}  // namespace cspm
