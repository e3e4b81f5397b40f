// cspm_device.h -- device-side data layout and the scalar building blocks shared by all kernels.
// gfx950 only.  Compiled with -ffp-contract=off: products and sums are individually rounded like the (SSE2, no-FMA)
// reference arithmetic, except at the sites of the tap engines where the DEVICE ORDER contracts a multiply-add by an explicit
// fma (cspm_tap.h; the oracle's CSOR_SUM_DEVICE order does the same).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cspm.h"

#pragma clang fp contract(off)

namespace cspm {

constexpr double kDoubleEps = 0.00000001;  // commfunc.h:26
constexpr double kDoubleMax = 1.7976931348623157e308;  // commfunc.h:27 numeric_limits<double>::max()
constexpr int kLutSize = 768;              // |dB|+|dG|+|dR| <= 765; the reference allocates 1000 (pre_cs_pc.cc:111)
constexpr int kLutZero = 767;              // entry forced to 0.0: masked taps read it, so they add wgt*tmp = +0.0
constexpr int kWave = 64;
constexpr int kEvalBlock = 256;            // chain engine, batch kernels: 4 waves = 4 plane evaluations per workgroup
constexpr uint32_t kBorderPix = 0x00030303u;  // BORDER_THRES in B, G and R (cc/grd_cc.h:6)

// One pyramid level of one PreSSPC/PreCSPC object (pre_cs_pc.h:41-56).
// Images are stored PADDED: row stride Wp = W + 2*pad, image column x at index pad + x, pad = D + half + 8 (every
// address a masked or clamped tap can form -- window overrun, disparity range -- stays inside the row's padding).
// Pad cells hold the GRD border constant (BORDER_THRES = 3 for every channel and for the gradient,
// cc/grd_cc.h:6), so the fused cost needs no border branch (cc/grd_cc.cpp:88-100, 134-147).
// The PatchMatch kernels read the array-of-structs `px`: one 12-byte element per pixel = packed colour
// + x-gradient, so one dwordx3 load fetches everything a tap needs of a pixel (the L1 address path
// charges a wave64 load ~16 cycles whatever its width; the L1 return path charges bytes).
struct __attribute__((packed, aligned(4))) PixG {
  double g;      // x-gradient of the f32 gray image (grd_cc.cpp:70-77); GRD only.  First, so that a dwordx3 load puts it
                 // in an even-aligned VGPR pair (64-bit operands need one; otherwise every load costs two v_mov)
  uint32_t pix;  // B | G<<8 | R<<16 (byte 3 = 0)
};
static_assert(sizeof(PixG) == 12, "PixG must be 12 bytes");

// census flavour of the element (CenCC, cc/cen_cc.cc): 80-bit code + colour; bit 31 of `pix` marks the pad cells,
// whose cost is CENCUS_BIT whatever the own code is (cen_cc.cc:56-62)
struct __attribute__((aligned(16))) PixC {
  uint32_t code[3];
  uint32_t pix;
};
static_assert(sizeof(PixC) == 16, "PixC must be 16 bytes");

// where the PatchMatch kernels get their cell costs from
enum { kSrcVolume = 0, kSrcGrd = 1, kSrcCen = 2,
       kSrcImg = 3 };  // GrdPC / CSPC (plane_cost/grd_pc.cc, cspc.cc): no cells at all -- the other view's colour and gradient are
                       // interpolated at the real-valued column x -+ q_disp.  Elements are PixG with g = Sobel of the 8U gray image;
                       // the pad cells hold the WRAPPED image columns (HandleBorder, commfunc.h:129-145), so no border branch either
enum { kSrcVol2 = 4 };  // raster sweep only (chain engine): GRD device cells materialised as PAIRS, vol2[d][y][x] = {cell(d), cell(d+1)} --
                        // the two cells a tap interpolates between (pre_cs_pc.cc:171-176) arrive with ONE 16-byte gather, the guide
                        // weight needs one 4-byte gather of the own colour: 2 gathers / 20 B per tap instead of 3 / 36 B

enum { kSrcGrd8 = 5 };  // raster sweep only (chain engine), fused GRD: the same pixels as kSrcGrd in PACKED 8-byte elements (Pix8 below) -- a tap's
                        // two adjacent other-view elements are 16 contiguous bytes = ONE dwordx4 gather, its own element one dwordx2 gather:
                        // 2 gathers / 24 B per tap instead of 3 / 36 B through the CU's L1 return path, which is what bounds the sweep

// The x-gradient of a GRD level (grd_cc.cpp:76-77) is gray[x+1] - gray[x-1] of the f32 gray image, and every f32 gray value of an
// 8-bit colour is a multiple of 2^-27 below 256 (0.114f has ulp 2^-27; tests/test_oracle_primitives.py enumerates all 2^24 colours):
// the gradient is an integer multiple of 2^-27 in (-256, 256) -- 36 bits.  With the 24 bits of colour that is a 60-bit pixel:
//   lo  = low 32 bits of u,  u = g * 2^27 + 2^35  (36-bit unsigned)
//   hi  = colour << 8 | u >> 32
// Decoding is exact and nearly free: the double with the bits {(hi & 0xF) | 0x43300000, lo} is X = 2^52 + u, and a GRD cell only needs
// |g_own - g_other| = |X_own - X_other| * 2^-27 (both exact), so the scale folds into the constants of the cell (grd8_cell, cspm_tap.h).
struct __attribute__((aligned(8))) Pix8 {
  uint32_t lo, hi;
};
static_assert(sizeof(Pix8) == 8, "Pix8 must be 8 bytes");
constexpr double kPix8Scale = 134217728.0;      // 2^27
constexpr double kPix8Bias = 34359738368.0;     // 2^35
__host__ __device__ inline bool pix8_encode(uint32_t pix, double g, Pix8 *out) {
  const double t = g * kPix8Scale + kPix8Bias;  // exact for every representable gradient
  const bool ok = t >= 0.0 && t < 68719476736.0 && t == (double)(unsigned long long)t && (pix >> 24) == 0u;
  const unsigned long long u = ok ? (unsigned long long)t : 0ull;
  out->lo = (uint32_t)u;
  out->hi = (pix << 8) | (uint32_t)(u >> 32);
  return ok;
}

struct Level {
  int W, H, D;            // wid_[s], hei_[s], max_disp_[s]
  int Wp, pad;
  const PixG *px[2];      // H rows of Wp (volume and fused-GRD sources)
  const uint4 *px16[2];   // GRD only: image v as the OTHER view's strip slots, H rows of Wp x 16 bytes {gradient (8 B), colour, colour of
                          // the next column towards larger disparity (x-1 in the right image, x+1 in the left image)} -- the exact
                          // LDS image of a strip slot, so the row engine moves it global -> LDS by DMA (cspm_rows.h)
  const Pix8 *px8[2];     // GRD only: the packed 8-byte elements of kSrcGrd8 (raster sweep), H rows of Wp; null unless built
  const PixC *pc[2];      // H rows of Wp (fused-census source)
  const uint32_t *pix[2]; // packed colour only, H rows of Wp (pyramid construction, introspection)
  const double *grd[2];   // x-gradient only, H rows of Wp (GRD volume / max kernels); GRD only
  const double *vol[2];   // cost_vol_[v][s]: (D+1) slabs of H*W doubles, d-major; null when fused
  const double *cvol[2];  // fused GRD only, when it fits the context's budget: the level's DEVICE cells (the bits of grd_cell()) as a volume of
                          // D+1 slabs x H rows x cvW columns, image column x at index cvpad + x (pad columns hold 0.0: only ever read under
                          // weight 0) -- the row engine's cell tables are then filled by LDS-DMA instead of being computed (cspm_rows.h)
  int cvW, cvpad;
  const double2 *vol2[2]; // kSrcVol2: D slabs of H*W pairs {cell(d), cell(d+1)} of the DEVICE cells (same bits as grd_cell()); null unless built
  double wgt;             // scale_wgt_[s]
};

struct Cost {
  int cs;      // 0: PreSSPC::GetPlaneCost, 1: PreCSPC::GetPlaneCost
  int fused;   // kSrcVolume: cells are read from vol; kSrcGrd / kSrcCen: computed on the fly from px / pc
  int levels;
  int half;    // half_wnd_
  int n;       // 2*half+1
  int T;       // n*n taps
  const int *early_ok;    // device flag: all scale weights, max_costs (and the cells of uploaded volumes) are >= 0
  const double *lut;      // lookup_exp_[i] = exp(-i/10), host-computed, kLutSize entries
  const double *lut_a;    // GRD colour term ALPHA*min(i*0.3333333333, TAU_CLR) (grd_cc.cpp:8-18), kLutSize entries
  const double *max_cost; // device, [view*CSPM_MAX_LEVELS + level]
  Level lv[CSPM_MAX_LEVELS];
};

// Plane field of one view, structure of arrays (replaces Plane** plane_[v] / double** min_cost_[v],
// cs_patchmatch.h:139-142).  Plane::point_ is not stored: it only feeds update_param().
struct Field {
  double *nx, *ny, *nz; // Plane::norm_
  double *a, *b, *c;    // Plane::param_
  double *cost;         // min_cost_
};

struct Pm {
  int W, H, max_dis;
  uint64_t seed;
  int rng_row_shared;
  int use_thresh; // early exit enabled
  int trust_cost; // every stored min_cost is the cost of the stored plane at that pixel (false after cspm_set_planes)
  Field f[2];
};

// ---- commfunc.h:117-121 Round2Int: magic-number round-half-even ----
__device__ __forceinline__ int round2int(double d) {
  d = d + 6755399441055744.0;
  return __double2loint(d);
}
// ---- commfunc.h:129-145 ----
__device__ __forceinline__ int handle_border(int loc, int size) {
  if (loc < 0) return loc + size;
  if (loc >= size) return loc - size;
  return loc;
}

// ---- plane.h:25-34 Plane::update_param (dot product in cv::Matx::dot order) ----
__device__ __forceinline__ void plane_param(double nx, double ny, double nz, double px, double py, double pz,
                                            double &a, double &b, double &c) {
  double denom = fmax(fabs(nz), kDoubleEps);
  if (nz < 0.0) denom = -denom;
  a = -nx / denom;
  b = -ny / denom;
  double s = nx * px;
  s += ny * py;
  s += nz * pz;
  c = s / denom;
}

// ---- counter-based RNG (specification in DESIGN.md "RNG"; the oracle implements the same spec) ----
constexpr uint64_t kGold = 0x9E3779B97F4A7C15ULL;
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}
__host__ __device__ __forceinline__ uint32_t stream_id(int phase, int iter, int step, int view) {
  return (uint32_t)((((phase * 16 + iter) * 32 + step) * 2) + view);
}
struct Rng {
  uint64_t base;
  __device__ __forceinline__ Rng(uint64_t seed, uint32_t stream, uint64_t pix) {
    uint64_t k = mix64(seed + kGold * ((uint64_t)stream + 1));
    base = mix64(k ^ (kGold * (pix + 1)));
  }
  __device__ __forceinline__ double u01(uint32_t draw) const {
    uint64_t r = mix64(base + kGold * ((uint64_t)draw + 1));
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
  }
  // cv::RNG::uniform(a,b) = u*(b-a)+a
  __device__ __forceinline__ double uniform(uint32_t draw, double a, double b) const { return u01(draw) * (b - a) + a; }
};

// order-preserving map double -> uint64 (for atomic min/max on costs of either sign)
__device__ __forceinline__ unsigned long long f64_key(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ __forceinline__ double key_f64(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
  union { unsigned long long u; double d; } x;
  x.u = b;
  return x.d;
}

}  // namespace cspm
