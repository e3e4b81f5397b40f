// cspm_chain.h -- the CHAIN ENGINE: one wavefront evaluates one or two candidate planes at ONE pixel; the support window
// is spread over the lanes.  It serves the code paths whose pixel order is inherently serial -- the raster sweep of
// CSPatchMatch::SpatialPropagation (cs_patchmatch.cc:163-216), where a pixel needs the final planes of its two
// predecessors -- and single evaluations (cspm_plane_cost_batch, the red-black option).
//
// Lane <-> tap mapping.  Window row dy has kRowMod = 7 interleaved chains (taps dx = j, j+7, j+14, ...: the partial sums
// S[j] of the ROWTREE7 order, cspm_tap.h).  A pass gives 9 rows x 7 chains to lanes 0..62 (lane 63 idles); only rows
// inside the image get lanes (levels 3-4 of a KITTI pyramid are shorter than the window: no masked rows), so a 35x35
// window takes ceil(rows/9) <= 4 passes of 5 taps per lane.  A lane's own-view addresses advance by a constant 7
// elements, the lanes of a row read 7 consecutive elements: row runs, like the window itself.
// A lane's chain j IS the multiplier of the device order's disparity (cspm_tap.h tap_disp): q_disp = fma(a, j, G) with the
// group base G = fma(a, q_x of the step's first column, q_disp_y) -- two fmas per tap and candidate, no tables.
// After the passes the chain sums go through LDS once: lane dy adds the 7 partial sums of row dy, and an xor butterfly
// over the lanes forms the row tree.
#pragma once
#include <type_traits>
#include "cspm_tap.h"

#pragma clang fp contract(off)

namespace cspm {

// per-wave LDS scratch of the chain engine
struct ChainScratch {
  double part[2][kMaxPasses * kWave];  // chain sums of up to two candidates, pass-major
};
// Plane::param() of the candidates of one evaluation at one level
struct ChainPlane { double a, b, c; };

// everything one level needs, wave-uniform
struct ChainLevel {
  int W, H, n, Dm1;
  int ox0, oy0;        // image coordinates of window tap (0,0)
  int r_lo, nrows;     // first window row inside the image, number of rows inside
  int passes, nsteps;  // ceil(nrows / 9), ceil(n / 7)
  int Wp, pad;
  bool has_valid;
  double maxc;
  const char *px, *opx;  // own / other view elements
  int dirE;              // byte step towards larger disparity in the other view: -E (left view) or +E
  double sgn;            // 2*view-1 as a double (GrdPC / CSPC: other_x = q_x + (2*view-1)*q_disp)
  const double *vol;
  const char *vol2;      // kSrcVol2: paired device cells, 16 bytes per (d, y, x)
  size_t slab;
  uint32_t Ip;
};

template <int SRC>
__device__ __forceinline__ ChainLevel make_chain_level(const Cost &cd, int s, int view, int cx, int cy) {
  const Level &L = cd.lv[s];
  constexpr int E = elem_size<SRC>();
  ChainLevel A;
  A.W = L.W; A.H = L.H; A.n = cd.n; A.Dm1 = L.D - 1;
  A.ox0 = cx - cd.half; A.oy0 = cy - cd.half;
  A.r_lo = max(0, -A.oy0);
  const int r_hi = min(cd.n - 1, L.H - 1 - A.oy0);
  A.nrows = r_hi - A.r_lo + 1;
  A.passes = (A.nrows + kChainRows - 1) / kChainRows;
  A.nsteps = (cd.n + kRowMod - 1) / kRowMod;
  A.Wp = L.Wp; A.pad = L.pad;
  A.has_valid = L.D >= 2;
  A.maxc = cd.max_cost[view * CSPM_MAX_LEVELS + s];
  A.vol2 = nullptr;
  if (SRC == kSrcVol2) {
    A.px = reinterpret_cast<const char *>(L.pix[view]); A.opx = A.px;  // colours only: the guide weight
    A.Ip = L.pix[view][cy * L.Wp + L.pad + cx];
    A.vol2 = reinterpret_cast<const char *>(L.vol2[view]);
    A.Dm1 = max(L.D - 1, 1);  // a level with D < 2 has no valid tap (has_valid): the clamped disparity only forms an address, slab 1 exists
  } else if (SRC == kSrcCen) {
    A.px = reinterpret_cast<const char *>(L.pc[view]); A.opx = reinterpret_cast<const char *>(L.pc[1 - view]);
    A.Ip = L.pc[view][cy * L.Wp + L.pad + cx].pix;
  } else if (SRC == kSrcGrd8) {
    A.px = reinterpret_cast<const char *>(L.px8[view]); A.opx = reinterpret_cast<const char *>(L.px8[1 - view]);
    A.Ip = pix8_colour(L.px8[view][cy * L.Wp + L.pad + cx].hi);
  } else {
    A.px = reinterpret_cast<const char *>(L.px[view]); A.opx = reinterpret_cast<const char *>(L.px[1 - view]);
    A.Ip = L.px[view][cy * L.Wp + L.pad + cx].pix;
  }
  A.Ip = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.Ip);  // the window centre's colour: one value per wave
  A.sgn = view == 0 ? -1.0 : 1.0;
  A.dirE = view == 0 ? -E : E;  // left view looks at x-d in the right image, right view at x+d in the left
  A.vol = L.vol[view];
  A.slab = (size_t)L.W * (size_t)L.H;
  return A;
}

#ifdef CSPM_SWEEP_TRACE
constexpr int kTraceSlots = 16;
__shared__ long long *s_tr;  // debug: the stamps of the item this workgroup works on (slots 0-7: wall clock per stage; 8-15: shader cycles inside one chain step)
#define EVAL_STAMP(slot) do { if (wave == 0 && lane == 0 && s_tr) s_tr[slot] = wall_clock64(); } while (0)
// s_memtime ordered after the values `dep...` are available (the compiler must have them in registers before the statement); volatile
// asm statements keep their order.  lgkmcnt(0) also drains the wave's LDS reads: the stamps sit where the code waits for them anyway.
__device__ __forceinline__ unsigned long long step_stamp(unsigned dep0, unsigned dep1) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep0), "v"(dep1) : "memory");
  return t;
}
#else
#define EVAL_STAMP(slot) do { } while (0)
#endif
#if defined(CSPM_SWEEP_TRACE) && defined(CSPM_STEP_TRACE)
#define STEP_STAMP(k, d0, d1) do { if (trace_step) tstamp[k] = step_stamp((unsigned)(d0), (unsigned)(d1)); } while (0)
#else
#define STEP_STAMP(k, d0, d1) do { } while (0)
#endif
// Software pipelining of the chain steps (round 5): a step is a chain of dependent round trips -- gather (260-280 cycles even on an
// empty GPU), exp-table LDS read (110), colour-table LDS read + cell arithmetic (170), ~200 cycles of address arithmetic and
// bookkeeping (profiles/r05_sweep_step_budget.txt) -- and the raster sweep is 1 616 dependent pixel evaluations of 20 such steps
// each.  The gathers of step st+1 (and of the next pass's first step) depend on nothing step st computes: they are issued BEFORE
// step st is consumed, so their round trip hides behind its arithmetic.  Same operations, same order of every sum: identical bits.
// MEASURED (profiles/r05_sweep_*): a sweep pixel's evaluation drops from 8.3 to 6.3 us -- and the sweep takes the same 20 ms: its anti-
// diagonals advance at the pace of the SLOWEST pixels on the critical path through the dependency lattice (two-candidate pixels,
// hand-over), not of the median one, and the pipelined kernel needs 97-109 VGPRs against 81, which other pairs' kernels would use.
// So the raster sweep keeps the plain loop (CSPM_SWEEP_PIPE = 0); single evaluations (cspm_plane_cost_batch, the red-black option)
// run the pipelined one, which keeps it compiled and tested: two schedules of the same arithmetic that must agree bit for bit.
#ifndef CSPM_SWEEP_PIPE
#define CSPM_SWEEP_PIPE 0
#endif
#ifndef CSPM_CHAIN_ALLV
#define CSPM_CHAIN_ALLV 1
#endif

// a value that is the same in every lane of the wave (the candidate planes of a pixel, the window centre's colour): telling the compiler
// so moves it to scalar registers -- the software-pipelined steps need the vector registers (two workgroups of five waves are resident
// per CU above 96 VGPRs, three at or below: tools/ubench/residency.hip)
__device__ __forceinline__ double wave_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// One level, NC candidates (1 or 2) at the same pixel: the plane-independent half of every tap (own element, guide
// weight) is computed once.  `pass_first/pass_step` let several waves share the passes of one level (single-scale sweep).
// The chain sums go straight into LDS: part[c][pass * 64 + lane] (round 5: no S[NC][passes] register array, a run-time pass loop --
// the software-pipelined steps need the registers); finish_level() turns them into the level sum.
template <int SRC, int NC, bool PIPE = false>
__device__ __forceinline__ void chain_passes(const Cost &cd, const ChainLevel &A, const Luts &lut, const ChainPlane (&pl)[NC], int lane,
                                             int pass_first, int pass_step, double *const (&part)[NC]) {
  constexpr int E = SRC == kSrcVol2 ? 4 : elem_size<SRC>();
  const int lutzero = kLutZero;
  const int lr = lane / kRowMod, j = lane - lr * kRowMod;  // lane 63: lr = 9 -> never a valid chain
#if !defined(CSPM_STEP_TRACE)
  if constexpr (PIPE && (SRC == kSrcGrd || SRC == kSrcCen || SRC == kSrcGrd8)) {
    struct PassCtx {
      bool chain_ok;
      int ob;          // byte offset of the chain's first tap
      double ty[NC];   // q_disp_y per candidate, :155
    };
    struct Fetched {   // everything step st needs from memory, and what its addresses were computed from
      uint4 P;
      uint4 o0[NC], o1[NC];  // kSrcGrd8: o0 = the pair {lower, upper address}; o1 unused
      DispSplit d[NC];
      bool ok;
    };
    const double jd = (double)j;
    const int qx0 = A.ox0 + j;
    auto make_ctx = [&](int p) {
      PassCtx c;
      const int r = p * kChainRows + lr;
      c.chain_ok = (lr < kChainRows) & (r < A.nrows);
      const int row = A.oy0 + A.r_lo + (c.chain_ok ? r : 0);  // image row; idle lanes shadow row r_lo with zero weight
      c.ob = (row * A.Wp + A.pad + qx0) * E;
#pragma unroll
      for (int k = 0; k < NC; ++k) c.ty[k] = pl[k].b * (double)row + pl[k].c;
      return c;
    };
    auto fetch = [&](const PassCtx &c, int st, Fetched &F) {
      const double xg = (double)(A.ox0 + kRowMod * st);  // q_x of the group's first column
      const int dx = j + kRowMod * st;
      F.ok = c.chain_ok & (dx < A.n) & ((unsigned)(qx0 + kRowMod * st) < (unsigned)A.W);
      if constexpr (SRC == kSrcGrd8) {
        const uint2 e = ld_pix8(A.px, c.ob + st * (kRowMod * E));
        F.P = uint4{e.x, e.y, 0u, 0u};
      } else {
        F.P = ld_elem<SRC>(A.px, c.ob + st * (kRowMod * E));  // always inside the padded allocation
      }
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const double q_disp = tap_disp(pl[k].a, jd, group_disp(pl[k].a, xg, c.ty[k]));  // :165, device order
        F.d[k] = split_disp(q_disp, A.Dm1, A.has_valid);
        if constexpr (SRC == kSrcGrd8) {
          F.o0[k] = ld_pix8_pair(A.opx, c.ob + st * (kRowMod * E) + __mul24(A.dirE, F.d[k].f) + (A.dirE < 0 ? -E : 0));
        } else {
          const int of = c.ob + st * (kRowMod * E) + __mul24(A.dirE, F.d[k].f);
          F.o0[k] = ld_elem<SRC>(A.opx, of);
          F.o1[k] = ld_elem<SRC>(A.opx, of + A.dirE);
        }
      }
    };
    auto consume = [&](const Fetched &F, double (&acc)[NC]) {
      uint32_t Iq;
      double XP = 0.0;
      if constexpr (SRC == kSrcGrd8) { Iq = pix8_colour(F.P.y); XP = pix8_x(F.P.x, F.P.y); }
      else Iq = pix_of<SRC>(F.P);
      const int sad0 = (int)__builtin_amdgcn_sad_u8(A.Ip, Iq, 0u);
      const int sad = F.ok ? sad0 : lutzero;  // masked taps get weight entry kLutZero = 0.0: they add +0.0
      const double wgt = lut.w[sad];          // :161-164
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        double c0, c1;
        if constexpr (SRC == kSrcGrd8) {
          const bool left = A.dirE < 0;  // the left view reads the right image at x-f and x-f-1: the pair is {f+1, f}
          const uint4 pr = F.o0[k];
          const double clo = grd8_cell(lut.a, Iq, XP, pix8_colour(pr.y), pix8_x(pr.x, pr.y));
          const double chi = grd8_cell(lut.a, Iq, XP, pix8_colour(pr.w), pix8_x(pr.z, pr.w));
          c0 = left ? chi : clo;
          c1 = left ? clo : chi;
        } else {
          cell_pair_of<SRC>(lut.a, F.P, F.o0[k], F.o1[k], c0, c1);
        }
        acc[k] = __builtin_fma(wgt, tap_value(F.d[k], c0, c1, A.maxc), acc[k]);  // :176-177
      }
    };
    // field-wise copies: assigning the structs would also copy their padding bytes, which the compiler does through scratch memory
    auto copy_fetched = [&](Fetched &d, const Fetched &f) {
      d.P = f.P;
      d.ok = f.ok;
#pragma unroll
      for (int k = 0; k < NC; ++k) { d.o0[k] = f.o0[k]; d.o1[k] = f.o1[k]; d.d[k].valid = f.d[k].valid; d.d[k].f = f.d[k].f; d.d[k].fr = f.d[k].fr; }
    };
    auto copy_ctx = [&](PassCtx &d, const PassCtx &c) {
      d.chain_ok = c.chain_ok;
      d.ob = c.ob;
#pragma unroll
      for (int k = 0; k < NC; ++k) d.ty[k] = c.ty[k];
    };
    if (pass_first >= A.passes) return;  // wave-uniform
    PassCtx cur = make_ctx(pass_first);
    Fetched F0, F1;
    fetch(cur, 0, F0);
    const int nsteps = A.nsteps;
    for (int p = pass_first; p < A.passes; p += pass_step) {
      const bool has_next = p + pass_step < A.passes;
      PassCtx nxt;
      if (has_next) nxt = make_ctx(p + pass_step);
      else copy_ctx(nxt, cur);
      double acc[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) acc[k] = 0.0;
      // F0 holds step st.  Issue what comes after it -- step st+1, or the first step of this wave's next pass (of THIS pass again when
      // there is none: a harmless reload) -- then consume it.  The fetches are unconditional: a fetch under a branch makes the
      // compiler wait for ALL outstanding loads (vmcnt(0)) at the join, which undoes the pipelining.
      int st = 0;
      for (; st + 1 < nsteps; st += 2) {
        fetch(cur, st + 1, F1);
        __builtin_amdgcn_sched_barrier(0);
        consume(F0, acc);
        const bool in2 = st + 2 < nsteps;  // wave-uniform
        PassCtx cb;
        cb.chain_ok = in2 ? cur.chain_ok : nxt.chain_ok;
        cb.ob = in2 ? cur.ob : nxt.ob;
#pragma unroll
        for (int k = 0; k < NC; ++k) cb.ty[k] = in2 ? cur.ty[k] : nxt.ty[k];
        fetch(cb, in2 ? st + 2 : 0, F0);
        __builtin_amdgcn_sched_barrier(0);
        consume(F1, acc);
      }
      if (st < nsteps) {  // an odd number of steps: F0 holds the last one; the next pass's first step goes to F1 and moves over
        fetch(nxt, 0, F1);
        __builtin_amdgcn_sched_barrier(0);
        consume(F0, acc);
        copy_fetched(F0, F1);
      }
#pragma unroll
      for (int k = 0; k < NC; ++k) part[k][p * kWave + lane] = acc[k];
      copy_ctx(cur, nxt);
    }
    return;
  }
#endif
  // All-valid levels (round 5): the disparity of a tap is linear in its column and row, so when it lies in [1 + 2^-20, D - 2^-20] at the
  // four corners of the window -- for every candidate -- every tap of the level takes the interpolation branch of pre_cs_pc.cc:166-175:
  // that level runs without clamp, validity test and select (4 of ~29 instructions per tap and candidate; the margin covers the roundings
  // of the device-order disparity against the corner values, as in the row engine's all-valid rows).  True for most pixels of a sweep:
  // its candidates are the neighbours' settled planes.  Same arithmetic for the taps that were valid anyway: identical bits.
  bool allv = (SRC == kSrcGrd || SRC == kSrcCen || SRC == kSrcGrd8 || SRC == kSrcVolume || SRC == kSrcVol2) && A.has_valid;
#if defined(CSPM_STEP_TRACE) || !CSPM_CHAIN_ALLV
  allv = false;
#endif
  if (allv) {
    const double x0 = (double)A.ox0, x1 = (double)(A.ox0 + kRowMod * A.nsteps - 1);  // the last step's masked taps beyond the window form addresses too
    const double y0 = (double)(A.oy0 + A.r_lo), y1 = (double)(A.oy0 + A.r_lo + A.nrows - 1);
    const double lo = 1.0 + 0x1p-20, hi = (double)(A.Dm1 + 1) - 0x1p-20;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double t0 = pl[c].b * y0 + pl[c].c, t1 = pl[c].b * y1 + pl[c].c;
      const double q00 = __builtin_fma(pl[c].a, x0, t0), q01 = __builtin_fma(pl[c].a, x1, t0);
      const double q10 = __builtin_fma(pl[c].a, x0, t1), q11 = __builtin_fma(pl[c].a, x1, t1);
      const double qmin = __builtin_fmin(__builtin_fmin(q00, q01), __builtin_fmin(q10, q11));
      const double qmax = __builtin_fmax(__builtin_fmax(q00, q01), __builtin_fmax(q10, q11));
      allv = allv & (qmin >= lo) & (qmax <= hi);  // false for NaN
    }
    allv = __builtin_amdgcn_ballot_w64(!allv) == 0ull;  // the planes are wave-uniform; this makes the branch below a scalar one
  }
  auto run_passes = [&](auto allv_tag) {
  constexpr bool ALLV = decltype(allv_tag)::value;
  for (int p = pass_first; p < A.passes; p += pass_step) {
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const int r = p * kChainRows + lr;
    const bool chain_ok = (lr < kChainRows) & (r < A.nrows);
    const int dy = A.r_lo + (chain_ok ? r : 0);  // window row; idle lanes shadow row r_lo with zero weight
    const int qx0 = A.ox0 + j;
    const int ob = ((A.oy0 + dy) * A.Wp + A.pad + qx0) * E;  // byte offset of the chain's first tap
    double ty[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ty[c] = pl[c].b * (double)(A.oy0 + dy) + pl[c].c;  // q_disp_y, :155
    const double jd = (double)j;
    for (int st = 0; st < A.nsteps; ++st) {
#if defined(CSPM_SWEEP_TRACE) && defined(CSPM_STEP_TRACE)
      // one step of the item's first level-0 pass is stamped: wave 0 (level 0 / pass 0), its second pass, the middle step
      const bool trace_step = (SRC == kSrcGrd || SRC == kSrcGrd8) && s_tr != nullptr && pass_first == 0 && A.n == cd.n && A.W == cd.lv[0].W && p == 1 && st == 2 &&
                              __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
      unsigned long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      STEP_STAMP(0, st, p);
#endif
      const double xg = (double)(A.ox0 + kRowMod * st);  // q_x of the group's first column
      const int dx = j + kRowMod * st;
      const bool ok = chain_ok & (dx < A.n) & ((unsigned)(qx0 + kRowMod * st) < (unsigned)A.W);
      uint4 P;
      double XP = 0.0;  // kSrcGrd8: the own element's biased gradient
      if constexpr (SRC == kSrcVol2) P = uint4{0u, 0u, *reinterpret_cast<const uint32_t *>(A.px + (size_t)(unsigned)(ob + st * (kRowMod * E))), 0u};
      else if constexpr (SRC == kSrcGrd8) {
        const uint2 e = ld_pix8(A.px, ob + st * (kRowMod * E));
        P = uint4{0u, 0u, pix8_colour(e.y), 0u};
        XP = pix8_x(e.x, e.y);
      } else P = ld_elem<SRC>(A.px, ob + st * (kRowMod * E));  // always inside the padded allocation
      const int sad0 = (int)__builtin_amdgcn_sad_u8(A.Ip, (SRC == kSrcVol2 || SRC == kSrcGrd8) ? P.z : pix_of<SRC>(P), 0u);
      const int sad = ok ? sad0 : lutzero;  // masked taps get weight entry kLutZero = 0.0: they add +0.0
      STEP_STAMP(1, sad0, P.z);             // the own element has arrived
      const double wgt = lut.w[sad];        // :161-164
      STEP_STAMP(2, __double2hiint(wgt), __double2loint(wgt));  // ... and the guide weight from the exp table
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const double q_disp = tap_disp(pl[c].a, jd, group_disp(pl[c].a, xg, ty[c]));  // :165, device order
        if constexpr (SRC == kSrcImg) {
          const ImgSplit g = split_img(q_disp, A.sgn * q_disp, (double)(qx0 + kRowMod * st), A.Dm1, A.has_valid);
          const int fxc = min(max(g.fx, -A.pad), A.W + A.pad - 2);  // clamps only taps of the "impossible disparity" branch
          const int of = ((A.oy0 + dy) * A.Wp + A.pad + fxc) * E;
          const uint4 o0 = ld_elem<SRC>(A.opx, of), o1 = ld_elem<SRC>(A.opx, of + E);
          const double cell = img_cell(pix_of<SRC>(P), g_of(P), pix_of<SRC>(o0), g_of(o0), pix_of<SRC>(o1), g_of(o1), g.fw);
          acc[c] = __builtin_fma(wgt, g.valid ? cell : A.maxc, acc[c]);
          continue;
        }
        const DispSplit d = ALLV ? split_disp_valid(q_disp) : split_disp(q_disp, A.Dm1, A.has_valid);
        double c0, c1;
        if constexpr (SRC == kSrcVol2) {
          // one 16-byte gather: {cell(f), cell(f+1)} of this tap's pixel.  Masked taps (weight 0) read the window centre's column.
          const int qy = A.oy0 + dy, qx = ok ? qx0 + kRowMod * st : A.ox0 + cd.half;
          const unsigned idx = __umul24((unsigned)d.f, (unsigned)A.slab) + (unsigned)(qy * A.W + qx);  // < 2^28 (alloc_cost checks)
          const double2 cc = *reinterpret_cast<const double2 *>(A.vol2 + (size_t)(idx << 4));
          c0 = cc.x;
          c1 = cc.y;
        } else if (SRC == kSrcVolume) {
          const int qy = A.oy0 + dy, qx = ok ? qx0 + kRowMod * st : A.ox0 + cd.half;
          const double *v = A.vol + (size_t)d.f * A.slab + (size_t)qy * A.W + qx;
          c0 = v[0];
          c1 = v[A.slab];
        } else if constexpr (SRC == kSrcGrd8) {
          // the two other-view elements of the tap (disparities f and f+1) are neighbours: ONE 16-byte gather from the lower address
          const bool left = A.dirE < 0;  // the left view reads the right image at x-f and x-f-1
          const int of = ob + st * (kRowMod * E) + __mul24(A.dirE, d.f) + (left ? -E : 0);
          const uint4 pr = ld_pix8_pair(A.opx, of);
          if (c == 0) STEP_STAMP(3, pr.x, pr.w);  // the other view's pair has arrived
          const double clo = grd8_cell(lut.a, P.z, XP, pix8_colour(pr.y), pix8_x(pr.x, pr.y));
          const double chi = grd8_cell(lut.a, P.z, XP, pix8_colour(pr.w), pix8_x(pr.z, pr.w));
          c0 = left ? chi : clo;
          c1 = left ? clo : chi;
          if (c == 0) STEP_STAMP(4, __double2hiint(c0), __double2hiint(c1));  // both cells (colour table round trip included)
        } else {
          const int of = ob + st * (kRowMod * E) + __mul24(A.dirE, d.f);
#if defined(CSPM_SWEEP_TRACE) && defined(CSPM_STEP_TRACE)
          const uint4 o0 = ld_elem<SRC>(A.opx, of), o1 = ld_elem<SRC>(A.opx, of + A.dirE);
          if (c == 0) STEP_STAMP(3, o0.x, o1.x);
          cell_pair_of<SRC>(lut.a, P, o0, o1, c0, c1);
          if (c == 0) STEP_STAMP(4, __double2hiint(c0), __double2hiint(c1));
#else
          cell_pair_of<SRC>(lut.a, P, ld_elem<SRC>(A.opx, of), ld_elem<SRC>(A.opx, of + A.dirE), c0, c1);
#endif
        }
        acc[c] = __builtin_fma(wgt, ALLV ? lerp_cells(d.fr, c0, c1) : tap_value(d, c0, c1, A.maxc), acc[c]);  // :176-177
      }
#if defined(CSPM_SWEEP_TRACE) && defined(CSPM_STEP_TRACE)
      STEP_STAMP(5, __double2hiint(acc[0]), __double2hiint(acc[NC - 1]));  // the step's accumulations
      if (trace_step) {
        STEP_STAMP(6, 0, 0);
        STEP_STAMP(7, 0, 0);  // two stamps back to back: what a stamp itself costs
        if (lane == 0)
          for (int k = 0; k < 8; ++k) s_tr[8 + k] = (long long)tstamp[k];
      }
#endif
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) part[c][p * kWave + lane] = acc[c];
  }
  };
  if (allv) run_passes(std::true_type{});
  else run_passes(std::false_type{});
}

// chain sums of ONE candidate (already stored pass-major in part[]) -> level sum, identical in all lanes:
// lane dy adds the 7 partial sums of window row dy left to right, the butterfly builds the row tree
__device__ __forceinline__ double finish_level(const ChainLevel &A, const double *part, int lane) {
  const int r = lane - A.r_lo;
  const bool in = (r >= 0) & (r < A.nrows);
  const int rc = in ? r : 0;
  const int p = rc / kChainRows, lr = rc - p * kChainRows;
  const double *q = part + p * kWave + lr * kRowMod;
  double R = q[0];
#pragma unroll
  for (int k = 1; k < kRowMod; ++k) R = R + q[k];
  R = in ? R : 0.0;
  return wave_tree_sum(R);
}

// Aggregated plane cost at (x,y) by ONE wave; +inf when the candidate is proven not to beat `thresh` (checked at
// level ends with the exact partial total: all terms are >= 0 when Cost::early_ok).
// (nx,ny,nz) = Plane::norm(), (pa,pb,pc) = Plane::param().
template <bool CS, int SRC>
__device__ __forceinline__ double eval_plane_chain(const Cost &cd, const Luts &lut, ChainScratch &m, int view, int x, int y, double nx,
                                                   double ny, double nz, double pa, double pb, double pc, double thresh,
                                                   bool use_thresh, int lane) {
  double cost = 0.0;
  double cur_disp = pa * (double)x + pb * (double)y + pc;  // pre_cs_pc.cc:139-140
  int cur_x = x, cur_y = y;
  const int levels = CS ? cd.levels : 1;
  for (int s = 0; s < levels; ++s) {
    double a = pa, b = pb, c = pc;
    if (CS) plane_param(nx, ny, nz, (double)cur_x, (double)cur_y, cur_disp, a, b, c);  // :144-149
    const ChainLevel A = make_chain_level<SRC>(cd, s, view, cur_x, cur_y);
    const ChainPlane pl[1] = {{wave_uniform(a), wave_uniform(b), wave_uniform(c)}};
    double *const parts[1] = {m.part[0]};
    wave_lds_fence();  // the previous level's reads of part[] are done
    chain_passes<SRC, 1, true>(cd, A, lut, pl, lane, 0, 1, parts);
    wave_lds_fence();
    const double sc = finish_level(A, m.part[0], lane);
    if (CS) cost += sc * cd.lv[s].wgt;  // :182
    else cost = sc;
    if (use_thresh && cost >= thresh) return __builtin_inf();
    cur_y /= 2;  // :183-185
    cur_x /= 2;
    cur_disp /= 2.0;
  }
  return cost;
}

// ------------------------------------------------------------------------------------------------
// cspm_plane_cost_batch: batched GetPlaneCost on explicit (x,y,plane) tuples -- the parity hook.
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_cost_batch(Cost cd, int view, int n, const int *__restrict__ xy,
                                                           const double *__restrict__ np, double *__restrict__ out) {
  __shared__ LutMem s_lut;
  __shared__ ChainScratch s_m[kEvalBlock / kWave];
  const Luts lut = load_luts(cd, s_lut);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long e = xcd_block() * (kEvalBlock / kWave) + wave;
  if (e >= n) return;
  const int lane = threadIdx.x & 63;
  const int x = xy[2 * e], y = xy[2 * e + 1];
  const double *p = np + 6 * e;
  const double c = eval_plane_chain<CS, SRC>(cd, lut, s_m[wave], view, x, y, p[0], p[1], p[2], p[3], p[4], p[5], kDoubleMax, false, lane);
  if (lane == 0) out[e] = c;
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::SpatialPropagation, red-black half-step (optional lower-quality schedule).
// ------------------------------------------------------------------------------------------------
template <bool CS, int SRC>
__global__ __launch_bounds__(kEvalBlock) void k_spatial_rb(Cost cd, Pm pm, int colour, int inc, int nb) {
  __shared__ LutMem s_lut;
  __shared__ ChainScratch s_m[kEvalBlock / kWave];
  const Luts lut = load_luts(cd, s_lut);
  const int halfW = (pm.W + 1) / 2;
  const long long per_view = (long long)halfW * pm.H;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long e = xcd_block() * (kEvalBlock / kWave) + wave;
  if (e >= 2 * per_view) return;
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
  const bool use_thresh = pm.use_thresh != 0 && *cd.early_ok != 0;
  const int nxs[4] = {x - inc, x, x + inc, x}, nys[4] = {y, y - inc, y, y + inc};
  for (int k = 0; k < nb; ++k) {
    if (nxs[k] < 0 || nxs[k] >= pm.W || nys[k] < 0 || nys[k] >= pm.H) continue;
    const long long j = (long long)nys[k] * pm.W + nxs[k];
    const Cand cand{f.nx[j], f.ny[j], f.nz[j], f.a[j], f.b[j], f.c[j]};
    const double cost = eval_plane_chain<CS, SRC>(cd, lut, s_m[wave], v, x, y, cand.nx, cand.ny, cand.nz, cand.a, cand.b, cand.c, best_cost,
                                                    use_thresh, lane);
    if (cost < best_cost) { best_cost = cost; best = cand; changed = true; }
  }
  if (changed && lane == 0) store_plane(f, i, best.nx, best.ny, best.nz, best.a, best.b, best.c, best_cost);
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::SpatialPropagation in the reference's order (cs_patchmatch.cc:163-216): the in-place
// raster sweep makes pixel (x,y) depend on (x-inc,y) and (x,y-inc) only, so all pixels of one
// anti-diagonal are independent.  A pixel tries the x-predecessor first, then the y-predecessor against
// the updated minimum (:198-212); the first sweep row has only the former (:178-186), the first column only
// the latter (:189-195).
//
// One workgroup evaluates both candidates of one pixel over the same window in one pass (the plane-independent half of
// every tap is computed once).  Work split: cross-scale -> one wave per pyramid level; single-scale -> one wave per
// chain pass.  No early exit: both candidate costs are needed in full when accepted.
// ------------------------------------------------------------------------------------------------
#ifndef CSPM_SWEEP_POLL_SLEEP
#define CSPM_SWEEP_POLL_SLEEP 1  // s_sleep argument (x 64 cycles) between two polls of a predecessor's granules
#endif
#ifndef CSPM_SWEEP_PRIO
#define CSPM_SWEEP_PRIO 3
#endif
#ifndef CSPM_SWEEP_WPL
#define CSPM_SWEEP_WPL 1
#endif
constexpr int kSweepWpl = CSPM_SWEEP_WPL;  // waves per pyramid level in a cross-scale sweep workgroup
constexpr int kSweepMaxWaves = CSPM_MAX_LEVELS * kSweepWpl > kMaxPasses ? CSPM_MAX_LEVELS * kSweepWpl : kMaxPasses;  // cross-scale: a wave per level (x kSweepWpl); single scale: a wave per chain pass

// views into the dynamic LDS of a sweep launch: sized by the waves actually launched (sweep_shared_bytes), so that a second
// kernel -- another stereo pair's refinement -- still finds LDS on the CU
struct SweepShared {
  LutMem &lut;
  double (*lvl)[CSPM_MAX_LEVELS];  // [2][levels]: cross-scale, exact level sums
  ChainScratch *m;                 // one per wave (two candidates' chain sums each)
};
__host__ __device__ inline size_t sweep_shared_bytes(int waves) {
  return sizeof(LutMem) + 2 * CSPM_MAX_LEVELS * sizeof(double) + (size_t)(waves < 1 ? 1 : waves) * sizeof(ChainScratch);
}
__device__ __forceinline__ SweepShared sweep_shared(unsigned char *smem) {
  return SweepShared{*reinterpret_cast<LutMem *>(smem), reinterpret_cast<double (*)[CSPM_MAX_LEVELS]>(smem + sizeof(LutMem)),
                     reinterpret_cast<ChainScratch *>(smem + sizeof(LutMem) + 2 * CSPM_MAX_LEVELS * sizeof(double))};
}

// Both candidate costs at pixel (x,y) of view v; every wave of the workgroup calls it.  `both` = the two candidates differ
// (otherwise only c0 is evaluated and cost1 = cost0).  Results are valid in wave 0 after the call.
// the plane-independent part of a sweep pixel's evaluation: this wave's level (cross-scale) or level 0, its window geometry, the window
// centre's colour (a global load) and max_cost (a scalar load).  The persistent sweep computes it BEFORE it waits for the predecessors'
// planes, so those loads and the address arithmetic overlap the wait instead of following it (0.7 us of every pixel, profiles/r05_sweep_step_budget.txt)
template <bool CS, int SRC>
__device__ __forceinline__ ChainLevel sweep_level_setup(const Cost &cd, int v, int x, int y, int wave) {
  if (CS) {
    const int level = wave / kSweepWpl;
    int cur_x = x, cur_y = y;
    for (int s = 0; s < level; ++s) { cur_y /= 2; cur_x /= 2; }
    return make_chain_level<SRC>(cd, level < cd.levels ? level : 0, v, cur_x, cur_y);
  }
  return make_chain_level<SRC>(cd, 0, v, x, y);
}

template <bool CS, int SRC>
__device__ __forceinline__ void eval_pixel_pair(const Cost &cd, const Luts &lut, const SweepShared &sh, int v, int x, int y, const Cand &c0,
                                                const Cand &c1, bool both, int wave, int lane, double &cost0, double &cost1, const ChainLevel &A) {
  if (CS) {
    // kSweepWpl waves per pyramid level share its chain passes (a sweep pixel is latency-bound: its evaluation is on the
    // critical path of the whole sweep).  Level of this wave: (cur_x, cur_y, cur_disp) after `level` halvings
    // (pre_cs_pc.cc:139-140,183-185).
    const int level = wave / kSweepWpl, part_of = wave - level * kSweepWpl;
    double d0 = c0.a * (double)x + c0.b * (double)y + c0.c, d1 = c1.a * (double)x + c1.b * (double)y + c1.c;
    int cur_x = x, cur_y = y;
    for (int s = 0; s < level; ++s) { cur_y /= 2; cur_x /= 2; d0 /= 2.0; d1 /= 2.0; }
    double *part0 = sh.m[level * kSweepWpl].part[0], *part1 = sh.m[level * kSweepWpl].part[1];  // the level's first wave's scratch serves all its waves
    if (level < cd.levels) {
      ChainPlane pl[2];
      plane_param(c0.nx, c0.ny, c0.nz, (double)cur_x, (double)cur_y, d0, pl[0].a, pl[0].b, pl[0].c);  // :144-149
      pl[0].a = wave_uniform(pl[0].a); pl[0].b = wave_uniform(pl[0].b); pl[0].c = wave_uniform(pl[0].c);
      if (both) {
        plane_param(c1.nx, c1.ny, c1.nz, (double)cur_x, (double)cur_y, d1, pl[1].a, pl[1].b, pl[1].c);
        pl[1].a = wave_uniform(pl[1].a); pl[1].b = wave_uniform(pl[1].b); pl[1].c = wave_uniform(pl[1].c);
        EVAL_STAMP(4);
        double *const parts[2] = {part0, part1};
        chain_passes<SRC, 2, CSPM_SWEEP_PIPE != 0>(cd, A, lut, pl, lane, part_of, kSweepWpl, parts);
        EVAL_STAMP(5);
        if (kSweepWpl == 1) {
          wave_lds_fence();
          const double s0 = finish_level(A, part0, lane);
          const double s1 = finish_level(A, part1, lane);
          if (lane == 0) { sh.lvl[0][level] = s0; sh.lvl[1][level] = s1; }
        }
      } else {
        const ChainPlane p1[1] = {pl[0]};
        double *const parts[1] = {part0};
        chain_passes<SRC, 1, CSPM_SWEEP_PIPE != 0>(cd, A, lut, p1, lane, part_of, kSweepWpl, parts);
        if (kSweepWpl == 1) {
          wave_lds_fence();
          const double s0 = finish_level(A, part0, lane);
          if (lane == 0) { sh.lvl[0][level] = s0; sh.lvl[1][level] = s0; }
        }
      }
    }
    if (kSweepWpl > 1) {
      __syncthreads();  // the chain sums of every level are complete
      if (level < cd.levels) {
        // candidate 0 is finished by the level's first wave, candidate 1 by its second
        if (part_of == 0) {
          const double s0 = finish_level(A, part0, lane);
          if (lane == 0) { sh.lvl[0][level] = s0; if (!both) sh.lvl[1][level] = s0; }
        } else if (part_of == 1 && both) {
          const double s1 = finish_level(A, part1, lane);
          if (lane == 0) sh.lvl[1][level] = s1;
        }
      }
    }
    // FOLDED last level (round 6, CSPM_OPT_SWEEP_FOLD): a workgroup launched with ONE WAVE FEWER than the cost has levels -- four waves,
    // one per SIMD, for the usual five levels -- evaluates the last (coarsest) level with the waves of levels 1 .. nw-1 once they have
    // finished their own: they share its chain passes (a 78 x 24 level of a KITTI-size pair has three), the chain sums meet in the
    // scratch behind the waves' own, and two of them finish the two candidates.  Same taps, same chains, same row tree: identical sums.
    // Why: a five-wave workgroup sits 2 + 1 + 1 + 1 on the four SIMDs and two of them leave room for ONE four-wave workgroup of another
    // pair's refinement (168 VGPRs) where three run on an empty CU; two four-wave sweep workgroups leave room for TWO
    // (profiles/r06_corun.txt: the refinement runs at 36 % of its speed beside a sweep, the sweep is not slowed at all).
    const int nw = (int)(blockDim.x >> 6);
    if (kSweepWpl == 1 && cd.levels == nw + 1) {
      const int last = cd.levels - 1;
      double *fold0 = sh.m[nw].part[0], *fold1 = sh.m[nw].part[1];
      ChainLevel B = A;
      if (wave >= 1) {
        double e0 = c0.a * (double)x + c0.b * (double)y + c0.c, e1 = c1.a * (double)x + c1.b * (double)y + c1.c;
        int lx = x, ly = y;
        for (int s = 0; s < last; ++s) { ly /= 2; lx /= 2; e0 /= 2.0; e1 /= 2.0; }
        B = make_chain_level<SRC>(cd, last, v, lx, ly);
        ChainPlane pl[2];
        plane_param(c0.nx, c0.ny, c0.nz, (double)lx, (double)ly, e0, pl[0].a, pl[0].b, pl[0].c);
        pl[0].a = wave_uniform(pl[0].a); pl[0].b = wave_uniform(pl[0].b); pl[0].c = wave_uniform(pl[0].c);
        if (both) {
          plane_param(c1.nx, c1.ny, c1.nz, (double)lx, (double)ly, e1, pl[1].a, pl[1].b, pl[1].c);
          pl[1].a = wave_uniform(pl[1].a); pl[1].b = wave_uniform(pl[1].b); pl[1].c = wave_uniform(pl[1].c);
          double *const parts[2] = {fold0, fold1};
          chain_passes<SRC, 2, CSPM_SWEEP_PIPE != 0>(cd, B, lut, pl, lane, wave - 1, nw - 1, parts);
        } else {
          const ChainPlane p1[1] = {pl[0]};
          double *const parts[1] = {fold0};
          chain_passes<SRC, 1, CSPM_SWEEP_PIPE != 0>(cd, B, lut, p1, lane, wave - 1, nw - 1, parts);
        }
      }
      __syncthreads();  // the folded level's chain sums are complete
      if (wave == 1) {
        const double s0 = finish_level(B, fold0, lane);
        if (lane == 0) { sh.lvl[0][last] = s0; if (!both) sh.lvl[1][last] = s0; }
      } else if (wave == 2 && both) {
        const double s1 = finish_level(B, fold1, lane);
        if (lane == 0) sh.lvl[1][last] = s1;
      }
    }
    __syncthreads();
    cost0 = cost1 = 0.0;
    if (wave == 0) {
      for (int s = 0; s < cd.levels; ++s) {  // :182, levels in order
        cost0 += sh.lvl[0][s] * cd.lv[s].wgt;
        cost1 += sh.lvl[1][s] * cd.lv[s].wgt;
      }
    }
  } else {
    // single scale: the waves share the passes of the one level; chain sums meet in wave 0's scratch
    const int nw = (int)(blockDim.x >> 6);
    const ChainPlane pl[2] = {{wave_uniform(c0.a), wave_uniform(c0.b), wave_uniform(c0.c)}, {wave_uniform(c1.a), wave_uniform(c1.b), wave_uniform(c1.c)}};
    double *const parts[2] = {sh.m[0].part[0], sh.m[0].part[1]};  // the chain sums of all waves meet in wave 0's scratch
    if (both) {
      chain_passes<SRC, 2, CSPM_SWEEP_PIPE != 0>(cd, A, lut, pl, lane, wave, nw, parts);
    } else {
      const ChainPlane p1[1] = {pl[0]};
      double *const parts1[1] = {parts[0]};
      chain_passes<SRC, 1, CSPM_SWEEP_PIPE != 0>(cd, A, lut, p1, lane, wave, nw, parts1);
    }
    __syncthreads();
    cost0 = cost1 = 0.0;
    if (wave == 0) {
      cost0 = finish_level(A, parts[0], lane);
      cost1 = both ? finish_level(A, parts[1], lane) : cost0;
    }
  }
}

// one launch per anti-diagonal k (sweep coordinates xs+ys == k, image x = inc>0 ? xs : W-1-xs); one workgroup per pixel
template <bool CS, int SRC>
__global__ __launch_bounds__(kSweepMaxWaves *kWave) void k_spatial_diag(Cost cd, Pm pm, int k, int inc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepShared sh = sweep_shared(smem);
  const Luts lut = load_luts(cd, sh.lut);
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
  const bool have0 = xs > 0, have1 = ys > 0;
  if (!have0 && !have1) return;
  const long long jx = i - inc, jy = i - (long long)inc * pm.W;
  const long long j0 = have0 ? jx : jy, j1 = have1 ? jy : jx;
  const Cand c0{f.nx[j0], f.ny[j0], f.nz[j0], f.a[j0], f.b[j0], f.c[j0]};
  const Cand c1{f.nx[j1], f.ny[j1], f.nz[j1], f.a[j1], f.b[j1], f.c[j1]};
  const bool same01 = c0.nx == c1.nx && c0.ny == c1.ny && c0.nz == c1.nz && c0.a == c1.a && c0.b == c1.b && c0.c == c1.c;
  double cost0, cost1;
  const ChainLevel A = sweep_level_setup<CS, SRC>(cd, v, x, y, wave);
  eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c0, c1, have0 && have1 && !same01, wave, lane, cost0, cost1, A);
  if (threadIdx.x == 0) {
    double best_cost = f.cost[i];
    int pick = -1;
    if (have0 && cost0 < best_cost) { best_cost = cost0; pick = 0; }
    if (have1 && cost1 < best_cost) { best_cost = cost1; pick = 1; }
    if (pick >= 0) {
      const Cand &w = pick == 0 ? c0 : c1;
      store_plane(f, i, w.nx, w.ny, w.nz, w.a, w.b, w.c, best_cost);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same raster sweep as ONE persistent launch (default): workgroups pull pixels in diagonal-major
// order from a device-wide counter and wait, per pixel, for the final planes of its two predecessors
// instead of for a kernel boundary.  Dataflow instead of W+H-2 launches per sweep: a pixel starts as
// soon as ITS predecessors are final, diagonals overlap, and the fixed cost per launch is gone.
//
// Inter-workgroup protocol (MI355X: 8 XCDs with private, mutually non-coherent L2s; per-CU L1 never refreshed by other
// CUs' stores): a pixel hands its FINAL plane to its two successors as 12 data-tagged granules -- naturally aligned 8-byte
// words {32 bits of the plane's six doubles, epoch of this sweep}, each written by ONE agent-scope (write-through) store and
// read by agent-scope (L1-bypassing) loads.  The consumer polls the granules themselves: when all tags carry the epoch it
// already holds the data -- no separate flag, no producer-side drain, no second round trip (a flag + payload hand-over
// costs 1.7-1.9x a granule hand-over on this chip).  8-byte accesses are single-copy atomic, so a granule is never torn and
// needs no ordering against the others.  No fences, no reliance on placement or dispatch order.  Deadlock freedom: pixels
// are claimed in an order in which predecessors come first (below), so every granule a workgroup waits for belongs to a pixel
// that a running workgroup has claimed or will claim without waiting for us.  Every spin is bounded (wall clock); a timeout
// raises ctrl[1] and all workgroups drain.
//
// Row bands (round 4).  The image rows are cut into `nbands` (8) bands; workgroup b mod 8 -- on MI355X: the workgroups of XCD
// b mod 8, dispatch being round-robin over the XCDs -- pulls the pixels of ITS band in diagonal-major order from the band's own
// counter.  The global order is unchanged (a pixel still waits for exactly its two predecessors); what changes is who evaluates
// what: an XCD's L2 then only has to hold the window rows of a 47-row band instead of the whole anti-diagonal front (with the
// paired-cell volumes the front of a KITTI pair is ~3 MB per view set against a 4 MiB L2: 8 % of the L2 requests missed).
// Deadlock freedom with bands: within a band claims are in diagonal-major order, so a pixel's in-band predecessors are claimed
// earlier; its predecessor in the band above belongs to a queue that never waits for this band (dependencies only point up
// and left), so by induction over the bands every awaited pixel is reached as long as each band has one running workgroup --
// the first `nbands` workgroups of the grid.  The plane field itself (what later kernels read) is written with plain stores: nobody reads it across workgroups
// inside the sweep except its owner.
// ------------------------------------------------------------------------------------------------
constexpr int kGranPerPixel = 12;
constexpr int kSweepMaxBands = 8;
struct Sweep {
  unsigned int *ctrl;         // [1] error (sticky), [2 + b] next item of row band b
  unsigned long long *gran[2];  // per view, per pixel: kGranPerPixel data-tagged granules {32 bits of the final plane, epoch}
  const unsigned int *start;  // per band b, W+H entries: start[b * (W+H) + k] = the band's items (both views) on diagonals < k
  int nbands;                 // row bands (sweep coordinates): band b = rows [b*H/nbands, (b+1)*H/nbands)
  unsigned int epoch, total;
  long long timeout_ticks;  // bound of one wait for a predecessor, in ticks of the 100 MHz constant clock
  long long *trace;  // debug (-DCSPM_SWEEP_TRACE): 8 wall-clock stamps per item
  // dataflow scheduling (k_spatial_flow)
  unsigned int *ready[2];       // per view and pixel: how many of its predecessors are final (zeroed before the launch)
  unsigned long long *queue;    // ready pixels nobody continued into: {epoch, view * W * H + pixel}, data-tagged like the granules
  unsigned int *qctl;           // [0] slots reserved by poppers, [1] slots filled by pushers, [2] views whose last pixel is final
};
#ifdef CSPM_SWEEP_TRACE
#define SWEEP_STAMP(slot) do { if (threadIdx.x == 0 && sw.trace) sw.trace[((size_t)2 * pm.W * by0 + item) * kTraceSlots + (slot)] = wall_clock64(); } while (0)
#else
#define SWEEP_STAMP(slot) do { } while (0)
#endif

// lanes 0..23 of one wave: lane l polls granule l % 12 of predecessor l / 12 (`need` false: nothing to wait for).  Returns the
// wave-uniform verdict; on success `g` holds the lane's granule.
__device__ __forceinline__ bool wait_granules(const unsigned long long *p, bool need, unsigned int epoch, unsigned int *err, long long timeout_ticks,
                                              unsigned long long &g) {
  const long long t0 = wall_clock64();
  g = 0ull;
  for (unsigned spins = 1;; ++spins) {
    if (need) g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ready = !need || (unsigned int)(g >> 32) == epoch;
    if (__builtin_amdgcn_ballot_w64(!ready) == 0ull) return true;
    __builtin_amdgcn_s_sleep(CSPM_SWEEP_POLL_SLEEP);
    if ((spins & 255u) == 0u) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (wall_clock64() - t0 > timeout_ticks) {  // CSPM_OPT_SWEEP_TIMEOUT_MS, default 3 s
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

template <bool CS, int SRC>
__global__ __launch_bounds__(kSweepMaxWaves *kWave, CSPM_SWEEP_MINW) void k_spatial_sweep(Cost cd, Pm pm, Sweep sw, int inc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepShared sh = sweep_shared(smem);
  __shared__ double s_plane[2][6];
  __shared__ unsigned int s_item;
  __shared__ int s_ok;
  const Luts lut = load_luts(cd, sh.lut);
#if CSPM_SWEEP_PRIO
  // The sweep is a chain of 1 616 dependent pixel evaluations with few waves: when it shares SIMDs with the throughput kernels
  // of other pairs in flight, its instructions go first.
  __builtin_amdgcn_s_setprio(CSPM_SWEEP_PRIO);
#endif
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int ndiag = pm.W + pm.H - 1;
  const int band = (int)(blockIdx.x % (unsigned)sw.nbands);
  const int by0 = (int)((long long)band * pm.H / sw.nbands), by1 = (int)((long long)(band + 1) * pm.H / sw.nbands);  // sweep rows [by0, by1)
  const unsigned int *bstart = sw.start + (size_t)band * (size_t)(ndiag + 1);
  const unsigned int btotal = bstart[ndiag];
  unsigned int *claim = &sw.ctrl[2 + band];
  int k = by0;  // the band's first diagonal
  unsigned int next_item = 0;
  if (threadIdx.x == 0) {
    if (__hip_atomic_load(&sw.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) next_item = btotal;  // an earlier sweep failed
    else next_item = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (;;) {
    if (threadIdx.x == 0) {
      s_item = next_item;
      s_ok = 1;
      // claim the following item now: the atomic's latency hides behind this item's work.  Claims of a
      // workgroup stay increasing, which is all the deadlock argument needs.
      if (next_item < btotal) next_item = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned int item = s_item;
    if (item >= btotal) return;
    SWEEP_STAMP(0);
#ifdef CSPM_SWEEP_TRACE
    if (threadIdx.x == 0) { s_tr = sw.trace ? sw.trace + ((size_t)2 * pm.W * by0 + item) * kTraceSlots : nullptr; if (s_tr) { s_tr[4] = 0; s_tr[5] = 0; for (int k = 8; k < kTraceSlots; ++k) s_tr[k] = 0; } }
#ifndef CSPM_STEP_TRACE
    {  // where the workgroup's waves sit: HW_ID (wave slot, SIMD, CU, SE) of every wave, 12 bits each, and the XCC in slot 15
      __shared__ unsigned int s_hwid[kSweepMaxWaves];
      unsigned int hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      if (lane == 0) s_hwid[wave] = hwid;
      __syncthreads();
      if (threadIdx.x == 0 && sw.trace) {
        long long packed = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6) && k < 5; ++k) packed |= (long long)(s_hwid[k] & 0xFFFu) << (12 * k);
        sw.trace[((size_t)2 * pm.W * by0 + item) * kTraceSlots + 14] = packed;
        sw.trace[((size_t)2 * pm.W * by0 + item) * kTraceSlots + 15] = (long long)(((s_hwid[0] >> 13) & 7u) | ((xcc & 0xFu) << 4));
      }
    }
#endif
#endif
    while (k + 1 < ndiag && item >= bstart[k + 1]) ++k;  // items of one workgroup only increase
    const int ys_lo = max(by0, k - (pm.W - 1)), ys_hi = min(by1 - 1, k);
    const int cnt = ys_hi - ys_lo + 1;
    const int r = (int)(item - bstart[k]);
    const int v = r / cnt;
    const int ys = ys_lo + (r - v * cnt), xs = k - ys;
    const int x = inc > 0 ? xs : pm.W - 1 - xs, y = inc > 0 ? ys : pm.H - 1 - ys;
    const Field &f = pm.f[v];
    const long long i = (long long)y * pm.W + x;
    const long long jx = i - inc, jy = i - (long long)inc * pm.W;
    const bool have0 = xs > 0, have1 = ys > 0;
    SWEEP_STAMP(1);
    // 0. everything of the evaluation that does not depend on the candidate planes, issued before the wait
    const ChainLevel A = sweep_level_setup<CS, SRC>(cd, v, x, y, wave);
    // 1. wait for the predecessors' planes: lanes 0..23 of wave 0 poll one granule each; the data arrives with the tags
    if (wave == 0) {
      const int pred = lane >= kGranPerPixel ? 1 : 0, part = lane - pred * kGranPerPixel;
      const bool need = lane < 2 * kGranPerPixel && (pred == 0 ? have0 : have1);
      unsigned long long g;
      const bool ok = wait_granules(sw.gran[v] + (pred == 0 ? jx : jy) * kGranPerPixel + (need ? part : 0), need, sw.epoch, &sw.ctrl[1], sw.timeout_ticks, g);
      if (lane < 2 * kGranPerPixel) reinterpret_cast<uint32_t *>(&s_plane[0][0])[lane] = (uint32_t)g;  // s_plane[pred][part / 2], half part % 2
      if (lane == 0 && !ok) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) return;
    SWEEP_STAMP(2);
    // 2. both candidate costs in one pass over the window
    double cost0 = 0.0, cost1 = 0.0;
    bool eval0 = false, eval1 = false;
    if (have0 || have1) {
      // with one candidate missing, both slots hold the existing one (the duplicate is computed once)
      const int p0 = have0 ? 0 : 1, p1 = have1 ? 1 : 0;
      const Cand c0{s_plane[p0][0], s_plane[p0][1], s_plane[p0][2], s_plane[p0][3], s_plane[p0][4], s_plane[p0][5]};
      const Cand c1{s_plane[p1][0], s_plane[p1][1], s_plane[p1][2], s_plane[p1][3], s_plane[p1][4], s_plane[p1][5]};
      // Result-preserving shortcuts (no arithmetic skipped that could change an outcome):
      //  * both predecessors hold bitwise the same plane (very common once the sweep has passed over them): the second
      //    evaluation would return the bits of the first and `cost1 < min(cur, cost0)` would fail;
      //  * a predecessor holds bitwise the pixel's OWN plane: its cost here is the stored min_cost (same function, same
      //    pixel, same plane), and `cost < min_cost` fails.  The own plane is only ever written by this workgroup.
      const Cand own{f.nx[i], f.ny[i], f.nz[i], f.a[i], f.b[i], f.c[i]};
      const bool same01 = c0.nx == c1.nx && c0.ny == c1.ny && c0.nz == c1.nz && c0.a == c1.a && c0.b == c1.b && c0.c == c1.c;
      const bool trust = pm.trust_cost != 0;
      const bool own0 = trust && c0.nx == own.nx && c0.ny == own.ny && c0.nz == own.nz && c0.a == own.a && c0.b == own.b && c0.c == own.c;
      const bool own1 = trust && c1.nx == own.nx && c1.ny == own.ny && c1.nz == own.nz && c1.a == own.a && c1.b == own.b && c1.c == own.c;
      eval0 = have0 && !own0;
      eval1 = have1 && !own1 && !(have0 && same01);
      SWEEP_STAMP(3);
      if (eval0 && eval1) {
        eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c0, c1, true, wave, lane, cost0, cost1, A);
      } else if (eval0) {
        eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c0, c0, false, wave, lane, cost0, cost1, A);
      } else if (eval1) {
        eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c1, c1, false, wave, lane, cost1, cost0, A);
      }
    }
    // 3. accept (x-predecessor first, then y-predecessor against the updated minimum; :198-212), publish the FINAL plane.
    //    cost0 / cost1 are uniform in wave 0, so every lane of it takes the same decision.
    if (wave == 0) {
      double best_cost = f.cost[i];  // own pixel: nobody else writes it during the sweep
      int pick = -1;
      if (eval0 && cost0 < best_cost) { best_cost = cost0; pick = 0; }
      if (eval1 && cost1 < best_cost) { best_cost = cost1; pick = 1; }
      SWEEP_STAMP(6);
      const long long comp = f.ny - f.nx;  // the six components of a view's plane field are equally spaced arrays
      if (lane < kGranPerPixel) {
        const int k = lane >> 1;
        const double val = pick >= 0 ? s_plane[pick][k] : f.nx[k * comp + i];
        const unsigned int half = (lane & 1) ? (unsigned int)__double2hiint(val) : (unsigned int)__double2loint(val);
        __hip_atomic_store(sw.gran[v] + (size_t)i * kGranPerPixel + lane, ((unsigned long long)sw.epoch << 32) | half, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      if (pick >= 0) {  // the plane field, for the kernels after the sweep
        if (lane < 6) f.nx[lane * comp + i] = s_plane[pick][lane];
        if (lane == 6) f.cost[i] = best_cost;
      }
      SWEEP_STAMP(7);
    }
    __syncthreads();  // s_item / s_plane / scratch are reused by the next item
  }
}

// ------------------------------------------------------------------------------------------------
// The same raster sweep, scheduled by DATAFLOW (round 5; an OPTION, CSPM_OPT_SWEEP_FLOW: built, bit-identical, measured SLOWER -- 27 ms per
// sweep against 20: see the end of this comment).  What the trace of k_spatial_sweep showed (tools/sweep_trace.py,
// profiles/r05_sweep_critical_path.txt): on the critical path through the dependency lattice a pixel costs 12.4 us, of which the
// evaluation is 6.1, the hand-over of a plane 0.8 -- and 4.3-5.7 us are spent with the successor NOT YET CLAIMED when its last
// predecessor publishes: workgroups claim pixels in a fixed diagonal-major order, so the successor that matters is picked up by
// whichever workgroup happens to finish some other pixel, and workgroups that claimed early hold registers while they wait.
// Here nobody claims ahead and nobody waits for a particular pixel: every pixel counts its final predecessors (`ready`), and the
// workgroup whose pixel completes that count CONTINUES with the successor at once -- its own final plane stays in LDS, the other
// predecessor's plane is already published (one load).  When both successors become ready it continues with one and pushes the other
// to a queue that idle workgroups pop.  Every resident workgroup computes all the time; the order of evaluation differs, the
// dependencies do not: every pixel still sees the FINAL planes of its two predecessors -- the reference's in-place raster order.
// Deadlock freedom: a workgroup only ever blocks on an empty queue slot, holding nothing anybody needs; a ready pixel is either being
// evaluated or in the queue.  Visibility: granules and queue entries are data-tagged with the sweep's epoch and read with agent-scope
// loads, so a reader never trusts an ordering -- a tag that is not there yet just means another poll.  All spins are wall-clock bounded.
// MEASURED (C3, profiles/r05_sweep_critical_path.txt): the late claims are gone (0.7 us per pixel of the critical path), a pixel takes
// 9.0 us from start to publish -- and the sweep takes 26.9 ms instead of 20.0.  Without slack every delay propagates: the realised
// critical path now runs through the TWO-candidate pixels (53 % of its pixels against 4 % of all pixels: they cost 10-11 us and line up
// along the depth edges, which a monotone lattice path can follow), the ready counter's atomic adds 0.8 us to every pixel, and ready
// pixels queue for a free workgroup all the same -- a CU evaluates two pixels at a time at full speed (above).  The ordered sweep hides
// all of that behind its waiting workgroups.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void flow_push(const Sweep &sw, unsigned int id) {
  const unsigned int slot = __hip_atomic_fetch_add(&sw.qctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(sw.queue + slot, ((unsigned long long)sw.epoch << 32) | id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread: reserve the next queue slot and wait for its entry; -1 when the sweep is over (or failed)
__device__ __forceinline__ int flow_pop(const Sweep &sw) {
  if (__hip_atomic_load(&sw.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return -1;  // an earlier sweep failed
  const unsigned int slot = __hip_atomic_fetch_add(&sw.qctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long long t0 = wall_clock64();
  for (unsigned spins = 1;; ++spins) {
    const unsigned long long e = __hip_atomic_load(sw.queue + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned int)(e >> 32) == sw.epoch) return (int)(unsigned int)e;
    __builtin_amdgcn_s_sleep(2);
    if ((spins & 15u) == 0u) {
      if (__hip_atomic_load(&sw.qctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 2u) return -1;  // both views are final
      if (__hip_atomic_load(&sw.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return -1;
      if (wall_clock64() - t0 > sw.timeout_ticks * 4) {  // nothing became ready for far longer than any evaluation takes
        __hip_atomic_store(&sw.ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
      }
    }
  }
}
#ifdef CSPM_SWEEP_TRACE
#define FLOW_STAMP(slot) do { if (threadIdx.x == 0 && sw.trace) sw.trace[(size_t)item * kTraceSlots + (slot)] = wall_clock64(); } while (0)
#else
#define FLOW_STAMP(slot) do { } while (0)
#endif

// Residency (tools/ubench/residency.hip): five-wave workgroups leave one workgroup's worth of wave slots unused -- two are resident per CU
// at 97-128 VGPRs, three at <= 96.  Measured with this kernel capped at 96 (CSPM_FLOW_MINW = 5): the compiler's code under that cap
// evaluates a pixel in 9.3 us instead of 5.4 even on an empty GPU, and three workgroups that all COMPUTE oversubscribe the CU (23 us per
// pixel, 52 ms per sweep).  So: the cap of the ordered sweep, two workgroups per CU.
#ifndef CSPM_FLOW_MINW
#define CSPM_FLOW_MINW CSPM_SWEEP_MINW
#endif
template <bool CS, int SRC>
__global__ __launch_bounds__(kSweepMaxWaves *kWave, CSPM_FLOW_MINW) void k_spatial_flow(Cost cd, Pm pm, Sweep sw, int inc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepShared sh = sweep_shared(smem);
  __shared__ double s_plane[2][6];  // [0] the x-predecessor's final plane, [1] the y-predecessor's
  __shared__ int s_item, s_from, s_ok;
  const Luts lut = load_luts(cd, sh.lut);
#if CSPM_SWEEP_PRIO
  __builtin_amdgcn_s_setprio(CSPM_SWEEP_PRIO);
#endif
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const unsigned int npix = (unsigned int)pm.W * (unsigned int)pm.H;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the two pixels without predecessors
    const unsigned int origin = inc > 0 ? 0u : npix - 1u;
    flow_push(sw, origin);
    flow_push(sw, npix + origin);
  }
  int next = -1, carry = -1;  // thread 0: the pixel this workgroup continues with, and which of its predecessors we are
  for (;;) {
    if (threadIdx.x == 0) {
      int it = next, from = carry;
      if (it < 0) { it = flow_pop(sw); from = -1; }
      s_item = it;
      s_from = from;
      s_ok = 1;
    }
    __syncthreads();
    const int item = s_item;
    if (item < 0) return;
    const int from = s_from;
    FLOW_STAMP(0);
#ifdef CSPM_SWEEP_TRACE
    if (threadIdx.x == 0) { s_tr = sw.trace ? sw.trace + (size_t)item * kTraceSlots : nullptr; if (s_tr) { s_tr[4] = 0; s_tr[5] = 0; for (int k = 8; k < kTraceSlots; ++k) s_tr[k] = 0; s_tr[15] = from; } }
#endif
    const int v = (unsigned int)item >= npix ? 1 : 0;
    const int pix = item - v * (int)npix;
    const int y = pix / pm.W, x = pix - y * pm.W;
    const int xs = inc > 0 ? x : pm.W - 1 - x, ys = inc > 0 ? y : pm.H - 1 - y;
    const Field &f = pm.f[v];
    const long long i = pix;
    const long long jx = i - inc, jy = i - (long long)inc * pm.W;
    const bool have0 = xs > 0, have1 = ys > 0;
    FLOW_STAMP(1);
    // 1. the predecessors' planes.  The one we continued from is in s_plane already; the other one is final (that is what made this
    //    pixel ready) and published: lanes 0..23 of wave 0 load one granule each, issued BEFORE the level set-up so that it overlaps.
    unsigned long long g = 0ull;
    const int pred = lane >= kGranPerPixel ? 1 : 0, part = lane - pred * kGranPerPixel;
    const bool need = wave == 0 && lane < 2 * kGranPerPixel && (pred == 0 ? have0 : have1) && pred != from;
    const unsigned long long *gp = sw.gran[v] + (pred == 0 ? jx : jy) * kGranPerPixel + (need ? part : 0);
    if (need) g = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // 0. everything of the evaluation that does not depend on the candidate planes
    const ChainLevel A = sweep_level_setup<CS, SRC>(cd, v, x, y, wave);
    if (wave == 0) {
      bool ok = true;
      if (__builtin_amdgcn_ballot_w64(need && (unsigned int)(g >> 32) != sw.epoch) != 0ull)  // not visible yet: poll (rare)
        ok = wait_granules(gp, need, sw.epoch, &sw.ctrl[1], sw.timeout_ticks, g);
      if (need) reinterpret_cast<uint32_t *>(&s_plane[0][0])[lane] = (uint32_t)g;  // s_plane[pred][part / 2], half part % 2
      if (lane == 0 && !ok) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) return;
    FLOW_STAMP(2);
    // 2. both candidate costs in one pass over the window (as k_spatial_sweep)
    double cost0 = 0.0, cost1 = 0.0;
    bool eval0 = false, eval1 = false;
    if (have0 || have1) {
      const int p0 = have0 ? 0 : 1, p1 = have1 ? 1 : 0;
      const Cand c0{s_plane[p0][0], s_plane[p0][1], s_plane[p0][2], s_plane[p0][3], s_plane[p0][4], s_plane[p0][5]};
      const Cand c1{s_plane[p1][0], s_plane[p1][1], s_plane[p1][2], s_plane[p1][3], s_plane[p1][4], s_plane[p1][5]};
      const Cand own{f.nx[i], f.ny[i], f.nz[i], f.a[i], f.b[i], f.c[i]};
      const bool same01 = c0.nx == c1.nx && c0.ny == c1.ny && c0.nz == c1.nz && c0.a == c1.a && c0.b == c1.b && c0.c == c1.c;
      const bool trust = pm.trust_cost != 0;
      const bool own0 = trust && c0.nx == own.nx && c0.ny == own.ny && c0.nz == own.nz && c0.a == own.a && c0.b == own.b && c0.c == own.c;
      const bool own1 = trust && c1.nx == own.nx && c1.ny == own.ny && c1.nz == own.nz && c1.a == own.a && c1.b == own.b && c1.c == own.c;
      eval0 = have0 && !own0;
      eval1 = have1 && !own1 && !(have0 && same01);
      FLOW_STAMP(3);
      if (eval0 && eval1) {
        eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c0, c1, true, wave, lane, cost0, cost1, A);
      } else if (eval0) {
        eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c0, c0, false, wave, lane, cost0, cost1, A);
      } else if (eval1) {
        eval_pixel_pair<CS, SRC>(cd, lut, sh, v, x, y, c1, c1, false, wave, lane, cost1, cost0, A);
      } else {
        __syncthreads();  // every wave has read s_plane before wave 0 overwrites it below
      }
    } else {
      FLOW_STAMP(3);
      __syncthreads();
    }
    // 3. accept (x-predecessor first, then y-predecessor against the updated minimum; :198-212), publish the FINAL plane, count it at
    //    the successors and decide what this workgroup does next.
    if (wave == 0) {
      double best_cost = f.cost[i];
      int pick = -1;
      if (eval0 && cost0 < best_cost) { best_cost = cost0; pick = 0; }
      if (eval1 && cost1 < best_cost) { best_cost = cost1; pick = 1; }
      FLOW_STAMP(6);
      const long long comp = f.ny - f.nx;  // the six components of a view's plane field are equally spaced arrays
      double fin = 0.0;  // lanes 0..11: component lane >> 1 of the final plane
      if (lane < kGranPerPixel) {
        const int k = lane >> 1;
        fin = pick >= 0 ? s_plane[pick][k] : f.nx[k * comp + i];
        const unsigned int half = (lane & 1) ? (unsigned int)__double2hiint(fin) : (unsigned int)__double2loint(fin);
        __hip_atomic_store(sw.gran[v] + (size_t)i * kGranPerPixel + lane, ((unsigned long long)sw.epoch << 32) | half, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      if (pick >= 0) {  // the plane field, for the kernels after the sweep
        if (lane < 6) f.nx[lane * comp + i] = s_plane[pick][lane];
        if (lane == 6) f.cost[i] = best_cost;
      }
      // (no wait for the granule stores: a successor that finds a stale tag polls -- the tags, not an ordering, make the data valid)
      // successors: lane 0 the next pixel of the row (this pixel is its x-predecessor), lane 1 the one below (its y-predecessor)
      const bool ex = lane == 0 ? xs + 1 < pm.W : (lane == 1 ? ys + 1 < pm.H : false);
      const long long si = lane == 0 ? i + inc : i + (long long)inc * pm.W;
      const unsigned int want = lane == 0 ? (ys > 0 ? 2u : 1u) : (xs > 0 ? 2u : 1u);
      bool rdy = false;
      if (ex) rdy = __hip_atomic_fetch_add(sw.ready[v] + si, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == want;
      const unsigned long long rb = __builtin_amdgcn_ballot_w64(rdy);
      const bool r0 = (rb & 1ull) != 0ull, r1 = (rb & 2ull) != 0ull;
      const int cont = r0 ? 0 : (r1 ? 1 : -1);  // continue along the row when that pixel is ready, else downwards
      if (lane == 1 && r0 && r1) flow_push(sw, (unsigned int)(v * (int)npix) + (unsigned int)si);
      if (lane == 0) {
        next = cont < 0 ? -1 : v * (int)npix + (int)(cont == 0 ? i + inc : i + (long long)inc * pm.W);
        carry = cont;
        if (xs + 1 == pm.W && ys + 1 == pm.H) __hip_atomic_fetch_add(&sw.qctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // this view is final
      }
      // our final plane becomes predecessor `cont` of the pixel we continue with (nobody else reads s_plane any more)
      if (cont >= 0 && lane < kGranPerPixel && (lane & 1) == 0) s_plane[cont][lane >> 1] = fin;
      FLOW_STAMP(7);
    }
    __syncthreads();  // s_item / s_plane / scratch are reused by the next item
  }
}

}  // namespace cspm
