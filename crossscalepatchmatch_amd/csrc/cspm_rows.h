// cspm_rows.h -- the ROW ENGINE: one LANE per pixel.  A wavefront owns a run of 64 x-adjacent pixels of one image row
// and every lane evaluates ITS OWN candidate plane at ITS OWN pixel, walking the support window serially
// (rows outer, columns inner) -- the loop nest of PreCSPC::GetPlaneCost (pre_cs_pc.cc:151-181) itself.
// Used by InitRandomPlane, PlaneRefinement and ViewPropagation (cs_patchmatch.cc:115-148, 229-277, 292-345):
// per-pixel independent work, > 95 % of all window taps of a run.
//
// Why lanes = pixels.  With lanes = taps (cspm_chain.h) every tap costs three gathers through the CU's L1 address path
// (own element, two of the other view) and that path, not arithmetic, bounds the kernel.  Here
//   * both views' rows are staged once per window row into wave-private LDS strips (other view: [cx_min-half-D, cx_max+half]
//     for the left view; own view: [cx_min-half, cx_max+half]): every tap of every lane reads its operands from the strips
//     with one address computation, however incoherent the 64 planes are (random initialisation, early refinement steps);
//     the strips are coalesced row runs that arrive one window row ahead of the taps -- by LDS-DMA into the other of two
//     strip sets for the fused GRD cost, through registers for the other cost sources;
//   * rows and columns outside the image cost nothing (rows are skipped by scalar control flow; only waves that touch
//     the image border carry the per-tap column mask); rows on which every lane interpolates (nearly all once the planes have
//     settled) run without clamp, validity test and select;
//   * the running sums are registers of the lane: no cross-lane reduction, no per-level table set-up;
//   * where the 64 lanes of a wave agree on a few disparities -- the coarse pyramid levels always, the fine ones once the planes have
//     settled -- the taps read per-row TABLES of cells (and, where they fit, of guide weights) instead: the cells of a window row
//     that the wave can touch are a rectangle (columns x disparities) of the level's device-cell volume and move into LDS by DMA
//     (Level::cvol), or are computed there once per row when the cost object carries no volume (cell mode below).
// Per tap and lane: 23.3 VALU instructions in an all-valid row, 27.4 in a general row, 9.6 - 11.6 in a table row, and 6.3 / 3.6 LDS
// reads, no global load.
#pragma once
#include "cspm_tap.h"

#pragma clang fp contract(off)

namespace cspm {

#ifndef CSPM_ROW_WAVES
#define CSPM_ROW_WAVES 4
#endif
constexpr int kRowWaves = CSPM_ROW_WAVES;     // waves per workgroup (they share the two lookup tables)
#ifndef CSPM_ROW_MINW
#define CSPM_ROW_MINW 3                       // waves per SIMD the register allocator must leave room for: 168 VGPRs (157 used, no
                                              // scratch).  Measured in round 2 (C3, ms per k_refine launch): 2 waves 66.1, 3 waves
                                              // 55.7, 4 waves 58.3 (128 VGPRs), 5 waves 64.3, 6 waves 67.2; round 3: 4 waves per SIMD
                                              // (one 16-wave workgroup per CU, 128 VGPRs, 104 B scratch) 45.1 against 38.5;
                                              // workgroups of 2/3/4 waves at 3 per SIMD are equal, larger ones slower
#endif
#ifndef CSPM_VIEW_MINW
#define CSPM_VIEW_MINW CSPM_ROW_MINW
#endif
#ifndef CSPM_INIT_MINW
#define CSPM_INIT_MINW CSPM_ROW_MINW
#endif
#ifndef CSPM_ROW_EXIT
#define CSPM_ROW_EXIT 1   // early exit tested after every window row (0: at level ends only)
#endif
#ifndef CSPM_CELL_BUMP
#define CSPM_CELL_BUMP 1   // table rows with an immediate pitch: likewise (cell_row_taps)
#endif
#ifndef CSPM_ROW_BUMP
#define CSPM_ROW_BUMP 1    // general rows: the groups of seven step their LDS bases by hand (row_taps)
#endif
#ifndef CSPM_EDGE_ALLV
#define CSPM_EDGE_ALLV 1   // waves at the image's left / right border: all-valid taps (with the column mask) on the rows that allow it
#endif
#ifndef CSPM_TABLE_DMA
#define CSPM_TABLE_DMA 1   // cell tables filled by LDS-DMA from the level's device-cell volume when the cost object carries one (0: always computed)
#endif
#ifndef CSPM_TABLE_CLUSTERS
#define CSPM_TABLE_CLUSTERS 1     // ... and with the rows of TWO disparity clusters when a wave's lanes lie on two surfaces
#endif
#ifndef CSPM_TABLE_DMA_SINGLE
#define CSPM_TABLE_DMA_SINGLE 1   // ... also with ONE table buffer where two do not fit (0: compute the table then)
#endif
#ifndef CSPM_PREFER_WTAB
#define CSPM_PREFER_WTAB 0   // DMA-filled range tables: 1 = the per-row guide-weight table wherever it fits; 0 = per-tap weights for waves of <= CSPM_WTAB_MAXC centres whose window is inside the image (building the table costs more than it saves once the cells are not computed: measured)
#endif
#ifndef CSPM_WTAB_MAXC
#define CSPM_WTAB_MAXC 64
#endif
#ifndef CSPM_CELL_SUB
#define CSPM_CELL_SUB 4  // table rows: taps of a group of seven whose LDS round trips are overlapped (4 + 3)
#endif
#ifndef CSPM_OWN_SINGLE
#define CSPM_OWN_SINGLE 1  // own-view gradient reads of a tap batch as volatile single reads off one base (0: through laundered copies of the base)
#endif
#ifndef CSPM_CELL_PITCH_IMM
#define CSPM_CELL_PITCH_IMM 1  // DMA-filled tables: the usual pitches as instruction immediates (cell_row_taps); 0: always CellRow::stride
#endif
#ifndef CSPM_CELL_PAD
#define CSPM_CELL_PAD 1    // cell tables with a pitch of a multiple of 256 bytes when they fit (no bank conflicts between table rows)
#endif
#ifndef CSPM_RANGE_MODE
#define CSPM_RANGE_MODE 1  // cell mode with range-restricted tables on the levels whose full tables do not fit (0: general taps there)
#endif
#ifndef CSPM_CELL_MODE
#define CSPM_CELL_MODE 1  // coarse levels of the fused GRD cost: per-row cell and weight tables (cell mode below); 0 = always the general taps
#endif
constexpr int kRowBlock = kRowWaves * kWave;
#ifdef CSPM_ROW_STATS
// debug (tools/row_stats.py): per phase slot (0 init, 1 view, 2+step refinement), pyramid level and bucket, the number of staged
// window rows (per wave) whose 64 lanes all take the interpolation branch, bucketed by the number of integer disparities the
// wave touches on that row (<= 4, 8, 16, 32, more); bucket 5 = rows with some lane outside [1, D); bucket 6 = unstaged rows
__device__ unsigned long long g_rowstat[16 * 8 * 8];
// the same slots and levels for the range-restricted cell mode, per level pass of a wave: 0 passes that got to the range test,
// 1 some lane not interpolating everywhere, 2 range tables with the weight table, 3 without, 4 too many disparities for the LDS,
// 5 sum of ND over the passes that took range mode, 6 sum of ND over the passes of bucket 4, 7 full cell mode
__device__ unsigned long long g_rangestat[16 * 8 * 8];
// wave cycles (s_memtime) by level: [level][0] rows of cell / range mode: total, [1] waiting for the row's strips, [2] table build,
// [3] number of such rows; [4..7] the same for the DMA-staged general rows ([6] unused)
__device__ unsigned long long g_rowtime[8 * 8];
// level passes that fail the range test only by the SPAN of their disparities, by the size of the UNION of the lanes' intervals:
// [level][0] <= 8, [1] <= 11, [2] <= 16, [3] <= 24, [4] more
__device__ unsigned long long g_unionstat[8 * 8];
#define ROWTIME_NOW() __builtin_readcyclecounter()
#define ROWTIME_ADD(slot, v) do { rowtime_acc[(slot) & 3] += (unsigned long long)(v); rowtime_base = (slot) & 4; } while (0)
#define ROWTIME_FLUSH() do { if (lane == 0 && rowtime_acc[3]) for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&g_rowtime[s * 8 + rowtime_base + k_], rowtime_acc[k_]); } while (0)
#else
#define ROWTIME_NOW() 0ull
#define ROWTIME_ADD(slot, v) do { } while (0)
#define ROWTIME_FLUSH() do { } while (0)
#endif

// Two wave-private LDS strips per window row (sized by strip_capacity / own_capacity, carved from the launch's dynamic LDS):
//   other view: 16-byte slots, see rd_cells();   own view: gradients (8 B) and colours (4 B) as two arrays (GRD, volumes),
//   16-byte {code, colour} slots (census).
constexpr int kStripRegs = 6;                 // other-view elements a lane carries from global memory to LDS: strips of <= 384 slots
constexpr int kOwnRegs = 2;                   // own-view elements per lane: 64 centres + window <= 128
__host__ __device__ inline int strip_capacity(int max_dis, int half) {
  const int want = kWave + 2 * half + max_dis + 2;  // 64 centres + window + disparity range (level 0 is the widest)
  return want <= kStripRegs * kWave ? want : kStripRegs * kWave;  // wider than that: the level reads both views from global memory
}
__host__ __device__ inline int own_capacity(int half) { return kWave + 2 * half + 2; }
// LDS of one wave.  One strip set = other-view slots (cap x 16 B), own-view gradients (ocap x 8 B), own-view colours (ocap x 4 B).
// The fused-GRD path double-buffers it (the strips of window row dy+1 arrive by LDS-DMA while row dy is evaluated); the other
// cost sources stage through registers into one set of cap + ocap 16-byte slots.
__host__ __device__ inline int strip_set_bytes(int cap, int ocap) { return (cap * 16 + ocap * 12 + 15) / 16 * 16; }
__host__ __device__ inline int wave_lds_bytes(int cap, int ocap) {
  const int dbl = 2 * strip_set_bytes(cap, ocap), single = (cap + ocap) * 16;
  return dbl > single ? dbl : single;
}

// LDS-DMA (gfx950): 64 lanes x 16 (or 4) bytes from per-lane global addresses `sbase + voff` straight into LDS at
// [lds_dst + lane * 16 (4)) -- no VGPR carries the data, no ds_write is issued.  The compiler neither counts these loads nor
// knows that they write LDS: the caller waits (dma_wait) before it reads what they fetched, and M0 -- the destination base,
// compiler-reserved -- is saved and restored inside the statement (the recipe of cdna_hip_programming.md).
__device__ __forceinline__ void dma_b128(const char *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_b32(const char *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// every DMA of this wave has landed and every LDS read it issued has returned (the next DMA may overwrite what they read)
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// a wave-uniform pointer the compiler may be holding in VGPRs -> SGPR pair
__device__ __forceinline__ const char *uniform_ptr(const char *p) {
  const uintptr_t v = (uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char *)(((uintptr_t)hi << 32) | lo);
}

// what a lane holds of one element on its way from global memory to LDS
template <int SRC> struct StripReg { typedef u32x3 type; };
template <> struct StripReg<kSrcCen> { typedef uint4 type; };
template <int SRC>
__device__ __forceinline__ typename StripReg<SRC>::type ld_strip(const char *row, int byte_off) {
  if constexpr (SRC == kSrcCen) return *reinterpret_cast<const uint4 *>(row + (size_t)(unsigned)byte_off);
  else return *reinterpret_cast<const u32x3_a4 *>(row + (size_t)(unsigned)byte_off);
}

// GRD other-view slot k (16 bytes) = {gradient of column k (8 B), colour of column k, colour of the NEXT column towards larger
// disparity (k-1 for the left view, k+1 for the right view)}: one ds_read_b128 gives a tap its first cell and the colour of
// its second, one ds_read_b64 of the neighbouring slot the second gradient -- 24 bytes in two LDS instructions at the
// full 256 B/clk (12-byte ds_read_b96 would run at 96 B/clk, 8-byte reads of 16-byte slots use half the banks).
// (GrdPC / CSPC: slot k carries the colour of column k+1 for both views -- a tap reads columns fx and fx+1.)
// The same as a volatile access: the compiler keeps it a single instruction (two 8-byte reads off one base would be merged into
// ds_read2_b64, which runs at half the rate of two ds_read_b64) and still folds the constant part of the address into the offset field
__device__ __forceinline__ double lds_ld_single(int adr) { return *(volatile __attribute__((address_space(3))) const double *)(uintptr_t)(unsigned)adr; }
// `adr`: LDS address of the LOWER of the two slots a tap reads (left view, GRD / census: slot of x-f-1; otherwise the first cell)
template <int SRC, int VIEW>
__device__ __forceinline__ void rd_cells(int adr, uint4 &o0, uint4 &o1) {
  constexpr bool down = VIEW == 0 && SRC != kSrcImg;  // second cell one slot below the first
  constexpr int a0 = down ? 16 : 0, a1 = down ? 0 : 16;
  if constexpr (SRC == kSrcCen) {
    o0 = lds_ld<uint4>(adr + a0);
    o1 = lds_ld<uint4>(adr + a1);
  } else {
    const uint4 s0 = lds_ld<uint4>(adr + a0);
    const uint2 g1 = lds_ld<uint2>(adr + a1);
    o0 = uint4{s0.x, s0.y, s0.z, 0u};
    o1 = uint4{g1.x, g1.y, s0.w, 0u};
  }
}
template <int SRC> struct StageReg { uint4 v; };
template <int SRC, int VIEW>
__device__ __forceinline__ StageReg<SRC> ld_stage(const char *row, int byte_off) {
  constexpr int E = elem_size<SRC>();
  StageReg<SRC> r;
  if constexpr (SRC == kSrcCen) {
    r.v = *reinterpret_cast<const uint4 *>(row + (size_t)(unsigned)byte_off);
  } else {
    const u32x3 e = *reinterpret_cast<const u32x3_a4 *>(row + (size_t)(unsigned)byte_off);
    const uint32_t nb = *reinterpret_cast<const uint32_t *>(row + (size_t)(unsigned)(byte_off + ((VIEW == 0 && SRC != kSrcImg) ? -E : E) + 8));
    r.v = uint4{e.x, e.y, e.z, nb};
  }
  return r;
}
// own-view strip: element of window column dx of a lane.  GRD / volumes: adr_g addresses the gradient array (8-byte stride),
// adr_p the colour array (4-byte stride) -- consecutive lanes hit consecutive banks; census: adr_g addresses 16-byte slots.
template <int SRC>
__device__ __forceinline__ uint4 rd_own(int adr_g, int adr_p, int j) {
  if constexpr (SRC == kSrcCen) {
    return lds_ld<uint4>(adr_g + j * 16);
  } else {
    const uint32_t pix = lds_ld<uint32_t>(adr_p + j * 4);
    uint2 g{0u, 0u};
    if constexpr (SRC == kSrcGrd || SRC == kSrcImg) {
#if CSPM_OWN_SINGLE
      const u32x2 v = *(volatile __attribute__((address_space(3))) const u32x2 *)(uintptr_t)(unsigned)(adr_g + j * 8);  // never ds_read2_b64 (lds_ld_single)
      g = uint2{v[0], v[1]};
#else
      g = lds_ld<uint2>(adr_g + j * 8);
#endif
    }
    return uint4{g.x, g.y, pix, 0u};
  }
}
template <int SRC>
__device__ __forceinline__ void wr_own(char *ostrip, int ocap, int idx, const typename StripReg<SRC>::type &e) {
  if constexpr (SRC == kSrcCen) {
    *reinterpret_cast<uint4 *>(ostrip + idx * 16) = e;
  } else {
    if constexpr (SRC == kSrcGrd || SRC == kSrcImg) *reinterpret_cast<uint2 *>(ostrip + idx * 8) = uint2{e.x, e.y};
    *reinterpret_cast<uint32_t *>(ostrip + ocap * 8 + idx * 4) = e.z;
  }
}
// base + f * K for a small compile-time K (one v_mad_i32_i24; K is an inline constant)
template <int K>
__device__ __forceinline__ int mad_const(int f, int base) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(f), "n"(K), "v"(base));
  return r;
}

// per-level constants of a wave
struct RowLevel {
  int W, n, half, Dm1;
  bool has_valid;
  double maxc;
  const double *vol;
  size_t slab;
};

// Row tree of the ROWTREE7 order as a binary counter: p<k> holds the sum of a completed block of 2^k window rows.
// `idx` (the window row number) is wave-uniform: the cascade is scalar control flow around at most one add per level.
struct RowTree {
  double p0, p1, p2, p3, p4, p5;
  __device__ __forceinline__ void push(int idx, double R) {
    double t = R;
    if ((idx & 1) == 0) { p0 = t; return; }
    t = p0 + t;
    if ((idx & 2) == 0) { p1 = t; return; }
    t = p1 + t;
    if ((idx & 4) == 0) { p2 = t; return; }
    t = p2 + t;
    if ((idx & 8) == 0) { p3 = t; return; }
    t = p3 + t;
    if ((idx & 16) == 0) { p4 = t; return; }
    t = p4 + t;
    p5 = t;  // idx & 32 == 0 always: windows have at most 45 rows
  }
  // the tree over 64 leaves whose rows >= count are +0.0: fold the pending blocks, small (late) ones first
  __device__ __forceinline__ double total(int count) const {
    double t = 0.0;
    bool have = false;
    if (count & 1) { t = p0; have = true; }
    if (count & 2) { t = have ? p1 + t : p1; have = true; }
    if (count & 4) { t = have ? p2 + t : p2; have = true; }
    if (count & 8) { t = have ? p3 + t : p3; have = true; }
    if (count & 16) { t = have ? p4 + t : p4; have = true; }
    if (count & 32) { t = have ? p5 + t : p5; }
    return t;
  }
};

// where the taps of one window row find their operands
struct RowSrc {
  // staged: LDS addresses (lds_addr) of window column 0 of the lane: other strip at f = 0 (biased one slot down for the left
  // view, see rd_cells); own strip gradient / colour arrays
  int adr_o, adr_g, adr_p;
  int adr_g2, adr_g3;  // == adr_g, but opaque to the compiler (see tap_batch stage 1)
  // unstaged: image rows in global memory and the byte offset of the lane's window column 0
  const char *own_row, *oth_row;
  int lane_off;
  // GrdPC / CSPC: the other view is addressed by image column fx (wave-uniform base, no per-lane part), clamped into the
  // range the strip / the padded row holds (only taps of the "impossible disparity" branch are ever clamped)
  int img_base, fx_lo, fx_hi;
};

// CNT consecutive taps (window columns g0 .. g0+CNT-1, CNT <= 7) of one window row for all 64 lanes, accumulated into the
// partial sums S[0..CNT-1] (g0 is a multiple of 7, so tap g0+j belongs to S[j]).  Everything that varies with j is an
// immediate: the taps form one basic block, and their LDS reads are in flight together.
//   VIEW   : 0 = left view (other view read at x-f, x-f-1), 1 = right view (x+f, x+f+1)
//   EDGE   : some lane's window leaves the image in x -> per-tap mask (e_rel = g0 - e_lo per lane, e_span)
//   STAGED : operands come from the two LDS strips, else from global memory
#ifndef CSPM_ROW_SUB
#define CSPM_ROW_SUB 2  // taps whose memory round trips are overlapped (sub-batch of a group of 7): register pressure vs latency hiding
#endif
//   ALLV   : every tap of this window row is known to take the interpolation branch in every lane (level_rows decides that per
//            row from the two end columns): no clamp, no validity test, no select -- 4 instructions per tap less
template <int SRC, int VIEW, bool EDGE, bool STAGED, bool ALLV, int J0, int J1>
__device__ __forceinline__ void tap_batch(const RowLevel &A, const Luts &lut, const RowSrc &R, int g0, int adr_o, int adr_g, int adr_g2, int adr_g3, int adr_p, int off_g,
                                          uint32_t Ip, double pa, double Gg, double qxg_d, int e_rel, int e_span, int qy, int cx_lane,
                                          double S[kRowMod]) {
  constexpr int E = elem_size<SRC>();
  constexpr int dirS = VIEW == 0 ? -16 : 16, dirE = VIEW == 0 ? -E : E;
  constexpr int N = J1 - J0;
  const int lutzero = kLutZero;
  // The taps are written stage by stage, not tap by tap: a tap is a chain of three dependent memory round trips (own
  // element -> guide weight; disparity -> the two cells of the other view -> colour term), and the chains overlap only if
  // the reads of one stage are issued back to back.
  uint4 P[N], o0[N], o1[N];
  double fr[N], wgt[N], tmp[N];
  bool valid[N], in_img[N];
  // stage 1: own elements.  (Two 8-byte reads off the same register would be merged into one ds_read2_b64, which runs at half the
  // rate of ds_read_b64: the gradient reads are volatile -- until round 4 they went through laundered copies of the base.)
  if constexpr (STAGED && CSPM_OWN_SINGLE && (SRC == kSrcGrd || SRC == kSrcImg)) {
    // colours first (they may pair up into ds_read2_b32, same rate), then the gradients as single 8-byte reads
#pragma unroll
    for (int k = 0; k < N; ++k) P[k].z = lds_ld<uint32_t>(adr_p + (J0 + k) * 4);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const u32x2 v = *(volatile __attribute__((address_space(3))) const u32x2 *)(uintptr_t)(unsigned)(adr_g + (J0 + k) * 8);  // never ds_read2_b64
      P[k].x = v[0]; P[k].y = v[1]; P[k].w = 0u;
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k)
      P[k] = STAGED ? rd_own<SRC>(CSPM_OWN_SINGLE || k == 0 ? adr_g : k == 1 ? adr_g2 : adr_g3, adr_p, J0 + k) : ld_elem<SRC>(R.own_row, off_g + (J0 + k) * E);
  }
  // stage 2: disparities (pure arithmetic), then the cells of the other view
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int j = J0 + k;
    const double q_disp = tap_disp(pa, (double)j, Gg);  // :165, device order (cspm_tap.h)
    in_img[k] = true;
    if (EDGE) in_img[k] = (unsigned)(e_rel + j) <= (unsigned)e_span;
    if constexpr (SRC == kSrcImg) {
      const ImgSplit g = split_img(q_disp, VIEW == 0 ? -q_disp : q_disp, qxg_d + (double)j, A.Dm1, A.has_valid);  // the tap's column (exact)
      fr[k] = g.fw;
      valid[k] = ALLV ? true : g.valid;
      const int fxc = min(max(g.fx, R.fx_lo), R.fx_hi);
      if (STAGED) {
        rd_cells<SRC, VIEW>(fxc * 16 + R.img_base, o0[k], o1[k]);
      } else {
        o0[k] = ld_elem<SRC>(R.oth_row, fxc * E + R.img_base);
        o1[k] = ld_elem<SRC>(R.oth_row, fxc * E + R.img_base + E);
      }
      continue;
    }
    const DispSplit d = ALLV ? split_disp_valid(q_disp) : split_disp(q_disp, A.Dm1, A.has_valid);
    fr[k] = d.fr;
    valid[k] = d.valid;
    if (SRC == kSrcVolume) {
      const int qx = in_img[k] ? cx_lane - A.half + g0 + j : cx_lane;
      const double *v = A.vol + (size_t)d.f * A.slab + (size_t)qy * A.W + qx;
      tmp[k] = lerp_cells(d.fr, v[0], v[A.slab]);
    } else if (STAGED) {
      rd_cells<SRC, VIEW>(mad_const<dirS>(d.f, adr_o) + j * 16, o0[k], o1[k]);
    } else {
      const int of = mad_const<dirE>(d.f, off_g);
      o0[k] = ld_elem<SRC>(R.oth_row, of + j * E);
      o1[k] = ld_elem<SRC>(R.oth_row, of + j * E + dirE);
    }
  }
  // stage 3: guide weights (:161-164); outside the image: weight entry kLutZero = 0.0, the tap adds +0.0
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int sad = (int)__builtin_amdgcn_sad_u8(Ip, pix_of<SRC>(P[k]), 0u);
    if (EDGE) sad = in_img[k] ? sad : lutzero;
    wgt[k] = lut.w[sad];
  }
  // stage 4: cell costs (the colour term is the third round trip), interpolated (:171-175)
  if constexpr (SRC == kSrcImg) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      tmp[k] = img_cell(pix_of<SRC>(P[k]), g_of(P[k]), pix_of<SRC>(o0[k]), g_of(o0[k]), pix_of<SRC>(o1[k]), g_of(o1[k]), fr[k]);
  } else if (SRC != kSrcVolume) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double c0, c1;
      cell_pair_of<SRC>(lut.a, P[k], o0[k], o1[k], c0, c1);
      tmp[k] = lerp_cells(fr[k], c0, c1);
    }
  }
  // stage 5: the "impossible disparity" branch (:166-169), weight (:176), accumulate
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double maxc = A.maxc;
    const double t = (ALLV || valid[k]) ? tmp[k] : maxc;
    S[J0 + k] = __builtin_fma(wgt[k], t, S[J0 + k]);
  }
}

// CNT consecutive taps (window columns g0 .. g0+CNT-1, CNT <= 7) of one window row for all 64 lanes, accumulated into the
// partial sums S[0..CNT-1] (g0 is a multiple of 7, so tap g0+j belongs to S[j]).  Everything that varies with j is an
// immediate.
//   VIEW   : 0 = left view (other view read at x-f, x-f-1), 1 = right view (x+f, x+f+1)
//   EDGE   : some lane's window leaves the image in x -> per-tap mask (e_rel = g0 - e_lo per lane, e_span)
//   STAGED : operands come from the two LDS strips, else from global memory
template <int SRC, int VIEW, bool EDGE, bool STAGED, bool ALLV, int CNT>
__device__ __forceinline__ void tap_group(const RowLevel &A, const Luts &lut, const RowSrc &R, int g0, int ga, uint32_t Ip, double pa, double rowterm,
                                          double &qxg_d, int e_rel, int e_span, int qy, int cx_lane, double S[kRowMod]) {
  constexpr int E = elem_size<SRC>();
  constexpr int SUB = CSPM_ROW_SUB;
  const double Gg = group_disp(pa, qxg_d, rowterm);  // the group's disparity base (device order, cspm_tap.h)
  const double qxg = qxg_d;
  qxg_d += (double)kRowMod;                          // exact: small integers
  // ga: the group's first window column as far as the addresses go (g0, or 0 when row_taps steps the bases itself)
  const int adr_o = R.adr_o + ga * 16, adr_g = R.adr_g + ga * (SRC == kSrcCen ? 16 : 8), adr_p = R.adr_p + ga * 4, off_g = R.lane_off + ga * E;
  const int adr_g2 = R.adr_g2 + ga * (SRC == kSrcCen ? 16 : 8), adr_g3 = R.adr_g3 + ga * (SRC == kSrcCen ? 16 : 8);
  tap_batch<SRC, VIEW, EDGE, STAGED, ALLV, 0, (CNT < SUB ? CNT : SUB)>(A, lut, R, g0, adr_o, adr_g, adr_g2, adr_g3, adr_p, off_g, Ip, pa, Gg, qxg, e_rel, e_span, qy,
                                                                  cx_lane, S);
  if constexpr (CNT > SUB)
    tap_batch<SRC, VIEW, EDGE, STAGED, ALLV, SUB, (CNT < 2 * SUB ? CNT : 2 * SUB)>(A, lut, R, g0, adr_o, adr_g, adr_g2, adr_g3, adr_p, off_g, Ip, pa, Gg, qxg, e_rel,
                                                                              e_span, qy, cx_lane, S);
  if constexpr (CNT > 2 * SUB)
    tap_batch<SRC, VIEW, EDGE, STAGED, ALLV, 2 * SUB, CNT>(A, lut, R, g0, adr_o, adr_g, adr_g2, adr_g3, adr_p, off_g, Ip, pa, Gg, qxg, e_rel, e_span, qy, cx_lane, S);
}

// One window row of one level for all 64 lanes -> row total R (ROWTREE7: seven interleaved partial sums, combined left to right)
template <int SRC, int VIEW, bool EDGE, bool STAGED, bool ALLV = false>
__device__ __forceinline__ double row_taps(const RowLevel &A, const Luts &lut, const RowSrc &R, uint32_t Ip, double pa, double rowterm,
                                           double qx0_d, int e_lo, int e_span, int qy, int cx_lane) {
  double S[kRowMod];
#pragma unroll
  for (int j = 0; j < kRowMod; ++j) S[j] = 0.0;
  double qx_d = qx0_d;
  const int full = A.n / kRowMod * kRowMod;
  int g0 = 0;
  if constexpr (CSPM_ROW_BUMP && STAGED && SRC == kSrcGrd) {
    // The three LDS bases are stepped by hand and laundered: left to itself the loop optimiser keeps bases WITHOUT the row's buffer offset
    // and re-adds it in every group (seven address additions per group instead of three).
    RowSrc Rg = R;
    for (; g0 < full; g0 += kRowMod) {
      tap_group<SRC, VIEW, EDGE, STAGED, ALLV, kRowMod>(A, lut, Rg, g0, 0, Ip, pa, rowterm, qx_d, g0 - e_lo, e_span, qy, cx_lane, S);
      Rg.adr_o += kRowMod * 16; Rg.adr_g += kRowMod * 8; Rg.adr_p += kRowMod * 4;
      asm("" : "+v"(Rg.adr_o), "+v"(Rg.adr_g), "+v"(Rg.adr_p));
    }
  } else {
    for (; g0 < full; g0 += kRowMod)
      tap_group<SRC, VIEW, EDGE, STAGED, ALLV, kRowMod>(A, lut, R, g0, g0, Ip, pa, rowterm, qx_d, g0 - e_lo, e_span, qy, cx_lane, S);
  }
  // window sizes that are not a multiple of 7 (the usual 35 is): the remaining 1..6 taps
  switch (A.n - full) {
#define CSPM_TAIL(K) case K: tap_group<SRC, VIEW, EDGE, STAGED, ALLV, K>(A, lut, R, g0, g0, Ip, pa, rowterm, qx_d, g0 - e_lo, e_span, qy, cx_lane, S); break;
    CSPM_TAIL(1) CSPM_TAIL(2) CSPM_TAIL(3) CSPM_TAIL(4) CSPM_TAIL(5) CSPM_TAIL(6)
#undef CSPM_TAIL
    default: break;
  }
  double Rsum = S[0];
#pragma unroll
  for (int j = 1; j < kRowMod; ++j) Rsum = Rsum + S[j];
  return Rsum;
}

// ------------------------------------------------------------------------------------------------
// CELL MODE (fused GRD cost; full-range tables on the coarse pyramid levels, range-restricted ones wherever a wave's lanes agree on a few
// disparities; level_rows decides per level pass).  At level s the 64 lanes of a wave share 64 >> s distinct centres,
// the window row they walk has only (64 >> s) + 2*half distinct columns, and the level has few disparities: the wave touches
// NQ x D distinct cells and ncent x n distinct guide weights on a window row, against 64 x n x 2 cell evaluations and 64 x n
// weight look-ups when every lane works for itself.  So the wave first builds, per window row,
//   cells[d][q]  = myCostGrd(own column q, other column q -+ d), d = 1 .. D           (the SAME grd_cell(): same bits)
//   wgts[c][j]   = exp(-|I_centre(c) - I(c + j)| / 10), 0 for a column outside the image (the same table)
// in LDS and then walks the window: a tap is its disparity, two 8-byte cell reads, one weight read, the interpolation and the
// accumulation -- 8 VALU instructions and 6 LDS cycles instead of 23 and 21.  Same terms, same order: identical results.
// The cell table is DMA-filled from the device-cell volume when there is one (same bits: k_grd_volume wrote them with grd_cell()'s
// arithmetic), else built as above; measured in DESIGN.md sections 5.1 and 7.
// ------------------------------------------------------------------------------------------------
struct CellRow {
  int adr_c;   // lane: LDS address of the table row of disparity 0 at the lane's window column 0 (a virtual row: f indexes it)
  int stride;  // bytes between consecutive disparities: NQ * 8
  int adr_w;   // lane: LDS address of wgts[centre of the lane][0]                                (WTAB)
  int adr_p;   // lane: LDS address of the own view's colour of the lane's window column 0        (!WTAB: guide weights on the fly)
  uint32_t Ip; // the centre's colour                                                             (!WTAB)
};
// WTAB: the guide weights come from the per-row table wgts[centre][column]; otherwise every tap forms its own (own colour from the
// strip set, |dI| -> exp table: :161-164) -- levels with too many centres for the table, rows inside the image only
// STRIDE > 0: the table's row pitch in bytes as a compile-time constant -- the second cell is then read off the first one's address with
// an immediate displacement (one address computation per tap instead of two); 0: the pitch is CellRow::stride
template <bool ALLV, bool WTAB, int J0, int J1, int STRIDE = 0>
__device__ __forceinline__ void cell_batch(const RowLevel &A, const Luts &lut, const CellRow &C, int adr_c, int adr_c1, int adr_w, int adr_p, int g8,
                                           double pa, double Gg, double S[kRowMod]) {
  constexpr int N = J1 - J0;
  double c0[N], c1[N], w[N], fr[N];
  bool valid[N];
  uint32_t pq[N];
  if constexpr (!WTAB) {
#pragma unroll
    for (int k = 0; k < N; ++k) pq[k] = lds_ld<uint32_t>(adr_p + (J0 + k) * 4);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int j = J0 + k;
    const double q_disp = tap_disp(pa, (double)j, Gg);
    const DispSplit d = ALLV ? split_disp_valid(q_disp) : split_disp(q_disp, A.Dm1, A.has_valid);
    fr[k] = d.fr;
    valid[k] = d.valid;
    int a0;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a0) : "v"(d.f), "s"(C.stride), "v"(adr_c));
    c0[k] = lds_ld<double>(a0 + j * 8);
    if constexpr (STRIDE > 0) {
      c1[k] = lds_ld_single(a0 + STRIDE + j * 8);
    } else {
      int a1;
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a1) : "v"(d.f), "s"(C.stride), "v"(adr_c1));
      c1[k] = lds_ld<double>(a1 + j * 8);
    }
    if constexpr (WTAB) w[k] = lds_ld_single(adr_w + j * 8);  // single reads: merged into ds_read2_b64 they run at half rate
  }
  if constexpr (!WTAB) {
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = lut.w[(int)__builtin_amdgcn_sad_u8(C.Ip, pq[k], 0u)];
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double tmp = lerp_cells(fr[k], c0[k], c1[k]);
    const double maxc = A.maxc;
    const double t = (ALLV || valid[k]) ? tmp : maxc;
    S[J0 + k] = __builtin_fma(w[k], t, S[J0 + k]);
  }
}
template <bool ALLV, bool WTAB, int CNT, int STRIDE = 0>
__device__ __forceinline__ void cell_group(const RowLevel &A, const Luts &lut, const CellRow &C, int g0, double pa, double rowterm, double &qxg_d,
                                           double S[kRowMod]) {
  const double Gg = group_disp(pa, qxg_d, rowterm);
  qxg_d += (double)kRowMod;
  const int g8 = g0 * 8, adr_c = C.adr_c + g8, adr_c1 = C.adr_c + C.stride + g8, adr_w = C.adr_w + g8, adr_p = C.adr_p + g0 * 4;
  constexpr int SUB = CSPM_CELL_SUB;
  cell_batch<ALLV, WTAB, 0, (CNT < SUB ? CNT : SUB), STRIDE>(A, lut, C, adr_c, adr_c1, adr_w, adr_p, g8, pa, Gg, S);
  if constexpr (CNT > SUB) cell_batch<ALLV, WTAB, SUB, CNT, STRIDE>(A, lut, C, adr_c, adr_c1, adr_w, adr_p, g8, pa, Gg, S);
}
// PITCH_IMM: the pitches level_rows chooses for a 35 x 35 window on 64-pixel segments (level 0 padded / unpadded, level 1 padded /
// unpadded, levels 2, 3, 4) run the groups of seven with the pitch as an immediate; anything else (other window sizes, ragged
// segments, computed tables) with CellRow::stride.
template <bool ALLV, bool WTAB, bool PITCH_IMM = false>
__device__ __forceinline__ double cell_row_taps(const RowLevel &A, const Luts &lut, const CellRow &C, double pa, double rowterm, double qx0_d) {
  double S[kRowMod];
#pragma unroll
  for (int j = 0; j < kRowMod; ++j) S[j] = 0.0;
  double qx_d = qx0_d;
  const int full = A.n / kRowMod * kRowMod;
  int g0 = 0;
  bool done = false;
  if constexpr (PITCH_IMM && CSPM_CELL_PITCH_IMM) {
    done = true;
    switch (C.stride) {
#if CSPM_CELL_BUMP
      // (the table and weight / colour bases stepped by hand and laundered, as in row_taps: two address additions per group instead of four)
#define CSPM_PITCH(P)                                                                    \
  case P * 8: {                                                                          \
    CellRow Cg = C;                                                                      \
    for (; g0 < full; g0 += kRowMod) {                                                   \
      cell_group<ALLV, WTAB, kRowMod, P * 8>(A, lut, Cg, 0, pa, rowterm, qx_d, S);       \
      Cg.adr_c += kRowMod * 8;                                                           \
      if constexpr (WTAB) { Cg.adr_w += kRowMod * 8; asm("" : "+v"(Cg.adr_c), "+v"(Cg.adr_w)); } \
      else { Cg.adr_p += kRowMod * 4; asm("" : "+v"(Cg.adr_c), "+v"(Cg.adr_p)); }       \
    }                                                                                    \
  } break;
#else
#define CSPM_PITCH(P) case P * 8: for (; g0 < full; g0 += kRowMod) cell_group<ALLV, WTAB, kRowMod, P * 8>(A, lut, C, g0, pa, rowterm, qx_d, S); break;
#endif
      CSPM_PITCH(128) CSPM_PITCH(100) CSPM_PITCH(80) CSPM_PITCH(68) CSPM_PITCH(52) CSPM_PITCH(44) CSPM_PITCH(40)
#undef CSPM_PITCH
      default: done = false; break;
    }
  }
  if (!done)
    for (; g0 < full; g0 += kRowMod) cell_group<ALLV, WTAB, kRowMod>(A, lut, C, g0, pa, rowterm, qx_d, S);
  switch (A.n - full) {
#define CSPM_TAIL(K) case K: cell_group<ALLV, WTAB, K>(A, lut, C, g0, pa, rowterm, qx_d, S); break;
    CSPM_TAIL(1) CSPM_TAIL(2) CSPM_TAIL(3) CSPM_TAIL(4) CSPM_TAIL(5) CSPM_TAIL(6)
#undef CSPM_TAIL
    default: break;
  }
  double Rsum = S[0];
#pragma unroll
  for (int j = 1; j < kRowMod; ++j) Rsum = Rsum + S[j];
  return Rsum;
}

// Per-wave description of the 64 evaluation centres of one pass (wave-uniform unless noted)
struct RowCtx {
#ifdef CSPM_ROW_STATS
  int stat_slot;
#endif
#ifdef CSPM_COUNT_ALIVE
  int tag;           // histogram group: 0 = refinement steps 0-3, 1 = refinement steps >= 4, 2 = everything else
#endif
  int y;             // all lanes evaluate in row y
  int lane;
  char *strip;       // this wave's other-view strip, `cap` slots of 16 bytes
  char *ostrip;      // this wave's own-view strip, `ocap` elements (16 bytes reserved each)
  int cap, ocap;
};
__device__ __forceinline__ RowCtx make_row_ctx(unsigned char *smem, int y, int cap, int ocap) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char *base = reinterpret_cast<char *>(smem + sizeof(LutMem)) + (size_t)wave * (size_t)wave_lds_bytes(cap, ocap);
  return RowCtx{
#ifdef CSPM_ROW_STATS
      0,
#endif
#ifdef CSPM_COUNT_ALIVE
      2,
#endif
      y, (int)(threadIdx.x & 63), base, base + (size_t)cap * 16, cap, ocap};
}

// One level of eval_rows for a wave: centres cx (per lane) in row cy (uniform), plane (a, b, c per lane) -> level sum.
// (Outlining it -- noinline, to make the caller park its long-lived state once per level -- was measured: 2.8x slower,
// the LDS pointers degrade to flat pointers and the Cost block to scratch memory across the call.)
#ifndef CSPM_LEVEL_INLINE
#define CSPM_LEVEL_INLINE __forceinline__
#endif
template <int SRC, int VIEW>
__device__ CSPM_LEVEL_INLINE double level_rows(const Cost &cd, const Luts &lut, const RowCtx &ctx, int s, int cx, int cy, double a, double b,
                                             double c, bool exit_on, double need, bool dead_in) {
  constexpr int E = elem_size<SRC>();
  const int lane = ctx.lane;
#ifdef CSPM_ROW_STATS
  unsigned long long rowtime_acc[4] = {0, 0, 0, 0};
  int rowtime_base = 0;
#endif
  const Level &L = cd.lv[s];
  RowLevel A;
  A.W = L.W; A.n = cd.n; A.half = cd.half; A.Dm1 = L.D - 1;
  A.has_valid = L.D >= 2;
  A.maxc = cd.max_cost[VIEW * CSPM_MAX_LEVELS + s];
  A.vol = L.vol[VIEW];
  A.slab = (size_t)L.W * (size_t)L.H;
  const char *px, *opx;
  if (SRC == kSrcCen) {
    px = reinterpret_cast<const char *>(L.pc[VIEW]); opx = reinterpret_cast<const char *>(L.pc[1 - VIEW]);
  } else {
    px = reinterpret_cast<const char *>(L.px[VIEW]); opx = reinterpret_cast<const char *>(L.px[1 - VIEW]);
  }
  // the wave's span of centres: decides the strip windows and whether any lane needs the column mask
  int cmin = wave_min_i32(cx), cmax = wave_max_i32(cx);
  cmin = __builtin_amdgcn_readfirstlane(cmin);
  cmax = __builtin_amdgcn_readfirstlane(cmax);
  const bool edge = (cmin - A.half < 0) | (cmax + A.half >= A.W);
  // strip windows in padded columns.  Other view: the left view reads x-f-1 .. x-1, the right view x+1 .. x+f+1 (f in [1, D-1]).
  const int D = L.D;
  const int s_lo = VIEW == 0 ? L.pad + cmin - A.half - D : L.pad + cmin - A.half;
  const int s_hi = VIEW == 0 ? L.pad + cmax + A.half : L.pad + cmax + A.half + D + 1;
  const int s_len = SRC == kSrcVolume ? 0 : s_hi - s_lo + 1;
  const int o_lo = L.pad + cmin - A.half, o_len = cmax - cmin + 2 * A.half + 1;
  const bool staged = s_len <= ctx.cap && o_len <= ctx.ocap;  // wave-uniform
  RowSrc R;
  const int strip_a = lds_addr(ctx.strip), ostrip_a = lds_addr(ctx.ostrip);
  R.adr_o = strip_a + (L.pad + cx - A.half - s_lo) * 16 - ((VIEW == 0 && SRC != kSrcImg) ? 16 : 0);
  R.adr_g = ostrip_a + (cx - cmin) * (SRC == kSrcCen ? 16 : 8);
  R.adr_p = ostrip_a + ctx.ocap * 8 + (cx - cmin) * 4;
  R.adr_g2 = R.adr_g3 = R.adr_g;
  asm volatile("" : "+v"(R.adr_g2));
  asm volatile("" : "+v"(R.adr_g3));
  R.lane_off = (L.pad + cx - A.half) * E;
  R.img_base = staged ? strip_a + (L.pad - s_lo) * 16 : L.pad * E;
  R.fx_lo = staged ? s_lo - L.pad : -L.pad;
  R.fx_hi = staged ? s_hi - 1 - L.pad : L.W + L.pad - 2;
  const uint32_t Ip = SRC == kSrcCen ? L.pc[VIEW][cy * L.Wp + L.pad + cx].pix : L.px[VIEW][cy * L.Wp + L.pad + cx].pix;
  const double qx0_d = (double)(cx - A.half);
  const int e_lo = max(0, A.half - cx);                           // first window column inside the image
  const int e_span = min(A.n - 1, A.W - 1 - cx + A.half) - e_lo;  // last one, relative
  RowTree tree;
  const int dy_lo = max(0, A.half - cy), dy_hi = min(A.n - 1, L.H - 1 - cy + A.half);
  for (int dy = 0; dy < dy_lo; ++dy) tree.push(dy, 0.0);
  // Row-granular early exit (result-preserving, only with the early-exit licence): `need` is what the level sum must reach for
  // the lane's total to be >= its threshold, with a 2^-40 margin over every rounding on the way (eval_rows_view).  Row totals
  // are >= 0, so a running sum of completed rows that has reached it proves the rejection; once that holds in all 64 lanes the
  // wave abandons the level -- a rejected candidate's cost is never stored.  Returns +inf then.
  double partial = 0.0;
  auto all_rejected = [&](double Rsum) -> bool {
#if CSPM_ROW_EXIT
    if (!exit_on) return false;
    partial = partial + Rsum;
    return __builtin_amdgcn_ballot_w64(!(dead_in | (partial >= need))) == 0ull;
#else
    return false;
#endif
  };
  if constexpr (SRC == kSrcGrd) {
    // Cell mode: the wave builds per window row a table of the cells it can touch (and, where it fits, of the guide weights) in
    // LDS, next to ONE compact strip set, and the taps read those.
    //   full  : every disparity of the level (d = 1 .. D) and the weight table -- the coarse levels, where that fits;
    //   range : the levels above them, when every tap of every lane interpolates (the four window corners of each lane lie in
    //           [1 + 2^-20, D - 2^-20]: the disparity is linear over the window up to two roundings of < 2^-43 each) and the wave's
    //           integer disparities span few values [f_lo, f_hi]: a table of just those ND = f_hi - f_lo + 1 disparities, strips
    //           of just the columns they need, the weight table when it still fits and per-tap weights when not.
    // Same grd_cell(), same terms, same order: identical results whichever path a level takes.
    const int ncent = cmax - cmin + 1, NQ = o_len;
    const int lds_room = wave_lds_bytes(ctx.cap, ctx.ocap) - 64;
    const int own_bytes = (NQ * 8 + 15) / 16 * 16 + (NQ * 4 + 15) / 16 * 16, wtab_bytes = ncent * A.n * 8 + ncent * 4;
    // Table pitch.  A cell read is 64 lanes x 8 bytes at (table row of the lane's disparity, the lane's column): a half-wave is
    // conflict-free when its 32 lanes hit 32 different bank pairs.  With ~64 centres (level 0) a half-wave holds 32 different columns:
    // a pitch of a multiple of 32 entries (256 B) puts every table row on the same banks and no two lanes collide whatever their
    // disparities (unpadded -- 98 entries, two bank pairs further per row -- a lane one disparity up collides with the lane two
    // columns on).  With ~32 centres (level 1) two lanes share each column and may differ in disparity: the rows must NOT line up;
    // a pitch = 16 mod 32 gives the 16 columns of a half-wave two disjoint halves of the banks for rows r and r+1.  Coarser levels
    // (few columns per half-wave) are conflict-free as they are.  Padded when the LDS has the room.
    const int NQP = ncent >= 48 ? (NQ + 31) / 32 * 32 : ncent >= 24 ? (NQ + 15) / 32 * 32 + 16 : NQ;
    int d_base = 1, ND = D, pitch = NQ;
    bool cells_on = false, wtab = true, allv_level = false, tdma = false;
    // two clusters of disparities (a wave across a depth discontinuity): lanes of cluster B use table rows nd_a .. with base b_lo
    int nd_a = 0, b_lo = 0;
    bool lane_b = false;
    int tbuf = 2;  // DMA-filled tables: 2 = the next row's table lands while this row's taps run; 1 = it is fetched after them
    const int p2 = (NQ * 4 + 15) / 16 * 16;  // a run of own colours, in 16-byte DMA pieces
    if (CSPM_CELL_MODE && staged && D >= 2) {
      const bool full_fits = (NQ + D) * 16 + own_bytes + NQ * D * 8 + wtab_bytes <= lds_room;
      // DMA-filled tables (Level::cvol): the device cells of this level are in memory, a table row is a run of a volume row
      const bool have_cvol = CSPM_TABLE_DMA && L.cvol[VIEW] != nullptr;
      // the range test, where it can matter: the full table does not fit, or the tables could be DMA-filled (two of them must fit)
      bool range_ok = false;
      int f_lo = 1, f_hi = 1;
      int fl_lane = 1, fh_lane = 1;  // the lane's own interval of integer disparities (valid when range_ok)
      if ((!full_fits || have_cvol) && CSPM_RANGE_MODE && D < 512 && dy_lo <= dy_hi) {
        const int jl = (A.n - 1) % kRowMod;
        const double rt0 = b * (double)(cy - A.half + dy_lo) + c, rt1 = b * (double)(cy - A.half + dy_hi) + c;  // q_disp_y of the first / last row
        const double q00 = tap_disp(a, 0.0, group_disp(a, qx0_d, rt0)), q01 = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rt0));
        const double q10 = tap_disp(a, 0.0, group_disp(a, qx0_d, rt1)), q11 = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rt1));
        const double qmin = __builtin_fmin(__builtin_fmin(q00, q01), __builtin_fmin(q10, q11));
        const double qmax = __builtin_fmax(__builtin_fmax(q00, q01), __builtin_fmax(q10, q11));
        const bool safe = (qmin >= 1.0 + 0x1p-20) & (qmax <= (double)D - 0x1p-20);  // false for NaN
        f_lo = fl_lane = safe ? (int)(qmin - 0x1p-20) : 1;
        f_hi = fh_lane = safe ? min((int)(qmax + 0x1p-20) + 1, D) : 1;  // qmax == D - 2^-20 exactly would name slab D + 1: no tap reads beyond slab D
        f_lo = wave_min_i32(f_lo);
        f_hi = wave_max_i32(f_hi);
        range_ok = __builtin_amdgcn_ballot_w64(!safe) == 0ull;
      }
      const int nd = f_hi - f_lo + 1;
      if (have_cvol) {
        // two tables (the DMA of row dy+1 lands while the taps of row dy read theirs), two runs of own colours, the weight table
        // where it fits; table rows of an even number of entries (16-byte pieces).  At most 8 DMA instructions per table.
        const int NQE = (NQ + 3) & ~3, NQD = NQP > NQE ? NQP : NQE;  // multiples of 4 entries: 16-byte pieces, and the pitches cell_row_taps knows as immediates
        // (a table's slabs are addressed by 32-bit DMA offsets from its first one: n_ slabs must span < 4 GiB -- the host only guarantees
        // that for 64 slabs, and a small window's table may hold more)
        auto span32 = [&](int n_) { return (unsigned long long)n_ * (unsigned long long)L.H * (unsigned long long)L.cvW * 8ull < (1ull << 32); };
        auto fits = [&](int nb, int n_, int pit, bool wt) { return nb * n_ * pit * 8 + 2 * p2 + (wt ? wtab_bytes : 0) <= lds_room && n_ * (pit / 2) <= 12 * kWave && span32(n_); };
        // in order of preference: two tables before one (with one, the fetch of the next row's table waits for this row's taps: its
        // latency is hidden by the other waves of the SIMD only), padded pitch before unpadded, the weight table before per-tap weights
        for (int nb = 2; nb >= 1 && !tdma; --nb) {
          if (range_ok) {
            if (!CSPM_PREFER_WTAB && CSPM_CELL_PAD && !edge && ncent <= CSPM_WTAB_MAXC && fits(nb, nd, NQD, false)) { tdma = true; pitch = NQD; wtab = false; }
            else if (!CSPM_PREFER_WTAB && !edge && ncent <= CSPM_WTAB_MAXC && fits(nb, nd, NQE, false)) { tdma = true; pitch = NQE; wtab = false; }
            else if (CSPM_CELL_PAD && fits(nb, nd, NQD, true)) { tdma = true; pitch = NQD; }
            else if (CSPM_CELL_PAD && !edge && fits(nb, nd, NQD, false)) { tdma = true; pitch = NQD; wtab = false; }
            else if (fits(nb, nd, NQE, true)) { tdma = true; pitch = NQE; }
            else if (!edge && fits(nb, nd, NQE, false)) { tdma = true; pitch = NQE; wtab = false; }
            if (tdma) { cells_on = true; d_base = f_lo; ND = nd; allv_level = true; tbuf = nb; }
          }
          if (!tdma && full_fits) {  // every disparity of a coarse level
            if (CSPM_CELL_PAD && fits(nb, D, NQD, true)) { tdma = true; pitch = NQD; }
            else if (fits(nb, D, NQE, true)) { tdma = true; pitch = NQE; }
            if (tdma) { cells_on = true; tbuf = nb; }
          }
          if (!CSPM_TABLE_DMA_SINGLE) break;
        }
        // Two surfaces in one wave (a depth discontinuity: 20-35 % of the level passes at levels 0-1 once the planes have settled):
        // the lanes' disparity intervals fall into two narrow clusters far apart.  Cut at the middle of the wave's span; when
        // no lane's interval straddles the cut, the table holds the rows of cluster A and then those of cluster B, and a lane
        // addresses the rows of its own cluster.
        if (CSPM_TABLE_CLUSTERS && !tdma && range_ok && nd > 2) {
          const int cut = (f_lo + f_hi + 1) / 2;
          const bool in_a = fh_lane < cut, in_b = fl_lane >= cut;
          if (__builtin_amdgcn_ballot_w64(!(in_a | in_b)) == 0ull) {
            int a_hi = in_a ? fh_lane : f_lo, bl = in_b ? fl_lane : f_hi;
            a_hi = wave_max_i32(a_hi);
            bl = wave_min_i32(bl);
            const int na = a_hi - f_lo + 1, nb_ = f_hi - bl + 1, nt = na + nb_;
            const bool span_ok = span32(f_hi - f_lo + 1);  // 32-bit DMA offsets: cluster B's rows are addressed from cluster A's first slab
            if (nt < nd && span_ok) {
              for (int nb = 2; nb >= 1 && !tdma; --nb) {
                if (CSPM_CELL_PAD && fits(nb, nt, NQD, true)) { tdma = true; pitch = NQD; }
                else if (CSPM_CELL_PAD && !edge && fits(nb, nt, NQD, false)) { tdma = true; pitch = NQD; wtab = false; }
                else if (fits(nb, nt, NQE, true)) { tdma = true; pitch = NQE; }
                else if (!edge && fits(nb, nt, NQE, false)) { tdma = true; pitch = NQE; wtab = false; }
                if (tdma) { cells_on = true; d_base = f_lo; ND = nt; allv_level = true; tbuf = nb; nd_a = na; b_lo = bl; lane_b = in_b; }
                if (!CSPM_TABLE_DMA_SINGLE) break;
              }
            }
          }
        }
      }
      if (!tdma) {
        if (full_fits) {
          cells_on = true;
          if (CSPM_CELL_PAD && (NQ + D) * 16 + own_bytes + NQP * D * 8 + wtab_bytes <= lds_room) pitch = NQP;
        } else if (range_ok && (NQ + nd) <= kStripRegs * kWave) {
          const int strip_bytes = (NQ + nd) * 16 + own_bytes;
          // in order of preference: padded pitch with the weight table, padded without, unpadded with, unpadded without
          if (CSPM_CELL_PAD && strip_bytes + NQP * nd * 8 + wtab_bytes <= lds_room) { cells_on = true; pitch = NQP; }
          else if (CSPM_CELL_PAD && !edge && strip_bytes + p2 + NQP * nd * 8 <= lds_room) { cells_on = true; wtab = false; pitch = NQP; }
          else if (strip_bytes + NQ * nd * 8 + wtab_bytes <= lds_room) { cells_on = true; }
          else if (!edge && strip_bytes + p2 + NQ * nd * 8 <= lds_room) { cells_on = true; wtab = false; }
          if (cells_on) { d_base = f_lo; ND = nd; allv_level = true; }
        }
      }
#ifdef CSPM_ROW_STATS
      if (range_ok && !cells_on && D <= 128) {  // what would a table of the UNION of the lanes' disparity intervals need?
        unsigned m[4] = {0u, 0u, 0u, 0u};
        for (int d = fl_lane; d <= fh_lane; ++d) m[(d >> 5) & 3] |= 1u << (d & 31);
        for (int off = 1; off < kWave; off <<= 1)
          for (int k = 0; k < 4; ++k) m[k] |= (unsigned)__shfl_xor((int)m[k], off, kWave);
        const int un = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
        if (lane == 0) atomicAdd(&g_unionstat[s * 8 + (un <= 8 ? 0 : un <= 11 ? 1 : un <= 16 ? 2 : un <= 24 ? 3 : 4)], 1ull);
      }
      if (lane == 0) {
        unsigned long long *g = &g_rangestat[(ctx.stat_slot * 8 + s) * 8];
        if (cells_on && !allv_level) atomicAdd(&g[7], 1ull);
        else {
          atomicAdd(&g[0], 1ull);
          if (!range_ok) atomicAdd(&g[1], 1ull);
          else if (cells_on) { atomicAdd(&g[wtab ? 2 : 3], 1ull); atomicAdd(&g[5], (unsigned long long)nd); }
          else { atomicAdd(&g[4], 1ull); atomicAdd(&g[6], (unsigned long long)nd); }
        }
      }
#endif
    }
    // wgts[c][j] = exp(-|I_centre(c) - I(c + j)| / 10), 0 for a column outside the image: lane = window column j, one centre per trip
    // (its colour is one broadcast read); the row's own colours at LDS address `adr_pix`, the centres' colours at `adr_ipc`
    auto build_wtab = [&](int adr_pix, int adr_ipc, int adr_wgt) {
      const int gcol0 = cmin - A.half;  // image column of table column q = 0
      for (int j0 = 0; j0 < A.n; j0 += kWave) {
        const int j = j0 + lane;
        const bool jon = j < A.n;
        const int js = jon ? j : 0;
        int adr_p = adr_pix + js * 4, adr_o = adr_wgt + js * 8;
        int col = gcol0 + js;
        constexpr int U = 4;
        for (int c0 = 0; c0 < ncent; c0 += U) {
          uint32_t ic[U], pq[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            ic[u] = lds_ld<uint32_t>(adr_ipc + (c0 + u) * 4);  // c0 + u < ncent + U: inside the wave's LDS, the value is not used beyond ncent
            pq[u] = lds_ld<uint32_t>(adr_p + u * 4);
          }
          double w[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            int sad = (int)__builtin_amdgcn_sad_u8(ic[u], pq[u], 0u);
            sad = ((unsigned)(col + u) < (unsigned)A.W) ? sad : kLutZero;  // outside the image: weight 0, the tap adds +0.0
            w[u] = lut.w[sad];
          }
          int ao = adr_o;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (jon && c0 + u < ncent) *(__attribute__((address_space(3))) double *)(uintptr_t)(unsigned)ao = w[u];
            ao += A.n * 8;
          }
          adr_p += U * 4; adr_o = ao; col += U;
        }
      }
    };
    if (cells_on && tdma) {
      // ---- tables filled by LDS-DMA from the level's device-cell volume.  LDS of the wave: table 0 | table 1 | colours 0 | colours 1
      // [| weights | centre colours].  A table is ND rows of `pitch` entries = pitch / 2 sixteen-byte pieces, NQE / 2 of them carrying
      // columns; piece e of the table (row e / ppr, piece e % ppr) is moved by lane e % 64 of DMA instruction e / 64.
      const int tbytes = ND * pitch * 8, ppr = pitch / 2, used = (NQ + 1) / 2, npiece = ND * ppr;
      const int off_p = tbuf * tbytes, off_w = off_p + 2 * p2, off_i = off_w + ncent * A.n * 8;
      const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(strip_a);
      const unsigned inv = ((1u << 20) + (unsigned)ppr - 1u) / (unsigned)ppr;  // e / ppr == (e * inv) >> 20 for e < 2048, ppr <= 64
      const size_t Wp = (size_t)L.Wp;
      const unsigned rowB = (unsigned)L.cvW * 8u, slabB = (unsigned)L.H * rowB;
      const char *cv = uniform_ptr(reinterpret_cast<const char *>(L.cvol[VIEW]) + (size_t)d_base * slabB + (size_t)(cy - A.half + dy_lo) * rowB +
                                   (size_t)(L.cvpad + cmin - A.half) * 8);
      const char *gp = uniform_ptr(reinterpret_cast<const char *>(L.pix[VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + o_lo) * 4);
      const int p_p16 = (NQ * 4 + 15) / 16;
      auto issue_tab = [&](int tb) {  // the table of the next row into table buffer tb
        const unsigned dst = lds0 + (unsigned)(tb * tbytes);
        for (int i = 0; i * kWave < npiece; ++i) {
          const unsigned e = (unsigned)(lane + i * kWave);
          const unsigned k = (e * inv) >> 20, pc = e - k * (unsigned)ppr;
          const unsigned dk = (nd_a > 0 && (int)k >= nd_a) ? k + (unsigned)(b_lo - d_base - nd_a) : k;  // table row k holds disparity d_base + dk
          if (e < (unsigned)npiece && pc < (unsigned)used) dma_b128(cv, dk * slabB + pc * 16u, dst + (unsigned)i * 1024u);
        }
        cv += rowB;
      };
      auto issue_pix = [&](int pb) {  // the own colours of the next row into colour run pb
        if (lane < p_p16) dma_b128(gp, (unsigned)lane * 16u, lds0 + (unsigned)(off_p + pb * p2));
        gp += Wp * 4;
      };
      dma_wait();  // the previous level's LDS reads have returned
      issue_pix(0);
      issue_tab(0);
      const int ipc_a = strip_a + off_i;
      if (wtab) *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(unsigned)(ipc_a + (cx - cmin) * 4) = Ip;
      CellRow C;
      C.stride = pitch * 8;
      C.adr_w = strip_a + off_w + (cx - cmin) * A.n * 8;
      C.Ip = Ip;
      // table row k holds disparity d_base + k (cluster B: b_lo + k - nd_a): f indexes row f - d_base (f - b_lo + nd_a)
      const int adr_c0 = strip_a + (cx - cmin) * 8 - (lane_b ? b_lo - nd_a : d_base) * C.stride;
      int par = 0;
      for (int dy = dy_lo; dy <= dy_hi; ++dy) {
        const int qy = cy - A.half + dy;
        [[maybe_unused]] const unsigned long long rt0 = ROWTIME_NOW();
        dma_wait();  // the table and the colours of row dy have landed; the taps of row dy-1 (the other table) have returned
        [[maybe_unused]] const unsigned long long rt1 = ROWTIME_NOW();
        if (dy < dy_hi) {
          issue_pix(par ^ 1);
          if (tbuf == 2) issue_tab(par ^ 1);
        }
        C.adr_c = adr_c0 + (tbuf == 2 ? par * tbytes : 0);
        C.adr_p = strip_a + off_p + par * p2 + (cx - cmin) * 4;
        if (wtab) build_wtab(strip_a + off_p + par * p2, ipc_a, strip_a + off_w);
        const double rowterm = b * (double)qy + c;  // q_disp_y, :155
        double Rsum;
        if (wtab) {
          bool allv = allv_level;
          if (!allv_level) {  // full tables: decide per row whether every tap interpolates
            const int jl = (A.n - 1) % kRowMod;
            const double q_first = tap_disp(a, 0.0, group_disp(a, qx0_d, rowterm));
            const double q_last = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rowterm));
            const double lo = 1.0 + 0x1p-20, hi = (double)L.D - 0x1p-20;
            const bool safe = (L.D < 512) & (q_first >= lo) & (q_first <= hi) & (q_last >= lo) & (q_last <= hi);
            allv = __builtin_amdgcn_ballot_w64(!safe) == 0ull;
          }
          Rsum = allv ? cell_row_taps<true, true, true>(A, lut, C, a, rowterm, qx0_d) : cell_row_taps<false, true, true>(A, lut, C, a, rowterm, qx0_d);
        } else {
          Rsum = cell_row_taps<true, false, true>(A, lut, C, a, rowterm, qx0_d);
        }
        tree.push(dy, Rsum);
#ifdef CSPM_ROW_STATS
        {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long rt3 = ROWTIME_NOW();
          ROWTIME_ADD(0, rt3 - rt0); ROWTIME_ADD(1, rt1 - rt0); ROWTIME_ADD(3, 1);
        }
#endif
        if (all_rejected(Rsum)) { dma_wait(); ROWTIME_FLUSH(); return __builtin_inf(); }
        if (tbuf == 1 && dy < dy_hi) {  // one table: the taps' reads have to be back before the next row's table may land on it
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          issue_tab(0);
        }
        par ^= 1;
      }
      ROWTIME_FLUSH();
      return tree.total(dy_hi + 1);
    }
    if (cells_on) {
      // strip window of the table's disparities (padded columns): slot(q, d) = q + (d_base + ND - 1) - d for the left view,
      // q + d - (d_base - 1) for the right view; NQ + ND slots
      const int c_lo = VIEW == 0 ? L.pad + cmin - A.half - (d_base + ND - 1) : L.pad + cmin - A.half + (d_base - 1);
      const int c_len = NQ + ND;
      // LDS of the wave: slots | own gradients | own colours [| own colours of the other row parity] | cells | weights | centre colours.
      // Per-tap weights (!wtab) read the own colours of the CURRENT row while the next row's strips arrive: that run is double-buffered.
      const int p_bytes = (NQ * 4 + 15) / 16 * 16;
      const int off_g = c_len * 16, off_p = off_g + (NQ * 8 + 15) / 16 * 16, off_c = off_p + (wtab ? 1 : 2) * p_bytes;  // 16-byte DMA pieces
      const int off_w = off_c + pitch * ND * 8, off_i = off_w + ncent * A.n * 8;
      const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(strip_a);
      const size_t Wp = (size_t)L.Wp;
      const char *g16 = uniform_ptr(reinterpret_cast<const char *>(L.px16[1 - VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + c_lo) * 16);
      const char *gg = uniform_ptr(reinterpret_cast<const char *>(L.grd[VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + o_lo) * 8);
      const char *gp = uniform_ptr(reinterpret_cast<const char *>(L.pix[VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + o_lo) * 4);
      const int g_p16 = (NQ * 8 + 15) / 16, p_p16 = (NQ * 4 + 15) / 16;  // 16-byte pieces of the own view's gradient / colour runs
      auto issue = [&](int par) {  // one compact strip set at lds0: slots | gradients | colours (of row parity `par`)
#pragma unroll
        for (int k = 0; k < kStripRegs; ++k)
          if (k * kWave < c_len) {
            if (lane + k * kWave < c_len) dma_b128(g16, (unsigned)(lane + k * kWave) * 16u, lds0 + (unsigned)k * 1024u);
          }
        if (lane < g_p16) dma_b128(gg, (unsigned)lane * 16u, lds0 + (unsigned)off_g);  // <= 64 pieces: the run has <= 128 columns
        if (lane < p_p16) dma_b128(gp, (unsigned)lane * 16u, lds0 + (unsigned)(off_p + par * p_bytes));
        g16 += Wp * 16; gg += Wp * 8; gp += Wp * 4;
      };
      dma_wait();  // the previous level's LDS reads have returned
      issue(0);
      int par = 0;
      // the centres' colours (window centre row cy, fixed for the level): lanes with the same centre write the same word
      const int ipc_a = strip_a + off_i;
      if (wtab) *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(unsigned)(ipc_a + (cx - cmin) * 4) = Ip;
      CellRow C;
      C.stride = pitch * 8;
      C.adr_c = strip_a + off_c + (cx - cmin) * 8 - d_base * C.stride;  // table row k holds disparity d_base + k: f indexes row f - d_base
      C.adr_w = strip_a + off_w + (cx - cmin) * A.n * 8;
      C.Ip = Ip;
      for (int dy = dy_lo; dy <= dy_hi; ++dy) {
        const int qy = cy - A.half + dy;
        [[maybe_unused]] const unsigned long long rt0 = ROWTIME_NOW();
        dma_wait();  // the strips of row dy have landed; the taps of row dy-1 have read their tables
        [[maybe_unused]] const unsigned long long rt1 = ROWTIME_NOW();
        // ---- cells[k][q], k = 0 .. ND-1 for disparities d_base .. d_base+ND-1: lane = table column q (its own element is read once),
        // disparities in batches of four (an entry is a chain of three dependent LDS round trips; the batch overlaps them).  The other
        // view's slot moves one slot per disparity and the table one row: every address in the batch is an immediate off two running bases.
        for (int q0 = 0; q0 < NQ; q0 += kWave) {  // one trip unless the row has more than 64 columns
          const int q = q0 + lane;
          const bool qon = q < NQ;
          const int qs = qon ? q : 0;
          const uint2 gq2 = lds_ld<uint2>(strip_a + off_g + qs * 8);
          const uint32_t pq = lds_ld<uint32_t>(strip_a + off_p + par * p_bytes + qs * 4);
          const double gq = __hiloint2double((int)gq2.y, (int)gq2.x);
          int adr_s = strip_a + (VIEW == 0 ? qs + ND - 1 : qs + 1) * 16;  // slot of disparity d_base
          int adr_t = strip_a + off_c + qs * 8;                             // cells[0][q]
          constexpr int U = 4;
          for (int d0 = 0; d0 < ND; d0 += U) {
            uint4 o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) o[u] = lds_ld<uint4>(adr_s + (VIEW == 0 ? -16 : 16) * u);  // beyond ND: inside the wave's LDS, the value is not stored
            double cell[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cell[u] = grd_cell(lut.a, pq, gq, o[u].z, __hiloint2double((int)o[u].y, (int)o[u].x));
            int at = adr_t;
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (qon && d0 + u < ND) *(__attribute__((address_space(3))) double *)(uintptr_t)(unsigned)at = cell[u];
              at += C.stride;
            }
            adr_s += (VIEW == 0 ? -16 : 16) * U;
            adr_t = at;
          }
        }
        if (wtab) build_wtab(strip_a + off_p, ipc_a, strip_a + off_w);
        const double rowterm = b * (double)qy + c;  // q_disp_y, :155
        double Rsum;
        dma_wait();  // the strip reads above have returned (and the table writes are queued behind them): the strips may go
        [[maybe_unused]] const unsigned long long rt2 = ROWTIME_NOW();
        C.adr_p = strip_a + off_p + par * p_bytes + (cx - cmin) * 4;  // this row's own colours (per-tap weights)
        if (!wtab) par ^= 1;
        if (dy < dy_hi) issue(par);
        if (wtab) {
          bool allv = allv_level;
          if (!allv_level) {  // full tables: decide per row whether every tap interpolates (see the general path below)
            const int jl = (A.n - 1) % kRowMod;
            const double q_first = tap_disp(a, 0.0, group_disp(a, qx0_d, rowterm));
            const double q_last = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rowterm));
            const double lo = 1.0 + 0x1p-20, hi = (double)L.D - 0x1p-20;
            const bool safe = (L.D < 512) & (q_first >= lo) & (q_first <= hi) & (q_last >= lo) & (q_last <= hi);
            allv = __builtin_amdgcn_ballot_w64(!safe) == 0ull;
          }
          Rsum = allv ? cell_row_taps<true, true>(A, lut, C, a, rowterm, qx0_d) : cell_row_taps<false, true>(A, lut, C, a, rowterm, qx0_d);
        } else {
          Rsum = cell_row_taps<true, false>(A, lut, C, a, rowterm, qx0_d);
        }
        tree.push(dy, Rsum);
#ifdef CSPM_ROW_STATS
        {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long rt3 = ROWTIME_NOW();
          ROWTIME_ADD(0, rt3 - rt0); ROWTIME_ADD(1, rt1 - rt0); ROWTIME_ADD(2, rt2 - rt1); ROWTIME_ADD(3, 1);
        }
#endif
        if (all_rejected(Rsum)) { dma_wait(); ROWTIME_FLUSH(); return __builtin_inf(); }
      }
      ROWTIME_FLUSH();
      return tree.total(dy_hi + 1);
    }
  }
  if constexpr (SRC == kSrcGrd) {
    if (staged) {
      // Fused GRD cells: the strips travel global -> LDS by DMA (Level::px16 holds the other view as ready-made strip slots,
      // Level::grd / Level::pix the own view's two arrays), double-buffered: the DMA for window row dy+1 is issued before the
      // taps of row dy and waited for after them.  No staging registers (30 VGPRs that used to be spilled around every row), no
      // ds_write (72 LDS-issue cycles per row), no exposed wait between fetch and commit.
      const int set = strip_set_bytes(ctx.cap, ctx.ocap);
      const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(strip_a);
      const size_t Wp = (size_t)L.Wp;
      const char *g16 = uniform_ptr(reinterpret_cast<const char *>(L.px16[1 - VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + s_lo) * 16);
      const char *gg = uniform_ptr(reinterpret_cast<const char *>(L.grd[VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + o_lo) * 8);
      const char *gp = uniform_ptr(reinterpret_cast<const char *>(L.pix[VIEW]) + ((size_t)(cy - A.half + dy_lo) * Wp + o_lo) * 4);
      // the own view's gradient and colour runs as 16-byte pieces (one DMA each; a piece may run up to 8 / 12 bytes past the run:
      // inside the padded rows in memory, inside the array's capacity -- ocap is even -- or the set's rounding in LDS)
      const int g_p16 = (o_len * 8 + 15) / 16, p_p16 = (o_len * 4 + 15) / 16;
      auto issue = [&](unsigned dst) {
#pragma unroll
        for (int k = 0; k < kStripRegs; ++k)
          if (k * kWave < s_len) {
            if (lane + k * kWave < s_len) dma_b128(g16, (unsigned)(lane + k * kWave) * 16u, dst + (unsigned)k * 1024u);
          }
        if (lane < g_p16) dma_b128(gg, (unsigned)lane * 16u, dst + (unsigned)(ctx.cap * 16));
        if (lane < p_p16) dma_b128(gp, (unsigned)lane * 16u, dst + (unsigned)(ctx.cap * 16 + ctx.ocap * 8));
        g16 += Wp * 16; gg += Wp * 8; gp += Wp * 4;  // the next image row
      };
      dma_wait();  // the previous level's strip reads have returned
      issue(lds0);
      int par = 0;
      for (int dy = dy_lo; dy <= dy_hi; ++dy) {
        const int qy = cy - A.half + dy;
        [[maybe_unused]] const unsigned long long rt0 = ROWTIME_NOW();
        dma_wait();  // row dy has landed; the reads of row dy-1 (the buffer the next DMA overwrites) have returned
        [[maybe_unused]] const unsigned long long rt1 = ROWTIME_NOW();
        if (dy < dy_hi) issue(lds0 + (unsigned)(par ? 0 : set));
        RowSrc Rr = R;
        const int boff = par ? set : 0;
        Rr.adr_o += boff; Rr.adr_g += boff; Rr.adr_g2 += boff; Rr.adr_g3 += boff; Rr.adr_p += boff; Rr.img_base += boff;
        const double rowterm = b * (double)qy + c;  // q_disp_y, :155
        double Rsum;
        // see the register-staged loop below for the all-valid test.  It looks at the window's end columns whether or not they are inside
        // the image: a wave at the image border (column mask) whose rows pass reads valid strip slots (the padded columns) for its masked
        // taps too, and takes the all-valid taps with the mask.
        const int jl = (A.n - 1) % kRowMod;
        const double q_first = tap_disp(a, 0.0, group_disp(a, qx0_d, rowterm));
        const double q_last = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rowterm));
        const double lo = 1.0 + 0x1p-20, hi = (double)L.D - 0x1p-20;
        const bool safe = (L.D >= 2) & (L.D < 512) & (q_first >= lo) & (q_first <= hi) & (q_last >= lo) & (q_last <= hi);
        const bool allv = __builtin_amdgcn_ballot_w64(!safe) == 0ull;
        if (!edge) {
          Rsum = allv ? row_taps<SRC, VIEW, false, true, true>(A, lut, Rr, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx)
                      : row_taps<SRC, VIEW, false, true>(A, lut, Rr, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
        } else {
          Rsum = (CSPM_EDGE_ALLV && allv) ? row_taps<SRC, VIEW, true, true, true>(A, lut, Rr, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx)
                                          : row_taps<SRC, VIEW, true, true>(A, lut, Rr, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
        }
        tree.push(dy, Rsum);
#ifdef CSPM_ROW_STATS
        {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long rt3 = ROWTIME_NOW();
          ROWTIME_ADD(4, rt3 - rt0); ROWTIME_ADD(5, rt1 - rt0); ROWTIME_ADD(7, 1);
        }
#endif
        if (all_rejected(Rsum)) { dma_wait(); ROWTIME_FLUSH(); return __builtin_inf(); }
        par ^= 1;
      }
      ROWTIME_FLUSH();
      return tree.total(dy_hi + 1);
    }
  }
  // The strips of window row dy+1 are fetched into registers while row dy is being evaluated and written to LDS afterwards:
  // the LDS queue of a wave is in order, so one buffer per strip suffices (reads of row dy precede the writes of row dy+1).
  // (Measured alternatives: a second other-view buffer written mid-row, so that the staging registers die early -- slower,
  // the allocator parks other values instead; an outlined level function -- 2.8x slower, see above.)
  StageReg<SRC> pre[kStripRegs];
  typename StripReg<SRC>::type opre[kOwnRegs];
  const size_t row_stride = (size_t)L.Wp * E;
  R.own_row = px + (size_t)(cy - A.half + dy_lo) * row_stride;
  R.oth_row = opx + (size_t)(cy - A.half + dy_lo) * row_stride;
  const int pre_off = (s_lo + lane) * E, opre_off = (o_lo + lane) * E;
  auto fetch = [&](const char *own_row, const char *oth_row) {
#pragma unroll
    for (int k = 0; k < kStripRegs; ++k)
      if (lane + k * kWave < s_len) pre[k] = ld_stage<SRC, VIEW>(oth_row, pre_off + k * kWave * E);
#pragma unroll
    for (int k = 0; k < kOwnRegs; ++k)
      if (lane + k * kWave < o_len) opre[k] = ld_strip<SRC>(own_row, opre_off + k * kWave * E);
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kStripRegs; ++k)
      if (lane + k * kWave < s_len) *reinterpret_cast<uint4 *>(ctx.strip + (lane + k * kWave) * 16) = pre[k].v;
#pragma unroll
    for (int k = 0; k < kOwnRegs; ++k)
      if (lane + k * kWave < o_len) wr_own<SRC>(ctx.ostrip, ctx.ocap, lane + k * kWave, opre[k]);
  };
  if (staged) {
    fetch(R.own_row, R.oth_row);
    wave_lds_fence();  // the previous level's strip reads are done
    commit();
  }
  for (int dy = dy_lo; dy <= dy_hi; ++dy) {
    const int qy = cy - A.half + dy;
    const bool more = dy < dy_hi;
    if (staged) {
      wave_lds_fence();
      if (more) fetch(R.own_row + row_stride, R.oth_row + row_stride);
    }
    const double rowterm = b * (double)qy + c;  // q_disp_y, :155
    double Rsum;
#ifdef CSPM_ROW_STATS
    {
      const int jl = (A.n - 1) % kRowMod;
      const double q_first = tap_disp(a, 0.0, group_disp(a, qx0_d, rowterm));
      const double q_last = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rowterm));
      const double lo = 1.0 + 0x1p-20, hi = (double)L.D - 0x1p-20;
      const bool safe = (L.D >= 2) & (q_first >= lo) & (q_first <= hi) & (q_last >= lo) & (q_last <= hi);
      int f_lo = safe ? (int)fmin(q_first, q_last) : 0, f_hi = safe ? (int)fmax(q_first, q_last) + 1 : 0;
      for (int off = 1; off < kWave; off <<= 1) {
        f_lo = min(f_lo, __shfl_xor(f_lo, off, kWave));
        f_hi = max(f_hi, __shfl_xor(f_hi, off, kWave));
      }
      const int nd = f_hi - f_lo + 1;
      int bucket = nd <= 4 ? 0 : nd <= 8 ? 1 : nd <= 16 ? 2 : nd <= 32 ? 3 : 4;
      if (__builtin_amdgcn_ballot_w64(!safe) != 0ull) bucket = 5;
      if (!staged) bucket = 6;
      if (lane == 0) atomicAdd(&g_rowstat[(ctx.stat_slot * 8 + s) * 8 + bucket], 1ull);
    }
#endif
    if (staged && !edge) {
      // The disparity along a window row is linear in the column (up to two roundings of < 2^-43 each while it is below 2^9): when
      // it is inside [1 + 2^-20, D - 2^-20] at both end columns, every tap of the row has static_cast<int>(q_disp) in [1, D-1] --
      // the interpolation branch of :166-175 -- and the taps need neither the clamp nor the test nor the select.  Decided per
      // row for the whole wave (true for nearly every row once the planes have settled; random planes take the general path).
      const int jl = (A.n - 1) % kRowMod;
      const double q_first = tap_disp(a, 0.0, group_disp(a, qx0_d, rowterm));
      const double q_last = tap_disp(a, (double)jl, group_disp(a, qx0_d + (double)(A.n - 1 - jl), rowterm));
      const double lo = 1.0 + 0x1p-20, hi = (double)L.D - 0x1p-20;
      const bool safe = (L.D >= 2) & (L.D < 512) & (q_first >= lo) & (q_first <= hi) & (q_last >= lo) & (q_last <= hi);
      Rsum = __builtin_amdgcn_ballot_w64(!safe) == 0ull ? row_taps<SRC, VIEW, false, true, true>(A, lut, R, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx)
                                                        : row_taps<SRC, VIEW, false, true>(A, lut, R, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
    } else if (staged) {
      Rsum = row_taps<SRC, VIEW, true, true>(A, lut, R, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
    } else {
      Rsum = row_taps<SRC, VIEW, true, false>(A, lut, R, Ip, a, rowterm, qx0_d, e_lo, e_span, qy, cx);
    }
    tree.push(dy, Rsum);
    if (all_rejected(Rsum)) return __builtin_inf();
    if (staged && more) {
      wave_lds_fence();
      commit();
    }
    R.own_row += row_stride;
    R.oth_row += row_stride;
  }
  return tree.total(dy_hi + 1);
}

#ifdef CSPM_COUNT_ALIVE
__device__ unsigned long long g_alive[16];  // debug: lanes still alive after level s, lanes evaluated at level s
// debug: level passes by the number of lanes that still carry a live candidate when the pass STARTS: [group][level][bucket], buckets
// 1-8, 9-16, 17-32, 33-48, 49-64 live lanes (a wave with no live lane never starts the pass)
__device__ unsigned long long g_alive_hist[3][8][5];
#endif

// A candidate plane as a lane holds it between uses: Plane::norm() and Plane::param()
struct RowPlane {
  double nx, ny, nz, a, b, c;
};

// Aggregated plane cost of 64 candidates, one per lane, each at its own centre column (per lane, inside the image) of row
// ctx.y of view VIEW.  Returns the cost per lane; lanes whose candidate is proven not to beat `thresh` (per lane) may return
// +inf instead (checked at level ends, only when use_thresh; the wave leaves early once every lane is rejected).
//
// The candidate is NOT passed in: `gen(x)` produces it -- from the random stream and the planes in global memory -- and is
// called again at every pyramid level.  Re-deriving six doubles costs ~10^2 instructions against the ~4*10^4 of a level,
// and it keeps the normal, the parameters and whatever they were derived from out of the registers while the window loop
// runs (the allocator would otherwise park them in scratch memory around every window row).  `x` is laundered through an
// empty asm at each level so that the re-derivation is not hoisted back out of the level loop.
template <bool CS, int SRC, int VIEW, class Gen>
__device__ __forceinline__ double eval_rows_view(const Cost &cd, const Luts &lut, const RowCtx &ctx, int x, Gen gen, double thresh,
                                                 bool use_thresh) {
  double cost = 0.0;
  bool dead = false;
  const int levels = CS ? cd.levels : 1;
  for (int s = 0; s < levels; ++s) {
#ifdef CSPM_COUNT_ALIVE
    if (use_thresh) {
      const int n_live = __popcll(__builtin_amdgcn_ballot_w64(!dead));
      const int bucket = n_live <= 8 ? 0 : n_live <= 16 ? 1 : n_live <= 32 ? 2 : n_live <= 48 ? 3 : 4;
      if (ctx.lane == 0) atomicAdd(&g_alive_hist[ctx.tag][s][bucket], 1ull);
    }
#endif
    int xs = x;
    asm volatile("" : "+v"(xs));
    const RowPlane p = gen(xs);
    double a = p.a, b = p.b, c = p.c;
    int cur_x = xs, cur_y = ctx.y;
    if (CS) {
      // Plane(org_norm, Point3d(cur_x,cur_y,cur_disp)).param() (:144-149) after s halvings (:183-185)
      double cur_disp = p.a * (double)xs + p.b * (double)ctx.y + p.c;  // pre_cs_pc.cc:139-140
      for (int k = 0; k < s; ++k) {
        cur_y /= 2;
        cur_x /= 2;
        cur_disp /= 2.0;
      }
      plane_param(p.nx, p.ny, p.nz, (double)cur_x, (double)cur_y, cur_disp, a, b, c);
    }
    // what this level's sum must reach to prove cost >= thresh: (thresh - cost so far) / weight, plus a 2^-40 margin that covers
    // the roundings of the division, of the row-tree sum against a running sum, of the product and of the final addition
    const double lw = CS ? cd.lv[s].wgt : 1.0;
    const double need = (thresh - cost) / lw * (1.0 + 0x1p-40);  // weight 0 (lambda = 0): +inf / NaN, never reached
    const double sc = level_rows<SRC, VIEW>(cd, lut, ctx, s, cur_x, cur_y, a, b, c, use_thresh, need, dead);
    if (use_thresh && __builtin_amdgcn_ballot_w64(!__builtin_isinf(sc)) == 0ull) { dead = true; break; }  // every lane proven rejected in mid-level (wave-uniform)
    if (CS) cost += sc * cd.lv[s].wgt;  // :182
    else cost = sc;
    if (use_thresh) {
      dead = dead | (cost >= thresh);
#ifdef CSPM_COUNT_ALIVE
      {
        const unsigned long long alive_mask = __builtin_amdgcn_ballot_w64(!dead);
        if (ctx.lane == 0) atomicAdd(&g_alive[s], (unsigned long long)__popcll(alive_mask));
        if (ctx.lane == 0) atomicAdd(&g_alive[8 + s], 64ull);
      }
#endif
      if (__builtin_amdgcn_ballot_w64(!dead) == 0ull) break;  // every lane is rejected
    }
  }
  return dead ? __builtin_inf() : cost;
}
template <bool CS, int SRC, class Gen>
__device__ __forceinline__ double eval_rows(const Cost &cd, const Luts &lut, const RowCtx &ctx, int view, int x, Gen gen, double thresh,
                                            bool use_thresh) {
  return view == 0 ? eval_rows_view<CS, SRC, 0>(cd, lut, ctx, x, gen, thresh, use_thresh)
                   : eval_rows_view<CS, SRC, 1>(cd, lut, ctx, x, gen, thresh, use_thresh);
}

// Pixel run of a wave: item = (view, y, 64-pixel segment).  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MiB
// L2; blockIdx % 8 -- an assumption that only locality and balance rest on).  Two ways of handing out the items, chosen per launch:
//
//  * CLAIMED COLUMN BANDS (launches of two or more rounds of resident waves: the row kernels of a KITTI-size or larger pair).  The items in (segment, view, row) order are cut into eight equal runs; a wave takes the next item of the
//    run of its XCD from that run's counter and, when the run is exhausted, of the next run that still has items.  An XCD thus works
//    through ~segs/8 adjacent 64-column segments of every row of both views: its L2 holds a (3 x 64 + window + disparity range)-
//    column band of the level images instead of seeing all of them (k_refine on C3: L2 hit rate 64 % -> 95 %, fabric traffic / 8,
//    L1 return path -22 %), consecutive claims are vertically adjacent rows, and the XCDs whose bands are cheaper (no image border)
//    finish the others' bands -- the launch has 25 % more workgroups than items for that; the surplus finds every run empty and leaves.
//  * INTERLEAVED ROW BLOCKS (launches that fit the GPU in less than two rounds, where little can be re-balanced while it runs): the
//    image rows (both views stacked) are cut into blocks of kRowBand rows, XCD k works through blocks k, k+8, k+16, ... and a
//    workgroup takes x-adjacent segments of one row: every XCD and every CU gets the same mix of image top, middle, bottom and
//    border columns (border rows have clipped windows and are cheaper, border columns carry the column mask and are dearer;
//    contiguous eighths of the image left half of the XCDs idle at the end -- measured, CSPM_ROW_BAND).
#ifndef CSPM_ROW_BAND
#define CSPM_ROW_BAND 4
#endif
constexpr int kRowBand = CSPM_ROW_BAND;
struct RowItem {
  int v, y, x0;
};
struct RowQueue {
  unsigned int *next;  // claimed column bands: [8] items claimed from run k so far, zeroed before the launch; null: interleaved row blocks
};
__host__ __device__ inline long long row_items(int W, int H, int views) { return (long long)((W + kWave - 1) / kWave) * views * H; }
__host__ __device__ inline long long row_items_per_xcd(int W, int H, int views, bool claimed) {
  const int segs = (W + kWave - 1) / kWave;
  if (claimed) return (row_items(W, H, views) + 7) / 8;
  const int nblk = (views * H + kRowBand - 1) / kRowBand;
  return (long long)((nblk + 7) / 8) * kRowBand * segs;
}
__device__ __forceinline__ bool row_item(int W, int H, int views, const RowQueue &rq, RowItem &it) {
  const int segs = (W + kWave - 1) / kWave;
  const int xcd = (int)(blockIdx.x % 8u);
  if (rq.next) {
    const long long total = row_items(W, H, views), per = row_items_per_xcd(W, H, views, true);
    long long g = -1;
    if ((threadIdx.x & 63) == 0) {
      for (int t = 0; t < 8 && g < 0; ++t) {
        const int q = (xcd + t) & 7;
        const long long lo = (long long)q * per;
        const long long cnt = lo + per <= total ? per : total - lo;
        if (cnt <= 0) continue;
        if ((long long)__hip_atomic_load(&rq.next[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) continue;
        const long long i = (long long)__hip_atomic_fetch_add(&rq.next[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i < cnt) g = lo + i;
      }
    }
    const int gi = __builtin_amdgcn_readfirstlane((int)g);  // < 2^31 items
    if (gi < 0) return false;
    const int seg = gi / (views * H);
    const int rem = gi - seg * (views * H);
    it.v = rem / H;
    it.y = rem - it.v * H;
    it.x0 = seg * kWave;
    return true;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long e = (long long)(blockIdx.x / 8u) * kRowWaves + wave;  // index among this XCD's waves
  if (e >= row_items_per_xcd(W, H, views, false)) return false;
  const int per_blk = kRowBand * segs;
  const int blk_local = (int)(e / per_blk);
  const int rem = (int)(e - (long long)blk_local * per_blk);
  const int row_in_blk = rem / segs, seg = rem - row_in_blk * segs;
  const int row = (blk_local * 8 + xcd) * kRowBand + row_in_blk;  // row of the stacked views
  if (row >= views * H) return false;
  it.v = row / H;
  it.y = row - it.v * H;
  it.x0 = seg * kWave;
  return true;
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::InitRandomPlane  (cs_patchmatch.cc:115-148)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ RowPlane init_plane(const Pm &pm, int v, int x, int y) {
  const long long i = (long long)y * pm.W + x;
  const Rng rng(pm.seed, stream_id(0, 0, 0, v), pm.rng_row_shared ? (uint64_t)x : (uint64_t)i);
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
  RowPlane p;
  p.nx = r0 * inv; p.ny = r1 * inv; p.nz = r2 * inv;
  plane_param(p.nx, p.ny, p.nz, (double)x, (double)y, rand_dis, p.a, p.b, p.c);  // :141-142
  return p;
}

template <bool CS, int SRC>
__global__ __launch_bounds__(kRowBlock, CSPM_INIT_MINW) void k_init(Cost cd, Pm pm, RowQueue rq, int cap, int ocap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LutMem &s_lut = *reinterpret_cast<LutMem *>(smem);
  const Luts lut = load_luts(cd, s_lut);
  RowItem it;
  if (!row_item(pm.W, pm.H, 2, rq, it)) return;
  const int lane = threadIdx.x & 63;
  RowCtx ctx = make_row_ctx(smem, it.y, cap, ocap);
  const bool live = it.x0 + lane < pm.W;
  const int x = live ? it.x0 + lane : pm.W - 1;  // tail lanes shadow the last pixel
  auto gen = [&](int xs) { return init_plane(pm, it.v, xs, it.y); };
  const double cost = eval_rows<CS, SRC>(cd, lut, ctx, it.v, x, gen, kDoubleMax, false);  // :143-144
  if (live) {
    const RowPlane p = gen(x);
    store_plane(pm.f[it.v], (long long)it.y * pm.W + x, p.nx, p.ny, p.nz, p.a, p.b, p.c, cost);
  }
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::PlaneRefinement  (cs_patchmatch.cc:292-345): several (by default all) halving steps of one iteration in
// one launch.  A pixel's steps depend only on that pixel's own earlier steps; its current plane lives in the plane field
// (global memory) and is re-read where it is needed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ RowPlane refine_plane(const Pm &pm, int v, int x, int y, int iter, int step, double z_iter, double n_iter) {
  const long long i = (long long)y * pm.W + x;
  const Field &f = pm.f[v];
  const double cnx = f.nx[i], cny = f.ny[i], cnz = f.nz[i], ca = f.a[i], cb = f.b[i], cc = f.c[i];
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
  RowPlane p;
  p.nx = d0 * inv; p.ny = d1 * inv; p.nz = d2 * inv;
  plane_param(p.nx, p.ny, p.nz, (double)x, (double)y, pz, p.a, p.b, p.c);    // :330
  return p;
}

template <bool CS, int SRC>
__global__ __launch_bounds__(kRowBlock, CSPM_ROW_MINW) void k_refine(Cost cd, Pm pm, RowQueue rq, int iter, int first_step, int nsteps, double z_iter, double n_iter, int cap, int ocap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LutMem &s_lut = *reinterpret_cast<LutMem *>(smem);
  const Luts lut = load_luts(cd, s_lut);
  RowItem it;
  if (!row_item(pm.W, pm.H, 2, rq, it)) return;
  const int lane = threadIdx.x & 63;
  RowCtx ctx = make_row_ctx(smem, it.y, cap, ocap);
  const bool live = it.x0 + lane < pm.W;
  const int x = live ? it.x0 + lane : pm.W - 1;
  const long long i = (long long)it.y * pm.W + x;
  const Field &f = pm.f[it.v];
  double cur_min = f.cost[i];
  const bool use_thresh = pm.use_thresh != 0 && *cd.early_ok != 0;
  for (int step = first_step; step < first_step + nsteps; ++step) {
#ifdef CSPM_ROW_STATS
    ctx.stat_slot = 2 + step;
#endif
#ifdef CSPM_COUNT_ALIVE
    ctx.tag = step < 4 ? 0 : 1;
#endif
    auto gen = [&](int xs) { return refine_plane(pm, it.v, xs, it.y, iter, step, z_iter, n_iter); };
    const double cost = eval_rows<CS, SRC>(cd, lut, ctx, it.v, x, gen, cur_min, use_thresh);
    if (cost < cur_min) {                                                      // :335-338
      const RowPlane p = gen(x);  // before the store below changes what it is derived from
      cur_min = cost;
      if (live) store_plane(f, i, p.nx, p.ny, p.nz, p.a, p.b, p.c, cost);
    }
    z_iter /= 2.0;  // :342-343
    n_iter /= 2.0;
  }
}

// ------------------------------------------------------------------------------------------------
// CSPatchMatch::ViewPropagation  (cs_patchmatch.cc:229-277), target view v.
// Phase 1 (k_view_eval): every pixel (x,y) of the OTHER view proposes its plane to pixel (cor_x,y) of
// view v and evaluates it there -- independent, because the pass only reads the other view's planes
// and the candidates of a pass do not depend on each other.  Lane = source pixel; its evaluation centre
// cor_x = x +- disparity moves with the disparity field: coherent where the field is smooth (the strips
// still cover the wave), anything else falls back to global gathers for that level.
// Phase 2 (k_view_resolve): the serial loop keeps, per target pixel, the candidate with the smallest
// cost that is < the pixel's current cost, the earliest in traversal order among equal costs.  One
// workgroup per row (cor_x stays in row y) reproduces exactly that with LDS atomics.
// ------------------------------------------------------------------------------------------------
struct ViewCand {
  double *cost; // candidate cost, +inf = rejected / none
  double *c;    // candidate param c (a, b follow from the source normal)
  int *cx;      // target column
  int *perm;    // per row: the source columns ordered by TARGET column (k_view_sort); null: lanes take consecutive source columns
};

// the proposal of source pixel (x,y) of view 1-v: target column (may be outside the image) and the plane anchored there
struct ViewProposal {
  RowPlane p;
  int cor_x;
};
__device__ __forceinline__ ViewProposal view_proposal(const Pm &pm, int v, int x, int y) {
  const long long i = (long long)y * pm.W + x;
  const Field &src = pm.f[1 - v];
  ViewProposal q;
  q.p.nx = src.nx[i]; q.p.ny = src.ny[i]; q.p.nz = src.nz[i];
  double disp = src.a[i] * (double)x + src.b[i] * (double)y + src.c[i];  // :245-246
  if (disp < 0.0) disp = 0.0;                                             // :247-252
  if (disp >= (double)pm.max_dis) disp = (double)pm.max_dis - 1.0;
  const int r = round2int(disp);
  q.cor_x = handle_border(v == 0 ? x + r : x - r, pm.W);                  // :255-261
  const bool inside = q.cor_x >= 0 && q.cor_x < pm.W;
  plane_param(q.p.nx, q.p.ny, q.p.nz, (double)(inside ? q.cor_x : x), (double)y, disp, q.p.a, q.p.b, q.p.c);  // :263-265
  return q;
}

// Round 6 -- lanes TARGET-adjacent instead of source-adjacent.  A wave of 64 consecutive SOURCE pixels whose disparities straddle a depth
// edge proposes into two stretches of the target row that lie up to max_dis apart; its strips cannot cover both and the whole wave
// falls back to global gathers (15-20 % of the level-0 window rows of a KITTI-size pair).  The resolve rule does not care who evaluated
// a proposal (k_view_resolve: smallest cost, earliest traversal rank among equals), so the proposals of a row are handed to the lanes
// in the order of their TARGET column: one workgroup per row counts the proposals per target column in LDS, scans, and writes the
// permutation (proposals without a target inside the image last); wave k of the row takes slots [64k, 64k + 64).  Equal targets keep
// no particular order -- every proposal is evaluated exactly once either way, by the same function of (source pixel, target pixel).
// Measured (C3, profiles/r06_viewsort/): unstaged level-0 window rows -25 %, k_view_eval 2.08 -> 1.87 ms per launch.  Cutting the
// order into runs whose targets span at most 64 columns (every run staged, no unstaged row left) was built too and is SLOWER, 2.35 ms:
// it needs 10-15 % more waves (a run ends at every occlusion gap) and a wave's time is set by how many different disparities its lanes
// carry, not by whether its strips were staged.
__global__ __launch_bounds__(256) void k_view_sort(Pm pm, int v, ViewCand vc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int *s_cnt = reinterpret_cast<unsigned int *>(smem);  // W + 1 keys (key W: no target)
  const int W = pm.W, y = blockIdx.x, nk = W + 1;
  unsigned int *s_part = s_cnt + nk;                              // 256 partial sums
  const Field &src = pm.f[1 - v];
  const long long row = (long long)y * W;
  for (int t = threadIdx.x; t < nk; t += blockDim.x) s_cnt[t] = 0u;
  __syncthreads();
  auto key_of = [&](int x) {
    const long long i = row + x;
    double disp = src.a[i] * (double)x + src.b[i] * (double)y + src.c[i];  // as view_proposal
    if (disp < 0.0) disp = 0.0;
    if (disp >= (double)pm.max_dis) disp = (double)pm.max_dis - 1.0;
    const int r = round2int(disp);
    const int cx = handle_border(v == 0 ? x + r : x - r, W);
    return cx >= 0 && cx < W ? cx : W;
  };
  for (int x = threadIdx.x; x < W; x += blockDim.x) atomicAdd(&s_cnt[key_of(x)], 1u);
  __syncthreads();
  // exclusive scan of the W + 1 counts: every thread scans a contiguous run, thread 0 scans the 256 run totals
  const int per = (nk + (int)blockDim.x - 1) / (int)blockDim.x;
  const int k0 = min((int)threadIdx.x * per, nk), k1 = min(k0 + per, nk);
  unsigned int run = 0u;
  for (int k = k0; k < k1; ++k) { const unsigned int n = s_cnt[k]; s_cnt[k] = run; run += n; }
  s_part[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int acc = 0u;
    for (int t = 0; t < (int)blockDim.x; ++t) { const unsigned int n = s_part[t]; s_part[t] = acc; acc += n; }
  }
  __syncthreads();
  for (int k = k0; k < k1; ++k) s_cnt[k] += s_part[threadIdx.x];
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += blockDim.x) vc.perm[row + atomicAdd(&s_cnt[key_of(x)], 1u)] = x;
}

template <bool CS, int SRC>
__global__ __launch_bounds__(kRowBlock, CSPM_VIEW_MINW) void k_view_eval(Cost cd, Pm pm, RowQueue rq, int v, ViewCand vc, int cap, int ocap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LutMem &s_lut = *reinterpret_cast<LutMem *>(smem);
  const Luts lut = load_luts(cd, s_lut);
  RowItem it;
  if (!row_item(pm.W, pm.H, 1, rq, it)) return;
  const int lane = threadIdx.x & 63;
  const int y = it.y;
  const bool live = it.x0 + lane < pm.W;
  const int slot = live ? it.x0 + lane : pm.W - 1;
  const int x = vc.perm ? vc.perm[(long long)y * pm.W + slot] : slot;  // the slot's source column (tail lanes shadow the row's last proposal)
  RowCtx ctx = make_row_ctx(smem, it.y, cap, ocap);
#ifdef CSPM_ROW_STATS
  ctx.stat_slot = 1;
#endif
  const long long i = (long long)y * pm.W + x;
  const ViewProposal q0 = view_proposal(pm, v, x, y);
  const bool inside = q0.cor_x >= 0 && q0.cor_x < pm.W;
  const int ex = inside ? q0.cor_x : x;  // lanes without a target evaluate in place and discard the result
  const bool use_thresh = pm.use_thresh != 0 && *cd.early_ok != 0;
  const double thr = use_thresh ? pm.f[v].cost[(long long)y * pm.W + ex] : kDoubleMax;
  const int dx = ex - x;  // the generator is handed the (laundered) evaluation column; the source column is dx to its left
  auto gen = [&](int exs) { return view_proposal(pm, v, exs - dx, y).p; };
  double cost = eval_rows<CS, SRC>(cd, lut, ctx, v, ex, gen, thr, use_thresh);  // :266-267
  if (!inside) cost = __builtin_inf();
  if (live) {
    vc.cost[i] = cost;
    vc.c[i] = view_proposal(pm, v, x, y).p.c;
    vc.cx[i] = q0.cor_x;
  }
}

}  // namespace cspm
