// cspm_kernels.h -- the HIP kernels of the PatchMatch-stereo hot path (gfx950, wave64).
//
// Plane costs (IPlaneCost::GetPlaneCost; PreSSPC: pre_ss_pc.cc:74-118, PreCSPC: pre_cs_pc.cc:133-188) are evaluated by two
// engines that add the same terms in the same order (cspm_tap.h):
//   cspm_rows.h   lanes = pixels: k_init, k_refine, k_view_eval (InitRandomPlane, PlaneRefinement, ViewPropagation)
//   cspm_chain.h  lanes = window taps of one pixel: k_spatial_sweep / k_spatial_diag / k_spatial_rb (SpatialPropagation),
//                 k_cost_batch (cspm_plane_cost_batch)
// This file holds everything else: the view-propagation resolver, PlaneToDisp, post-processing, image preparation and
// the cost-volume kernels.
//
// Cell costs come from one of three sources, selected at compile time:
//   SRC = kSrcGrd: GRD cell cost computed on the fly from the padded images + gradients (bit-identical
//                  to reading the device-cell volume: myCostGrd of cc/grd_cc.cpp:4-35 with its last multiply-add
//                  contracted, cspm_tap.h); nothing but ~40 B/pixel per view and level is ever read, so the
//                  working set stays in L2 / Infinity Cache.
//   SRC = kSrcCen: census / Hamming cell cost computed on the fly from 80-bit codes (cc/cen_cc.cc:47-66).
//   SRC = kSrcVolume: cost volumes in HBM (any CCMethod plugin; what the reference's PreSSPC/PreCSPC do).
#pragma once
#include "cspm_device.h"

// minimum waves per SIMD the register allocator leaves room for in the sweep kernel (2nd __launch_bounds__ argument).  A sweep workgroup
// is five waves (one per pyramid level), which the dispatcher places 2+1+1+1 on the CU's four SIMDs -- and the next workgroup the same
// way: TWO resident workgroups per CU need FOUR wave slots on the first SIMD, i.e. <= 128 VGPRs (measured in round 5: at 137-155 VGPRs
// only one workgroup per CU is resident and a sweep takes 38 ms instead of 20; it is also why 3 or 4 workgroups per CU never differed
// from 2: at 95 VGPRs = 5 slots per SIMD the third workgroup would need a sixth).
#ifndef CSPM_SWEEP_MINW
#define CSPM_SWEEP_MINW 4
#endif

#include "cspm_chain.h"
#include "cspm_rows.h"
#include "cspm_tap.h"
#include "cspm_foreign.h"

#pragma clang fp contract(off)

namespace cspm {

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
// CSPatchMatch::PostProcessing (cs_patchmatch.cc:508-588) on the 8-bit maps, both views per launch.
//   k_lr_check          one lane per pixel and view: the consistency flag
//   k_fill_rows         one workgroup per image row and view: nearest consistent pixel on either side by a two-level scan
//                       (lane-private runs, then the 256 run summaries), inconsistent pixels get the smaller of the two
//                       planes' disparities and are appended to the view's work list
//   k_weighted_median   one WAVEFRONT per listed pixel: the 35 columns of a window row sit in 35 lanes, the 256-bin histogram
//                       in 4 registers of each lane (bin = 64*k + lane), the additions are replayed in window order so every
//                       bin and the running total round as the reference's scalar loop does
// ------------------------------------------------------------------------------------------------
// LeftRightCheck (:347-369): a pixel is consistent when the other view, at the column its rounded disparity points to,
// holds a disparity within half a pixel, and its own disparity is positive
__global__ void k_lr_check(const uint8_t *__restrict__ dis0, const uint8_t *__restrict__ dis1, int W, int H, int dis_scale,
                           uint8_t *__restrict__ ok0, uint8_t *__restrict__ ok1) {
  const long long n = (long long)W * H;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const int v = i >= n ? 1 : 0;
  i -= v * n;
  const uint8_t *mine = v ? dis1 : dis0, *theirs = v ? dis0 : dis1;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const double d = mine[i] * 1.0 / dis_scale;
  const int ox = x + (2 * v - 1) * round2int(d);
  bool ok = false;
  if (ox >= 0 && ox < W) ok = fabs(d - theirs[(size_t)y * W + ox] * 1.0 / dis_scale) <= 0.5 && d > 0.0;
  (v ? ok1 : ok0)[i] = ok ? 1 : 0;
}

__device__ __forceinline__ uint8_t sat_u8(int q) { return (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
__device__ __forceinline__ double plane_disp_at(const Field &f, long long j, int x, int y) {
  double d = f.a[j] * (double)x;  // param().dot(Vec3d(x, y, 1.0))
  d += f.b[j] * (double)y;
  d += f.c[j] * 1.0;
  return d;
}

constexpr int kFillBlock = 256;
// dynamic LDS of k_fill_rows for an image of width W: flags, nearest-left table, run summaries
inline size_t fill_rows_shmem(int W) { return (size_t)((W + 3) & ~3) + sizeof(int) * ((size_t)W + 2 * kFillBlock + 1); }

// FillInvalid (:370-428).  Grid = 2*H workgroups (view-major).  todo[v*n ...] receives the row-major indices of the view's
// inconsistent pixels (rows in no particular order: the median treats them independently), todo_cnt[v] their number.
__global__ __launch_bounds__(kFillBlock) void k_fill_rows(Pm pm, int dis_scale, const uint8_t *__restrict__ ok0, const uint8_t *__restrict__ ok1,
                                                          uint8_t *__restrict__ dis0, uint8_t *__restrict__ dis1,
                                                          unsigned int *__restrict__ todo, unsigned int *__restrict__ todo_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fill_smem[];
  const int W = pm.W, H = pm.H, t = (int)threadIdx.x;
  const int v = (int)blockIdx.x / H, y = (int)blockIdx.x - v * H;
  const long long row = (long long)y * W, n = (long long)W * H;
  uint8_t *flag = fill_smem;
  int *near_l = reinterpret_cast<int *>(fill_smem + ((W + 3) & ~3));
  int *run_l = near_l + W, *run_r = run_l + kFillBlock, *list_base = run_r + kFillBlock;
  const uint8_t *ok = (v ? ok1 : ok0) + row;
  uint8_t *dis = (v ? dis1 : dis0) + row;
  for (int x = t; x < W; x += kFillBlock) flag[x] = ok[x];  // coalesced
  __syncthreads();
  // lane-private run [x0, x1): last consistent column in it, first consistent column in it, number of inconsistent ones
  const int per = (W + kFillBlock - 1) / kFillBlock, x0 = min(W, t * per), x1 = min(W, x0 + per);
  int last = -1, first = W, holes = 0;
  for (int x = x0; x < x1; ++x) {
    if (flag[x]) { last = x; if (first == W) first = x; }
    else ++holes;
  }
  run_l[t] = last;
  run_r[t] = first;
  __syncthreads();
  // inclusive max-scan of the run summaries towards the right, min-scan towards the left (Hillis-Steele over 256 entries)
  for (int step = 1; step < kFillBlock; step <<= 1) {
    const int a = t >= step ? run_l[t - step] : -1, b = t + step < kFillBlock ? run_r[t + step] : W;
    __syncthreads();
    run_l[t] = max(run_l[t], a);
    run_r[t] = min(run_r[t], b);
    __syncthreads();
  }
  int carry = t > 0 ? run_l[t - 1] : -1;
  const int carry_r = t + 1 < kFillBlock ? run_r[t + 1] : W;
  for (int x = x0; x < x1; ++x) {
    if (flag[x]) carry = x;
    near_l[x] = carry;
  }
  __syncthreads();
  // where this run's inconsistent pixels go in the view's work list: exclusive sum of `holes`, one reservation per row
  run_l[t] = holes;
  __syncthreads();
  for (int step = 1; step < kFillBlock; step <<= 1) {
    const int a = t >= step ? run_l[t - step] : 0;
    __syncthreads();
    run_l[t] += a;
    __syncthreads();
  }
  if (t == kFillBlock - 1) *list_base = (int)atomicAdd(&todo_cnt[v], (unsigned int)run_l[t]);
  __syncthreads();
  unsigned int *out = todo + (size_t)v * n + (unsigned int)*list_base + (unsigned int)(run_l[t] - holes);
  const Field &f = pm.f[v];
  carry = carry_r;
  for (int x = x1 - 1; x >= x0; --x) {
    if (flag[x]) { carry = x; continue; }
    const int l = near_l[x], r = carry;
    if (l >= 0 && r < W) {
      const double dl = plane_disp_at(f, row + l, x, y), dr = plane_disp_at(f, row + r, x, y);
      dis[x] = sat_u8(dis_scale * round2int(dl <= dr ? dl : dr));
    } else if (l >= 0) {
      dis[x] = sat_u8(dis_scale * round2int(plane_disp_at(f, row + l, x, y)));
    } else if (r < W) {
      dis[x] = sat_u8(dis_scale * round2int(plane_disp_at(f, row + r, x, y)));
    }
    *out++ = (unsigned int)(row + x);
  }
}

__device__ __forceinline__ double lane_value(double v, int src_lane) {  // src_lane wave-uniform
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)u, src_lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(u >> 32), src_lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

constexpr int kMedianBlock = 256;
constexpr int kMedianRows = 6;  // window rows whose loads are in flight together
// WeightedMedian(valid, 35, WMF_GAMMA) (:430-506): inconsistent pixels only, consistent neighbours only.  2*half_wnd+1 <= 64.
// The 256 bins live in the wave's 2 KB of LDS; the additions are replayed in window order (same-bin additions are a dependent
// chain whatever the layout, and so is the running total), the lanes' parallel work is the gather that feeds the replay.
__global__ __launch_bounds__(kMedianBlock) void k_weighted_median(const uint32_t *__restrict__ pix0, const uint32_t *__restrict__ pix1, int Wp, int pad,
                                                                  int W, int H, const uint8_t *__restrict__ ok0, const uint8_t *__restrict__ ok1,
                                                                  const double *__restrict__ lut, uint8_t *__restrict__ dis0, uint8_t *__restrict__ dis1,
                                                                  const unsigned int *__restrict__ todo, const unsigned int *__restrict__ todo_cnt,
                                                                  int half_wnd) {
  __shared__ double s_hist[kMedianBlock / kWave][256];
  const int lane = (int)(threadIdx.x & 63);
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double *hist = s_hist[wave_in_block];
  const unsigned int wave = blockIdx.x * (kMedianBlock / kWave) + (unsigned int)wave_in_block;
  const unsigned int nwaves = gridDim.x * (kMedianBlock / kWave);
  const long long n = (long long)W * H;
  for (int v = 0; v < 2; ++v) {
    const uint32_t *pix = v ? pix1 : pix0;
    const uint8_t *ok = v ? ok1 : ok0;
    uint8_t *dis = v ? dis1 : dis0;
    const unsigned int cnt = todo_cnt[v];
    for (unsigned int k = wave; k < cnt; k += nwaves) {
      const unsigned int i = todo[(size_t)v * n + k];
      const int y = (int)(i / (unsigned int)W), x = (int)(i - (unsigned int)y * (unsigned int)W);
      const uint32_t centre = pix[(size_t)y * Wp + pad + x];
#pragma unroll
      for (int j = 0; j < 4; ++j) hist[lane + 64 * j] = 0.0;
      double total = 0.0, open_val = 0.0;  // wave-uniform
      int open_bin = -1;
      const int qx = x - half_wnd + lane;
      const bool col_in = lane <= 2 * half_wnd && qx >= 0 && qx < W;
      const int y_lo = max(0, y - half_wnd), y_hi = min(H - 1, y + half_wnd);
      for (int qy0 = y_lo; qy0 <= y_hi; qy0 += kMedianRows) {
        int bin[kMedianRows];
        double wgt[kMedianRows];
        bool use[kMedianRows];
#pragma unroll
        for (int r = 0; r < kMedianRows; ++r) {  // independent gathers of up to kMedianRows window rows
          const int qy = min(qy0 + r, y_hi);
          const size_t q = (size_t)qy * W + (col_in ? qx : x);
          use[r] = col_in && qy0 + r <= y_hi && ok[q] != 0;
          bin[r] = dis[q];
          wgt[r] = lut[__builtin_amdgcn_sad_u8(centre, pix[(size_t)qy * Wp + pad + (col_in ? qx : x)], 0u)];
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < kMedianRows; ++r) {
          // replay the row's additions in column order
          for (unsigned long long pending = __builtin_amdgcn_ballot_w64(use[r]); pending; pending &= pending - 1) {
            const int src = __builtin_ctzll(pending);
            const int b = __builtin_amdgcn_readlane(bin[r], src);
            const double w = lane_value(wgt[r], src);
            if (b != open_bin) {  // neighbours mostly share a disparity: the open bin stays in a register until another one is hit
              if (lane == 0 && open_bin >= 0) hist[open_bin] = open_val;
              open_val = hist[b];
              open_bin = b;
            }
            open_val += w;
            total += w;
          }
        }
      }
      if (lane == 0 && open_bin >= 0) hist[open_bin] = open_val;
      wave_lds_fence();
      const double half_total = total / 2.0;
      if (half_total > 0.0) {  // else no consistent neighbour: the filled value stays
        double run = 0.0;
        int median = 0;
        bool found = false;
        for (int part = 0; part < 4 && !found; ++part) {
          const double hp = hist[64 * part + lane];
          for (int l = 0; l < 64; ++l) {
            run += lane_value(hp, l);
            if (run >= half_total) { median = 64 * part + l; found = true; break; }
          }
        }
        if (lane == 0) dis[i] = (uint8_t)median;
      }
      wave_lds_fence();  // the bins are cleared for the next pixel only after they have been read
    }
  }
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
// the packed 8-byte elements of the raster sweep (Level::px8, cspm_device.h Pix8); `bad` counts gradients that are not 36-bit multiples of
// 2^-27 -- impossible for the gradient of an 8-bit image (cspm_device.h), checked all the same
__global__ void k_make_px8(const uint32_t *__restrict__ pix, const double *__restrict__ grd, long long n, Pix8 *__restrict__ out, unsigned int *__restrict__ bad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Pix8 e;
  if (!pix8_encode(pix[i], grd[i], &e)) atomicAdd(bad, 1u);
  out[i] = e;
}
// GRD strip slots of image v as the other view (Level::px16): {gradient, colour, colour of column x + dir}
__global__ void k_make_px16(const uint32_t *__restrict__ pix, const double *__restrict__ grd, int Wp, int H, int dir, uint4 *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Wp * H) return;
  const int xp = (int)(i % Wp);
  const int xn = min(max(xp + dir, 0), Wp - 1);
  const double g = grd[i];
  out[i] = uint4{(uint32_t)__double2loint(g), (uint32_t)__double2hiint(g), pix[i], pix[i - xp + xn]};
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
// DEV = false: the reference's arithmetic to the last bit (the CCMethod::buildCV boundary, cspm_grd_build_cv_host).
// DEV = true : the cells of the DEVICE order (cspm_tap.h grd_cell): the final multiply-add is one fma.  These are the cells the
//              plane cost reads -- recomputed inside the tap engines by default, materialised with CSPM_OPT_GRD_VOLUMES -- and
//              whose max is max_cost.
template <class Src, bool DEV>
__global__ __launch_bounds__(256) void k_grd_volume(Src l, Src r, const double *__restrict__ lG, const double *__restrict__ rG,
                                                    int gWp, int gpad, int W, int H, int d0, int nd, int right_view,
                                                    double *__restrict__ vol, unsigned long long *max_key, double2 *__restrict__ vol2 = nullptr,
                                                    double *__restrict__ cvol = nullptr, int cvW = 0, int cvpad = 0) {
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
    cost = DEV ? __builtin_fma(1 - 0.1, grdDiff, 0.1 * clrDiff) : 0.1 * clrDiff + (1 - 0.1) * grdDiff;  // ALPHA
    if (vol) vol[i] = cost;
    if (cvol) cvol[((size_t)d * H + y) * cvW + cvpad + x] = cost;  // the padded volume the row engine's tables are DMA-filled from
    if (vol2) {  // the sweep's paired cells (kSrcVol2): slab d holds {cell(d), cell(d+1)}, d = 0 .. nd-2 (d0 == 0)
      if (d + 1 < nd) vol2[i].x = cost;
      if (d >= 1) vol2[i - slab].y = cost;
    }
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
// cvtColor(8UC3, CV_BGR2GRAY) of OpenCV 2.4 (grd_pc.cc:37, cspc.cc:55): the same fixed-point contract, straight from the bytes
__global__ void k_gray8_u8(const uint32_t *__restrict__ pix, int W, int H, int Wp, int pad, uint8_t *__restrict__ gray) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)W * H) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const uint32_t q = pix[(size_t)y * Wp + pad + x];
  gray[i] = (uint8_t)(((int)(q & 255u) * 1868 + (int)((q >> 8) & 255u) * 9617 + (int)((q >> 16) & 255u) * 4899 + (1 << 13)) >> 14);
}
// GrdPC / CSPC elements (kSrcImg): g = Sobel(gray8, CV_64F, 1, 0, 1) = gray[x+1] - gray[x-1], REFLECT_101 (grd_pc.cc:40, cspc.cc:58);
// the pad cells repeat the image periodically, which is what HandleBorder (commfunc.h:129-145) makes of a column outside the image
__global__ void k_make_aos_img(const uint32_t *__restrict__ pix, const uint8_t *__restrict__ gray, int W, int H, int Wp, int pad,
                               PixG *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Wp * H) return;
  const int y = (int)(i / Wp);
  int x = ((int)(i - (long long)y * Wp) - pad) % W;
  if (x < 0) x += W;
  PixG e;
  e.pix = pix[(size_t)y * Wp + pad + x];
  e.g = (double)((int)gray[(size_t)y * W + reflect101(x + 1, W)] - (int)gray[(size_t)y * W + reflect101(x - 1, W)]);
  out[i] = e;
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
