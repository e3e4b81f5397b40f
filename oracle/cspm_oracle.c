/*
 * cspm_oracle.c -- CPU restatement of the CrossScalePatchMatch hot path.  See cspm_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (reference needs OpenCV+gflags, absent here).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off: the reference is built for SSE2 (CSPM.vcxproj:163-188), i.e. no FMA; every
 * product and sum below is individually rounded, and the HIP kernels are compiled the same way.
 *
 * Citations are relative to /root/reference/CSPM/.
 */
#include "cspm_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K_DOUBLE_EPS 0.00000001 /* commfunc.h:26 */
#define K_DOUBLE_MAX DBL_MAX    /* commfunc.h:27 */
#define WGT_GAMMA 10.0          /* pre_cs_pc.h:16 */
#define WMF_GAMMA 10.0          /* cs_patchmatch.h:14 */
#define K_MAX_NORM 1.0          /* cs_patchmatch.h:145 */
#define K_Z_STOP 0.1            /* cs_patchmatch.h:146 */
/* cc/grd_cc.h:6-9 */
#define BORDER_THRES 3
#define TAU_CLR 10.0
#define TAU_GRD 2.0
#define ALPHA 0.1

/* ------------------------------------------------------------------ primitives */

/* commfunc.h:117-121  magic-number rounding: round-half-to-even to int32 */
int csor_round2int(double d) {
  d = d + 6755399441055744.0;
  int32_t lo;
  memcpy(&lo, &d, sizeof lo); /* little endian: low word of the mantissa */
  return lo;
}

/* commfunc.h:129-145  single wrap-around */
int csor_handle_border(int loc, int size) {
  if (loc < 0) return loc + size;
  if (loc >= size) return loc - size;
  return loc;
}

/* plane.h:25-34  Plane::update_param.  norm_.dot(point_) follows cv::Matx::dot: s=0; s+=a[i]*b[i]. */
void csor_plane_param(const double n[3], const double p[3], double prm[3]) {
  double denom = fmax(fabs(n[2]), K_DOUBLE_EPS);
  if (n[2] < 0.0) denom = -denom;
  prm[0] = -n[0] / denom;
  prm[1] = -n[1] / denom;
  double s = n[0] * p[0];
  s += n[1] * p[1];
  s += n[2] * p[2];
  prm[2] = s / denom;
}

/* pre_cs_pc.cc:111-114, pre_ss_pc.cc:60-64, cs_patchmatch.cc:434-437 */
void csor_exp_lut(double *lut, double gamma) {
  for (int i = 0; i < 1000; ++i) lut[i] = exp(-i * 1.0 / gamma);
}

/* pre_cs_pc.cc:86-109: regMat tridiagonal, scale_wgt[s] = inv(regMat)(0,s).
 * Mat::inv() default DECOMP_LU, restated from OpenCV 2.4 LUImpl (partial pivoting, reciprocal
 * pivots, identity right-hand side). */
int csor_scale_weights(int S, double lambda, double *w) {
  if (S < 1 || S > 16) return -1;
  double A[16 * 16], B[16 * 16];
  memset(A, 0, sizeof A);
  memset(B, 0, sizeof B);
  for (int s = 0; s < S; ++s) {
    B[s * S + s] = 1.0;
    if (S == 1) { A[0] = 1 + lambda; break; } /* reference would index out of range; 1x1 case */
    if (s == 0) {
      A[s * S + s] = 1 + lambda;
      A[s * S + s + 1] = -lambda;
    } else if (s == S - 1) {
      A[s * S + s] = 1 + lambda;
      A[s * S + s - 1] = -lambda;
    } else {
      A[s * S + s] = 1 + 2 * lambda;
      A[s * S + s - 1] = -lambda;
      A[s * S + s + 1] = -lambda;
    }
  }
  /* cv::invert(DECOMP_LU) of OpenCV 2.4 (lapack.cpp) does not run LU for n <= 3: it uses det2 / det3 and the
   * cofactors times 1/det.  Restated from memory of those sources (not verifiable here); row 0 only. */
  if (S == 1) { w[0] = 1. / A[0]; return 0; }
  if (S == 2) {
    double d = A[0] * A[3] - A[1] * A[2];
    if (d == 0.) return -2;
    d = 1. / d;
    w[0] = A[3] * d;
    w[1] = -A[1] * d;
    return 0;
  }
  if (S == 3) {
    double d = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (d == 0.) return -2;
    d = 1. / d;
    w[0] = (A[4] * A[8] - A[5] * A[7]) * d;
    w[1] = (A[2] * A[7] - A[1] * A[8]) * d;
    w[2] = (A[1] * A[5] - A[2] * A[4]) * d;
    return 0;
  }
  const double eps = DBL_EPSILON * 100;
  for (int i = 0; i < S; ++i) {
    int k = i;
    for (int j = i + 1; j < S; ++j)
      if (fabs(A[j * S + i]) > fabs(A[k * S + i])) k = j;
    if (fabs(A[k * S + i]) < eps) return -2;
    if (k != i) {
      for (int j = i; j < S; ++j) { double t = A[i * S + j]; A[i * S + j] = A[k * S + j]; A[k * S + j] = t; }
      for (int j = 0; j < S; ++j) { double t = B[i * S + j]; B[i * S + j] = B[k * S + j]; B[k * S + j] = t; }
    }
    double d = -1 / A[i * S + i];
    for (int j = i + 1; j < S; ++j) {
      double alpha = A[j * S + i] * d;
      for (k = i + 1; k < S; ++k) A[j * S + k] += alpha * A[i * S + k];
      for (k = 0; k < S; ++k) B[j * S + k] += alpha * B[i * S + k];
    }
    A[i * S + i] = -d;
  }
  for (int i = S - 1; i >= 0; --i)
    for (int j = 0; j < S; ++j) {
      double s = B[i * S + j];
      for (int k = i + 1; k < S; ++k) s -= A[i * S + k] * B[k * S + j];
      B[i * S + j] = s * A[i * S + i];
    }
  for (int s = 0; s < S; ++s) w[s] = B[0 * S + s];
  return 0;
}

/* ---- counter-based RNG shared (by specification, not by code) with the HIP kernels ----
 * The reference draws from cv::RNG seeded with time(NULL) (cs_patchmatch.cc:32,130,309); that
 * stream is neither reproducible nor available here, so both sides use this generator. */
#define GOLD 0x9E3779B97F4A7C15ULL
static inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}
uint32_t csor_stream_id(int phase, int iter, int step, int view) {
  return (uint32_t)((((phase * 16 + iter) * 32 + step) * 2) + view);
}
uint64_t csor_rng_u64(uint64_t seed, uint32_t stream, uint64_t pix, uint32_t draw) {
  uint64_t k = mix64(seed + GOLD * ((uint64_t)stream + 1));
  uint64_t b = mix64(k ^ (GOLD * (pix + 1)));
  return mix64(b + GOLD * ((uint64_t)draw + 1));
}
double csor_rng_u01(uint64_t seed, uint32_t stream, uint64_t pix, uint32_t draw) {
  return (double)(csor_rng_u64(seed, stream, pix, draw) >> 11) * (1.0 / 9007199254740992.0);
}
/* cv::RNG::uniform(double a,double b) = u*(b-a)+a */
static inline double uni(double u, double a, double b) { return u * (b - a) + a; }

int csor_refine_steps(int max_dis) { /* cs_patchmatch.cc:95,299-301,342 */
  int k = 0;
  for (double z = max_dis / 2.0; z >= K_Z_STOP; z /= 2.0) ++k;
  return k;
}

/* ------------------------------------------------------------------ pyramid */

static int reflect101(int p, int len) { /* cv::borderInterpolate(BORDER_REFLECT_101) */
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

/* OpenCV 2.4 pyrDown, 8U: separable [1 4 6 4 1], int accumulate, (v+128)>>8, REFLECT_101,
 * dst = ((w+1)/2, (h+1)/2).  Called at pre_cs_pc.cc:45. */
void csor_pyrdown_bgr8(const uint8_t *src, int w, int h, uint8_t *dst) {
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  int *rows = (int *)malloc(sizeof(int) * 5 * dw * 3);
  for (int y = 0; y < dh; ++y) {
    for (int k = 0; k < 5; ++k) {
      const int sy = reflect101(2 * y + k - 2, h);
      const uint8_t *s = src + (size_t)sy * w * 3;
      int *row = rows + k * dw * 3;
      for (int x = 0; x < dw; ++x) {
        const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x < w ? 2 * x : reflect101(2 * x, w),
                  x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
        for (int c = 0; c < 3; ++c)
          row[x * 3 + c] = s[x2 * 3 + c] * 6 + (s[x1 * 3 + c] + s[x3 * 3 + c]) * 4 + s[x0 * 3 + c] + s[x4 * 3 + c];
      }
    }
    uint8_t *d = dst + (size_t)y * dw * 3;
    const int *r0 = rows, *r1 = rows + dw * 3, *r2 = rows + 2 * dw * 3, *r3 = rows + 3 * dw * 3, *r4 = rows + 4 * dw * 3;
    for (int i = 0; i < dw * 3; ++i) d[i] = (uint8_t)((r2[i] * 6 + (r1[i] + r3[i]) * 4 + r0[i] + r4[i] + 128) >> 8);
  }
  free(rows);
}

/* ------------------------------------------------------------------ GRD cost computation */

/* grd_cc.cpp:70-73: convertTo(CV_32F) then cvtColor(CV_RGB2GRAY) on 32F:
 * gray = R*0.299f + G*0.587f + B*0.114f evaluated in float, left to right (OpenCV 2.4 RGB2Gray<float>). */
void csor_rgb2gray_f32(const double *rgb, int w, int h, float *gray) {
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    const float r = (float)rgb[i * 3 + 0], g = (float)rgb[i * 3 + 1], b = (float)rgb[i * 3 + 2];
    float t = r * 0.299f;
    t = t + g * 0.587f;
    t = t + b * 0.114f;
    gray[i] = t;
  }
}

/* grd_cc.cpp:76-77: Sobel(gray32F, CV_64F, dx=1, dy=0, ksize=1): kernel [-1 0 1], no smoothing,
 * BORDER_REFLECT_101 (=> 0 at both image borders), arithmetic in double. */
void csor_sobel_x_ks1(const float *gray, int w, int h, double *grd) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      grd[(size_t)y * w + x] = (double)gray[(size_t)y * w + xp] - (double)gray[(size_t)y * w + xm];
    }
}

/* grd_cc.cpp:4-19 */
static inline double cost_grd(const double *lC, const double *rC, const double *lG, const double *rG) {
  double clrDiff = 0;
  for (int c = 0; c < 3; c++) {
    double temp = fabs(lC[c] - rC[c]);
    clrDiff += temp;
  }
  clrDiff *= 0.3333333333;
  double grdDiff = fabs(lG[0] - rG[0]);
  clrDiff = clrDiff > TAU_CLR ? TAU_CLR : clrDiff;
  grdDiff = grdDiff > TAU_GRD ? TAU_GRD : grdDiff;
  return ALPHA * clrDiff + (1 - ALPHA) * grdDiff;
}
/* grd_cc.cpp:21-35  border variant: the other view is replaced by the constant BORDER_THRES */
static inline double cost_grd_border(const double *lC, const double *lG) {
  double clrDiff = 0;
  for (int c = 0; c < 3; c++) {
    double temp = fabs(lC[c] - BORDER_THRES);
    clrDiff += temp;
  }
  clrDiff *= 0.3333333333;
  double grdDiff = fabs(lG[0] - BORDER_THRES);
  clrDiff = clrDiff > TAU_CLR ? TAU_CLR : clrDiff;
  grdDiff = grdDiff > TAU_GRD ? TAU_GRD : grdDiff;
  return ALPHA * clrDiff + (1 - ALPHA) * grdDiff;
}

/* The same cell in the DEVICE order: identical terms, the final multiply-add contracted into one
 * fma -- one rounding fewer than grd_cc.cpp:18.  Only CSOR_SUM_DEVICE evaluations read these cells. */
static inline double cost_grd_dev(const double *lC, const double *rC, double lG, double rG) {
  double clrDiff = 0;
  for (int c = 0; c < 3; c++) clrDiff += fabs(lC[c] - rC[c]);
  clrDiff *= 0.3333333333;
  double grdDiff = fabs(lG - rG);
  clrDiff = clrDiff > TAU_CLR ? TAU_CLR : clrDiff;
  grdDiff = grdDiff > TAU_GRD ? TAU_GRD : grdDiff;
  return fma(1 - ALPHA, grdDiff, ALPHA * clrDiff);
}

static void grd_prepare(const double *l, const double *r, int w, int h, double **lG, double **rG) {
  float *g = (float *)malloc(sizeof(float) * (size_t)w * h);
  *lG = (double *)malloc(sizeof(double) * (size_t)w * h);
  *rG = (double *)malloc(sizeof(double) * (size_t)w * h);
  csor_rgb2gray_f32(l, w, h, g);
  csor_sobel_x_ks1(g, w, h, *lG);
  csor_rgb2gray_f32(r, w, h, g);
  csor_sobel_x_ks1(g, w, h, *rG);
  free(g);
}

/* grd_cc.cpp:60-109  GrdCC::buildCV */
void csor_grd_build_cv(const double *l, const double *r, int w, int h, int maxDis, double *vol) {
  double *lG, *rG;
  grd_prepare(l, r, w, h, &lG, &rG);
  for (int d = 0; d < maxDis; d++)
    for (int y = 0; y < h; y++) {
      const double *lData = l + (size_t)y * w * 3, *rData = r + (size_t)y * w * 3;
      const double *lGData = lG + (size_t)y * w, *rGData = rG + (size_t)y * w;
      double *cost = vol + ((size_t)d * h + y) * w;
      for (int x = 0; x < w; x++) {
        if (x - d >= 0) cost[x] = cost_grd(lData + 3 * x, rData + 3 * (x - d), lGData + x, rGData + x - d);
        else cost[x] = cost_grd_border(lData + 3 * x, lGData + x);
      }
    }
  free(lG);
  free(rG);
}

/* grd_cc.cpp:110-154  GrdCC::buildRightCV */
void csor_grd_build_right_cv(const double *l, const double *r, int w, int h, int maxDis, double *vol) {
  double *lG, *rG;
  grd_prepare(l, r, w, h, &lG, &rG);
  for (int d = 0; d < maxDis; d++)
    for (int y = 0; y < h; y++) {
      const double *lData = l + (size_t)y * w * 3, *rData = r + (size_t)y * w * 3;
      const double *lGData = lG + (size_t)y * w, *rGData = rG + (size_t)y * w;
      double *cost = vol + ((size_t)d * h + y) * w;
      for (int x = 0; x < w; x++) {
        if (x + d < w) cost[x] = cost_grd(lData + 3 * (x + d), rData + 3 * x, lGData + x + d, rGData + x);
        else cost[x] = cost_grd_border(rData + 3 * x, rGData + x);
      }
    }
  free(lG);
  free(rG);
}

/* both views' GRD volumes in the device order (cost_grd_dev); border cells as grd_cc.cpp:88-100,134-147 */
static void grd_build_dev(const double *l, const double *r, int w, int h, int maxDis, int right, double *vol) {
  double *lG, *rG;
  grd_prepare(l, r, w, h, &lG, &rG);
  const double bc[3] = {BORDER_THRES, BORDER_THRES, BORDER_THRES};
  for (int d = 0; d < maxDis; d++)
    for (int y = 0; y < h; y++) {
      const double *lData = l + (size_t)y * w * 3, *rData = r + (size_t)y * w * 3;
      const double *lGData = lG + (size_t)y * w, *rGData = rG + (size_t)y * w;
      double *cost = vol + ((size_t)d * h + y) * w;
      for (int x = 0; x < w; x++) {
        if (!right) {
          if (x - d >= 0) cost[x] = cost_grd_dev(lData + 3 * x, rData + 3 * (x - d), lGData[x], rGData[x - d]);
          else cost[x] = cost_grd_dev(lData + 3 * x, bc, lGData[x], BORDER_THRES);
        } else {
          if (x + d < w) cost[x] = cost_grd_dev(lData + 3 * (x + d), rData + 3 * x, lGData[x + d], rGData[x]);
          else cost[x] = cost_grd_dev(rData + 3 * x, bc, rGData[x], BORDER_THRES);
        }
      }
    }
  free(lG);
  free(rG);
}

/* ------------------------------------------------------------------ census cost computation */

/* cen_cc.cc:13-16: convertTo(CV_8U) (saturate_cast<uchar>(cvRound(v))) then cvtColor(CV_RGB2GRAY) on 8U =
 * OpenCV 2.4 fixed point: (R*4899 + G*9617 + B*1868 + (1<<13)) >> 14. */
static void rgb64_to_gray8(const double *rgb, int w, int h, uint8_t *gray) {
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    int c[3];
    for (int k = 0; k < 3; ++k) {
      double v = rgb[i * 3 + k];
      int q = (int)nearbyint(v); /* cvRound: round half to even */
      c[k] = q < 0 ? 0 : (q > 255 ? 255 : q);
    }
    gray[i] = (uint8_t)((c[0] * 4899 + c[1] * 9617 + c[2] * 1868 + (1 << 13)) >> 14);
  }
}

static inline int wrap(int v, int n) { /* (v + n) % n of cen_cc.cc:30,34; made non-negative for n < 4 (UB in the reference) */
  int r = v % n;
  return r < 0 ? r + n : r;
}

/* cen_cc.cc:19-45: 9x9 census, 80 bits, bit k = (centre > neighbour), wrap-around borders; bits packed LSB first */
static void census_codes(const uint8_t *gray, int w, int h, uint32_t *code /* 3 words per pixel */) {
  const int H_WD = 9 / 2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t *c = code + ((size_t)y * w + x) * 3;
      c[0] = c[1] = c[2] = 0;
      int bitCnt = 0;
      for (int wy = -H_WD; wy <= H_WD; wy++) {
        int qy = wrap(y + wy, h);
        for (int wx = -H_WD; wx <= H_WD; wx++)
          if (wy != 0 || wx != 0) {
            int qx = wrap(x + wx, w);
            if (gray[(size_t)y * w + x] > gray[(size_t)qy * w + qx]) c[bitCnt >> 5] |= 1u << (bitCnt & 31);
            bitCnt++;
          }
      }
    }
}
static inline int hamming80(const uint32_t *a, const uint32_t *b) {
  return __builtin_popcount(a[0] ^ b[0]) + __builtin_popcount(a[1] ^ b[1]) + __builtin_popcount(a[2] ^ b[2]);
}

static void cen_build(const double *l, const double *r, int w, int h, int maxDis, int right, double *vol) {
  uint8_t *lg = (uint8_t *)malloc((size_t)w * h), *rg = (uint8_t *)malloc((size_t)w * h);
  uint32_t *lc = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (size_t)w * h), *rc = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (size_t)w * h);
  rgb64_to_gray8(l, w, h, lg);
  rgb64_to_gray8(r, w, h, rg);
  census_codes(lg, w, h, lc);
  census_codes(rg, w, h, rc);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      for (int d = 0; d < maxDis; d++) {
        double *cost = vol + ((size_t)d * h + y) * w;
        cost[x] = 80; /* CENCUS_BIT, cen_cc.cc:56,123 */
        if (!right) { /* cen_cc.cc:47-66 */
          if (x - d >= 0) cost[x] = hamming80(lc + ((size_t)y * w + x) * 3, rc + ((size_t)y * w + x - d) * 3);
        } else {      /* cen_cc.cc:114-133 */
          if (x + d < w) cost[x] = hamming80(rc + ((size_t)y * w + x) * 3, lc + ((size_t)y * w + x + d) * 3);
        }
      }
  free(lg); free(rg); free(lc); free(rc);
}
/* CenCC::buildCV / buildRightCV (cc/cen_cc.cc:4-70, 72-137) */
void csor_cen_build_cv(const double *l, const double *r, int w, int h, int maxDis, double *vol) { cen_build(l, r, w, h, maxDis, 0, vol); }
void csor_cen_build_right_cv(const double *l, const double *r, int w, int h, int maxDis, double *vol) { cen_build(l, r, w, h, maxDis, 1, vol); }

/* ------------------------------------------------------------------ PreSSPC / PreCSPC */

#define CSOR_MAX_LEVELS 16
struct csor_pc {
  int cs;        /* 0: PreSSPC, 1: PreCSPC */
  int scale_num; /* levels (1 for SS) */
  int wnd_size, half_wnd;
  int wid[CSOR_MAX_LEVELS], hei[CSOR_MAX_LEVELS], max_disp[CSOR_MAX_LEVELS];
  uint8_t *img[2][CSOR_MAX_LEVELS];
  double *vol[2][CSOR_MAX_LEVELS];
  int img_kind;  /* 1: GrdPC / CSPC -- no volumes, cells computed from the images at real-valued positions */
  double *grd[2][CSOR_MAX_LEVELS]; /* GrdPC::grd_x_ / CSPC::grd_x_ (img_kind only) */
  double max_cost[2][CSOR_MAX_LEVELS];
  /* CSOR_SUM_DEVICE ("device order"): GRD cells with the last step contracted,
   * fma(1-ALPHA, grdDiff, ALPHA*clrDiff), and their max.  NULL / equal to the above when the device reads the very same
   * cells (census: exact integers; volumes a test overwrote = a foreign CCMethod). */
  double *vol_dev[2][CSOR_MAX_LEVELS];
  double max_cost_dev[2][CSOR_MAX_LEVELS];
  double scale_wgt[CSOR_MAX_LEVELS];
  double lookup_exp[1000];
};

/* cvtColor(BGR2RGB) + convertTo(CV_64F): pre_cs_pc.cc:60-64, pre_ss_pc.cc:31-36 */
static double *bgr8_to_rgb64(const uint8_t *bgr, int w, int h) {
  double *o = (double *)malloc(sizeof(double) * (size_t)w * h * 3);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    o[i * 3 + 0] = bgr[i * 3 + 2];
    o[i * 3 + 1] = bgr[i * 3 + 1];
    o[i * 3 + 2] = bgr[i * 3 + 0];
  }
  return o;
}


/* ---- GrdPC / CSPC ingredients (plane_cost/grd_pc.h:13-17, cspc.h:13-17) ---- */
#define IMG_COST_ALPHA 0.1
#define IMG_TAU_CLR 10.0
#define IMG_TAU_GRD 2.0
/* cvtColor(8UC3, CV_BGR2GRAY), OpenCV 2.4: (B*1868 + G*9617 + R*4899 + (1<<13)) >> 14   (grd_pc.cc:37, cspc.cc:55) */
static void bgr8_to_gray8(const uint8_t *bgr, int w, int h, uint8_t *gray) {
  for (size_t i = 0; i < (size_t)w * h; ++i)
    gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14);
}
/* Sobel(gray8U, CV_64F, 1, 0, ksize=1): [-1 0 1], BORDER_REFLECT_101   (grd_pc.cc:40, cspc.cc:58) */
static void sobel_x_ks1_u8(const uint8_t *gray, int w, int h, double *grd) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int xm = x - 1, xp = x + 1;
      if (w == 1) { xm = 0; xp = 0; }
      else { if (xm < 0) xm = -xm; if (xp >= w) xp = 2 * (w - 1) - xp; }
      grd[(size_t)y * w + x] = (double)((int)gray[(size_t)y * w + xp] - (int)gray[(size_t)y * w + xm]);
    }
}
/* commfunc.h:129-145 */
static inline int handle_border(int loc, int size) {
  if (loc < 0) return loc + size;
  if (loc >= size) return loc - size;
  return loc;
}

static double vol_max(const csor_pc *pc, const double *vol, int s) { /* pre_cs_pc.cc:75-82, pre_ss_pc.cc:51-58 */
  double m = -1.0;
  const size_t n = (size_t)(pc->max_disp[s] + 1) * pc->hei[s] * pc->wid[s];
  for (size_t i = 0; i < n; ++i)
    if (vol[i] > m) m = vol[i];
  return m;
}
/* The caller replaced the volumes (a foreign CCMethod's cells): both summation orders read exactly these cells. */
void csor_pc_refresh_max_cost(csor_pc *pc) {
  if (pc->img_kind) return;
  for (int v = 0; v < 2; ++v)
    for (int s = 0; s < pc->scale_num; ++s) {
      free(pc->vol_dev[v][s]);
      pc->vol_dev[v][s] = NULL;
      pc->max_cost[v][s] = pc->max_cost_dev[v][s] = vol_max(pc, pc->vol[v][s], s);
    }
}

csor_pc *csor_pc_create(const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h, int max_disp,
                        int wnd_size, int scale_num, double reg_lambda) {
  return csor_pc_create_cc(l_bgr, r_bgr, w, h, max_disp, wnd_size, scale_num, reg_lambda, CSOR_CC_GRD);
}

csor_pc *csor_pc_create_cc(const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h, int max_disp,
                           int wnd_size, int scale_num, double reg_lambda, int cc_kind) {
  if (scale_num < 0 || scale_num > CSOR_MAX_LEVELS || w < 1 || h < 1 || max_disp < 1) return NULL;
  csor_pc *pc = (csor_pc *)calloc(1, sizeof *pc);
  pc->cs = scale_num > 0;
  pc->scale_num = pc->cs ? scale_num : 1;
  pc->wnd_size = wnd_size;
  pc->half_wnd = wnd_size / 2;
  const uint8_t *src[2] = {l_bgr, r_bgr};
  /* pre_cs_pc.cc:36-55: pyramid, per-level dims and disparity ranges */
  for (int v = 0; v < 2; ++v)
    for (int s = 0; s < pc->scale_num; ++s) {
      if (s == 0) {
        pc->wid[0] = w; pc->hei[0] = h; pc->max_disp[0] = max_disp;
        pc->img[v][0] = (uint8_t *)malloc((size_t)w * h * 3);
        memcpy(pc->img[v][0], src[v], (size_t)w * h * 3);
      } else {
        pc->hei[s] = (pc->hei[s - 1] + 1) / 2;
        pc->wid[s] = (pc->wid[s - 1] + 1) / 2;
        pc->max_disp[s] = pc->max_disp[s - 1] / 2;
        pc->img[v][s] = (uint8_t *)malloc((size_t)pc->wid[s] * pc->hei[s] * 3);
        csor_pyrdown_bgr8(pc->img[v][s - 1], pc->wid[s - 1], pc->hei[s - 1], pc->img[v][s]);
      }
    }
  if (cc_kind == CSOR_CC_IMG) {
    /* GrdPC (grd_pc.cc:27-49) / CSPC (cspc.cc:37-61): per level cvtColor(BGR2GRAY) on 8UC3 and Sobel(gray, CV_64F, 1, 0, 1).
     * No volumes; the "impossible disparity" cost is the constant of grd_pc.cc:131-132 / cspc.cc:150-152. */
    pc->img_kind = 1;
    for (int v = 0; v < 2; ++v)
      for (int s = 0; s < pc->scale_num; ++s) {
        const int W = pc->wid[s], H = pc->hei[s];
        uint8_t *gray = (uint8_t *)malloc((size_t)W * H);
        bgr8_to_gray8(pc->img[v][s], W, H, gray);
        pc->grd[v][s] = (double *)malloc(sizeof(double) * (size_t)W * H);
        sobel_x_ks1_u8(gray, W, H, pc->grd[v][s]);
        free(gray);
        pc->max_cost[v][s] = pc->max_cost_dev[v][s] = IMG_COST_ALPHA * IMG_TAU_CLR + (1 - IMG_COST_ALPHA) * IMG_TAU_GRD;
      }
    if (pc->cs) csor_scale_weights(pc->scale_num, reg_lambda, pc->scale_wgt); /* cspc.cc:63-87 */
    else pc->scale_wgt[0] = 1.0;
    csor_exp_lut(pc->lookup_exp, WGT_GAMMA); /* grd_pc.cc:61-64, cspc.cc:88-92 */
    return pc;
  }
  /* pre_cs_pc.cc:57-84: volumes with max_disp_s+1 slabs, built with maxDis = max_disp_s+1 */
  for (int s = 0; s < pc->scale_num; ++s) {
    double *tl = bgr8_to_rgb64(pc->img[0][s], pc->wid[s], pc->hei[s]);
    double *tr = bgr8_to_rgb64(pc->img[1][s], pc->wid[s], pc->hei[s]);
    const size_t n = (size_t)(pc->max_disp[s] + 1) * pc->hei[s] * pc->wid[s];
    pc->vol[0][s] = (double *)calloc(n, sizeof(double));
    pc->vol[1][s] = (double *)calloc(n, sizeof(double));
    if (cc_kind == CSOR_CC_CEN) {
      csor_cen_build_cv(tl, tr, pc->wid[s], pc->hei[s], pc->max_disp[s] + 1, pc->vol[0][s]);
      csor_cen_build_right_cv(tl, tr, pc->wid[s], pc->hei[s], pc->max_disp[s] + 1, pc->vol[1][s]);
    } else {
      csor_grd_build_cv(tl, tr, pc->wid[s], pc->hei[s], pc->max_disp[s] + 1, pc->vol[0][s]);
      csor_grd_build_right_cv(tl, tr, pc->wid[s], pc->hei[s], pc->max_disp[s] + 1, pc->vol[1][s]);
      for (int v = 0; v < 2; ++v) {
        pc->vol_dev[v][s] = (double *)calloc(n, sizeof(double));
        grd_build_dev(tl, tr, pc->wid[s], pc->hei[s], pc->max_disp[s] + 1, v, pc->vol_dev[v][s]);
      }
    }
    free(tl);
    free(tr);
  }
  for (int v = 0; v < 2; ++v)
    for (int s = 0; s < pc->scale_num; ++s) {
      pc->max_cost[v][s] = vol_max(pc, pc->vol[v][s], s);
      pc->max_cost_dev[v][s] = pc->vol_dev[v][s] ? vol_max(pc, pc->vol_dev[v][s], s) : pc->max_cost[v][s];
    }
  if (pc->cs) csor_scale_weights(pc->scale_num, reg_lambda, pc->scale_wgt);
  else pc->scale_wgt[0] = 1.0;
  csor_exp_lut(pc->lookup_exp, WGT_GAMMA);
  return pc;
}

void csor_pc_destroy(csor_pc *pc) {
  if (!pc) return;
  for (int v = 0; v < 2; ++v)
    for (int s = 0; s < pc->scale_num; ++s) { free(pc->img[v][s]); free(pc->vol[v][s]); free(pc->vol_dev[v][s]); free(pc->grd[v][s]); }
  free(pc);
}
int csor_pc_levels(const csor_pc *pc) { return pc->scale_num; }
void csor_pc_level_dims(const csor_pc *pc, int s, int *w, int *h, int *md) { *w = pc->wid[s]; *h = pc->hei[s]; *md = pc->max_disp[s]; }
const uint8_t *csor_pc_image(const csor_pc *pc, int view, int s) { return pc->img[view][s]; }
double *csor_pc_volume(csor_pc *pc, int view, int s) { return pc->vol[view][s]; }
double csor_pc_max_cost(const csor_pc *pc, int view, int s) { return pc->max_cost[view][s]; }
/* the cells / max_cost a CSOR_SUM_DEVICE evaluation reads */
double *csor_pc_volume_dev(csor_pc *pc, int view, int s) { return pc->vol_dev[view][s] ? pc->vol_dev[view][s] : pc->vol[view][s]; }
double csor_pc_max_cost_dev(const csor_pc *pc, int view, int s) { return pc->max_cost_dev[view][s]; }
const double *csor_pc_scale_wgt(const csor_pc *pc) { return pc->scale_wgt; }

/* static_cast<int>(double) as x86 cvttsd2si executes it: out-of-range / NaN -> INT_MIN, which
 * lands in the "impossible disparity" branch (pre_cs_pc.cc:166-169). */
static inline int trunc_x86(double q) {
  if (!(q > -2147483649.0 && q < 2147483648.0)) return INT_MIN;
  return (int)q;
}

/* one window tap: pre_cs_pc.cc:160-177 / pre_ss_pc.cc:94-110.  Returns the guide weight in *wgt_out and the (interpolated)
 * cost it multiplies; q_disp is the caller's (the two summation orders form it differently, see level_cost).
 * dev != 0: the device order's cells and its contracted interpolation c0 + fr*(c1 - c0) as ONE fma -- the same value as
 * floor_wgt*c0 + (1-floor_wgt)*c1 up to rounding (floor_wgt = 1 - fr and 1 - floor_wgt = fr exactly for a valid tap). */
static inline __attribute__((always_inline)) double tap(const csor_pc *pc, int view, int s, const uint8_t *I_p, int q_x, int q_y,
                         double q_disp, int dev, double *wgt_out) {
  const uint8_t *I_q = pc->img[view][s] + ((size_t)q_y * pc->wid[s] + q_x) * 3;
  int sum = abs(I_p[0] - I_q[0]) + abs(I_p[1] - I_q[1]) + abs(I_p[2] - I_q[2]);
  *wgt_out = pc->lookup_exp[sum];
  int q_disp_floor = trunc_x86(q_disp);
  if (pc->img_kind) {
    /* GrdPC::GetPlaneCost (grd_pc.cc:128-169, the !USE_INTER build) / CSPC::GetPlaneCost (cspc.cc:147-174) */
    if (q_disp_floor <= 0 || q_disp_floor >= pc->max_disp[s])
      return IMG_COST_ALPHA * IMG_TAU_CLR + (1 - IMG_COST_ALPHA) * IMG_TAU_GRD;
    const int W = pc->wid[s];
    const double other_x = q_x + (2 * view - 1) * q_disp;
    int floor_x = trunc_x86(other_x);
    int ceil_x = floor_x + 1;
    const double floor_wgt = ceil_x - other_x;
    floor_x = handle_border(floor_x, W);
    ceil_x = handle_border(ceil_x, W);
    const uint8_t *I_other_y = pc->img[1 - view][s] + (size_t)q_y * W * 3;
    const uint8_t *I_floor = I_other_y + 3 * floor_x, *I_ceil = I_other_y + 3 * ceil_x;
    double clr_cost = fabs(I_q[0] - I_ceil[0] + floor_wgt * (I_ceil[0] - I_floor[0])) +
                      fabs(I_q[1] - I_ceil[1] + floor_wgt * (I_ceil[1] - I_floor[1])) +
                      fabs(I_q[2] - I_ceil[2] + floor_wgt * (I_ceil[2] - I_floor[2]));
    clr_cost *= 0.33333333333333;
    const double *G_other_y = pc->grd[1 - view][s] + (size_t)q_y * W;
    const double G_floor = G_other_y[floor_x], G_ceil = G_other_y[ceil_x];
    const double G_q = pc->grd[view][s][(size_t)q_y * W + q_x];
    const double grd_cost = fabs(G_q - G_ceil + floor_wgt * (G_ceil - G_floor));
    return IMG_COST_ALPHA * (clr_cost < IMG_TAU_CLR ? clr_cost : IMG_TAU_CLR) +
           (1 - IMG_COST_ALPHA) * (grd_cost < IMG_TAU_GRD ? grd_cost : IMG_TAU_GRD);
  }
  if (q_disp_floor <= 0 || q_disp_floor >= pc->max_disp[s]) return dev ? pc->max_cost_dev[view][s] : pc->max_cost[view][s];
  const size_t slab = (size_t)pc->hei[s] * pc->wid[s];
  const size_t at = (size_t)q_disp_floor * slab + (size_t)q_y * pc->wid[s] + q_x;
  if (dev) {
    const double *c0 = (pc->vol_dev[view][s] ? pc->vol_dev[view][s] : pc->vol[view][s]) + at;
    const double fr = q_disp - (double)q_disp_floor; /* exact */
    return fma(fr, c0[slab] - c0[0], c0[0]);
  }
  int q_disp_ceil = q_disp_floor + 1;
  const double floor_wgt = q_disp_ceil - q_disp;
  const double *c0 = pc->vol[view][s] + at;
  return floor_wgt * c0[0] + (1 - floor_wgt) * c0[slab];
}

/* Speed only: a second copy of level_cost for CPUs with FMA3, picked at load time.  fma() is correctly rounded by definition, so the
 * inlined vfmadd instruction and libm's fma() return the same bits; -ffp-contract=off still forbids the compiler to contract any
 * OTHER expression in either copy.  (A whole KITTI-size pair in the device order: 156 s -> about 100 s of the GPU suite.) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define CSOR_FMA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define CSOR_FMA_CLONES
#endif
#define ROWMOD_K 7   /* CSOR_SUM_DEVICE: interleaved partial sums per window row */
#define ROWTREE_N 64 /* ... and a balanced binary tree over the window rows (window sizes up to 45 < 64) */

/* aggregated cost of ONE level (inner loops of pre_cs_pc.cc:151-181 / pre_ss_pc.cc:82-115).
 * base/mul: total so far and the level's scale weight, used only for the early-exit test
 * base + partial*mul >= thresh (meaningful only when every term is >= 0; serial order: after every window row,
 * device order: at the level end).  Returns 1 and the level sum in *sum, or 0 if rejected early.
 *
 * CSOR_SUM_SERIAL: one running sum over (dy outer, dx inner) -- the reference's order and arithmetic.
 * CSOR_SUM_DEVICE ("ROWTREE7" with contracted multiply-adds):
 *   - a tap's disparity is formed per group of 7 window columns: q_disp(dx) = fma(a, dx % 7, G) with
 *     G = fma(a, q_x of the group's first column, q_disp_y)  (the reference: a*q_x + q_disp_y, pre_cs_pc.cc:165);
 *   - cells and interpolation as tap(dev = 1); a tap enters its partial sum as S = fma(wgt, value, S);
 *   - within window row dy (0-based), tap dx (0-based window column) is accumulated in dx order into the partial sum
 *     S[dx % 7]; the row total is R[dy] = (((((S0+S1)+S2)+S3)+S4)+S5)+S6;
 *   - the level sum is the balanced binary tree over R[0..63] (rows >= the window size and rows outside the image are
 *     +0.0): pairs (0,1),(2,3).., then pairs of pairs, ... -- what an xor butterfly over 64 lanes computes, and what a
 *     single lane computes with a binary-counter stack of six pending partial sums.
 *   Taps outside the image are skipped (adding +0.0 changes nothing: every partial sum starts at +0.0).
 *   Same terms as the serial order, other rounding. */
CSOR_FMA_CLONES
static int level_cost(const csor_pc *pc, int view, int s, int cx, int cy, double a, double b, double c,
                      int sum_order, double base, double mul, double thresh, int use_thresh, long long *taps, double *sum) {
  const int half = pc->half_wnd, W = pc->wid[s], H = pc->hei[s];
  const uint8_t *I_p = pc->img[view][s] + ((size_t)cy * W + cx) * 3;
  long long nt = 0;
  double acc = 0.0;
  double R[ROWTREE_N];
  for (int l = 0; l < ROWTREE_N; ++l) R[l] = 0.0;
  for (int dy = -half; dy <= half; ++dy) {
    int q_y = cy + dy;
    if (q_y < 0 || q_y >= H) continue;
    const double q_disp_y = b * q_y + c;
    if (sum_order == CSOR_SUM_SERIAL) {
      for (int dx = -half; dx <= half; ++dx) {
        int q_x = cx + dx;
        if (q_x >= 0 && q_x < W) {
          double wgt;
          const double t = tap(pc, view, s, I_p, q_x, q_y, a * q_x + q_disp_y, 0, &wgt); /* :165 */
          acc += wgt * t; /* :176-177 */
          ++nt;
        }
      }
      if (use_thresh && base + acc * mul >= thresh) { if (taps) *taps += nt; return 0; }
    } else {
      double S[ROWMOD_K];
      for (int j = 0; j < ROWMOD_K; ++j) S[j] = 0.0;
      for (int col0 = 0; col0 <= 2 * half; col0 += ROWMOD_K) { /* groups of 7 window columns; column col0 + j feeds S[j] */
        const int x0 = cx - half + col0;
        const double G = fma(a, (double)x0, q_disp_y);
        for (int j = 0; j < ROWMOD_K && col0 + j <= 2 * half; ++j) {
          const int q_x = x0 + j;
          if (q_x >= 0 && q_x < W) {
            double wgt;
            const double t = tap(pc, view, s, I_p, q_x, q_y, fma(a, (double)j, G), 1, &wgt);
            S[j] = fma(wgt, t, S[j]);
            ++nt;
          }
        }
      }
      double r = S[0];
      for (int j = 1; j < ROWMOD_K; ++j) r = r + S[j];
      R[dy + half] = r;
    }
  }
  if (sum_order != CSOR_SUM_SERIAL) {
    for (int off = 1; off < ROWTREE_N; off <<= 1)
      for (int l = 0; l < ROWTREE_N; l += 2 * off) R[l] = R[l] + R[l + off];
    acc = R[0];
    if (use_thresh && base + acc * mul >= thresh) { if (taps) *taps += nt; return 0; }
  }
  if (taps) *taps += nt;
  *sum = acc;
  return 1;
}

double csor_pc_cost_thresh(const csor_pc *pc, int ref_x, int ref_y, const double norm[3],
                           const double param[3], int view, int sum_order, double thresh, long long *taps) {
  const int use_thresh = thresh < K_DOUBLE_MAX;
  if (taps) *taps = 0;
  if (!pc->cs) {
    /* PreSSPC::GetPlaneCost (pre_ss_pc.cc:74-118): uses plane.param() as is, one level */
    double c;
    if (!level_cost(pc, view, 0, ref_x, ref_y, param[0], param[1], param[2], sum_order, 0.0, 1.0, thresh, use_thresh, taps, &c)) return INFINITY;
    return c;
  }
  /* PreCSPC::GetPlaneCost (pre_cs_pc.cc:133-188) */
  double cost = 0.0;
  double cur_disp = param[0] * ref_x + param[1] * ref_y + param[2]; /* :139-140 */
  int cur_y = ref_y, cur_x = ref_x;
  for (int s = 0; s < pc->scale_num; ++s) {
    const double pt[3] = {(double)cur_x, (double)cur_y, cur_disp};
    double prm[3];
    csor_plane_param(norm, pt, prm); /* :144  Plane cur_plane(org_norm, Point3d(cur_x,cur_y,cur_disp)) */
    double scale_cost;
    if (!level_cost(pc, view, s, cur_x, cur_y, prm[0], prm[1], prm[2], sum_order, cost, pc->scale_wgt[s], thresh, use_thresh, taps, &scale_cost))
      return INFINITY;
    cost += scale_cost * pc->scale_wgt[s]; /* :182 */
    cur_y /= 2; cur_x /= 2; cur_disp /= 2.0; /* :183-185 */
  }
  return cost;
}

/* the per-level sums of PreCSPC::GetPlaneCost (pre_cs_pc.cc:142-186) before they are weighted: level_out[s] = scale_cost of
 * level s (study tool: tools/lb_exit_study.py).  Returns the number of levels. */
int csor_pc_level_costs(const csor_pc *pc, int ref_x, int ref_y, const double norm[3], const double param[3], int view,
                        int sum_order, double *level_out) {
  if (!pc->cs) {
    level_cost(pc, view, 0, ref_x, ref_y, param[0], param[1], param[2], sum_order, 0.0, 1.0, K_DOUBLE_MAX, 0, NULL, &level_out[0]);
    return 1;
  }
  double cur_disp = param[0] * ref_x + param[1] * ref_y + param[2];
  int cur_y = ref_y, cur_x = ref_x;
  for (int s = 0; s < pc->scale_num; ++s) {
    const double pt[3] = {(double)cur_x, (double)cur_y, cur_disp};
    double prm[3];
    csor_plane_param(norm, pt, prm);
    level_cost(pc, view, s, cur_x, cur_y, prm[0], prm[1], prm[2], sum_order, 0.0, 1.0, K_DOUBLE_MAX, 0, NULL, &level_out[s]);
    cur_y /= 2; cur_x /= 2; cur_disp /= 2.0;
  }
  return pc->scale_num;
}

double csor_pc_cost(const csor_pc *pc, int x, int y, const double norm[3], const double param[3], int view, int sum_order) {
  return csor_pc_cost_thresh(pc, x, y, norm, param, view, sum_order, K_DOUBLE_MAX, NULL);
}

long long csor_pc_taps(const csor_pc *pc, int x, int y) {
  long long t = 0;
  int cx = x, cy = y;
  for (int s = 0; s < pc->scale_num; ++s) {
    const int h = pc->half_wnd;
    int x0 = cx - h < 0 ? 0 : cx - h, x1 = cx + h >= pc->wid[s] ? pc->wid[s] - 1 : cx + h;
    int y0 = cy - h < 0 ? 0 : cy - h, y1 = cy + h >= pc->hei[s] ? pc->hei[s] - 1 : cy + h;
    t += (long long)(x1 - x0 + 1) * (y1 - y0 + 1);
    cx /= 2; cy /= 2;
  }
  return t;
}

/* ------------------------------------------------------------------ CSPatchMatch */

typedef struct { double n[3], p[3], prm[3]; } plane_t; /* plane.h:45-48 */

struct csor_pm {
  int wid, hei, max_dis, dis_scale;
  uint8_t *img[2];
  uint8_t *dis[2];
  plane_t *plane[2];
  double *min_cost[2];
  long long evals;
};

static const csor_pm_opts k_default_opts = {12345, CSOR_RNG_PER_PIXEL, CSOR_SCHED_RASTER, CSOR_SUM_SERIAL, 1, 4, 0, 0};

/* cs_patchmatch.cc:3-34 */
csor_pm *csor_pm_create(const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h, int max_dis, int dis_scale) {
  csor_pm *pm = (csor_pm *)calloc(1, sizeof *pm);
  pm->wid = w; pm->hei = h; pm->max_dis = max_dis; pm->dis_scale = dis_scale;
  const uint8_t *src[2] = {l_bgr, r_bgr};
  for (int v = 0; v < 2; ++v) {
    pm->img[v] = (uint8_t *)malloc((size_t)w * h * 3);
    memcpy(pm->img[v], src[v], (size_t)w * h * 3);
    pm->dis[v] = (uint8_t *)calloc((size_t)w * h, 1);
    pm->plane[v] = (plane_t *)calloc((size_t)w * h, sizeof(plane_t));
    pm->min_cost[v] = (double *)malloc(sizeof(double) * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) pm->min_cost[v][i] = K_DOUBLE_MAX;
  }
  return pm;
}
void csor_pm_destroy(csor_pm *pm) {
  if (!pm) return;
  for (int v = 0; v < 2; ++v) { free(pm->img[v]); free(pm->dis[v]); free(pm->plane[v]); free(pm->min_cost[v]); }
  free(pm);
}
const uint8_t *csor_pm_dis(const csor_pm *pm, int view) { return pm->dis[view]; }
double *csor_pm_planes(csor_pm *pm, int view) { return (double *)pm->plane[view]; }
double *csor_pm_min_cost(csor_pm *pm, int view) { return pm->min_cost[view]; }
long long csor_pm_evals(const csor_pm *pm) { return pm->evals; }

static inline double plane_cost(const csor_pc *pc, int x, int y, const plane_t *pl, int v, const csor_pm_opts *o) {
  return csor_pc_cost(pc, x, y, pl->n, pl->prm, v, o->sum_order);
}
static inline uint64_t pix_key(const csor_pm *pm, const csor_pm_opts *o, int x, int y) {
  return o->rng_mode == CSOR_RNG_ROW_SHARED ? (uint64_t)x : (uint64_t)y * pm->wid + x;
}

/* cs_patchmatch.cc:115-148  InitRandomPlane.
 * z0 ~ U(kDoubleEps, max_dis) as in :134-135.  The reference fills the normal with N(0,1)^3 and
 * normalises (:137-140), i.e. a uniformly distributed direction; cv::RNG's ziggurat is not
 * reproducible on a GPU bit for bit, so both sides draw the direction by rejection sampling in
 * the unit ball (no transcendental functions) -- same distribution, see DESIGN.md "RNG". */
static void apply_threads(const csor_pm_opts *o) {
#ifdef _OPENMP
  if (o->threads > 0) omp_set_num_threads(o->threads);
#else
  (void)o;
#endif
}
void csor_pm_init(csor_pm *pm, const csor_pc *pc, const csor_pm_opts *o) {
  if (!o) o = &k_default_opts;
  apply_threads(o);
  for (int v = 0; v < 2; ++v) {
    const uint32_t sid = csor_stream_id(0, 0, 0, v);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < pm->hei; ++y)
      for (int x = 0; x < pm->wid; ++x) {
        plane_t *pl = &pm->plane[v][(size_t)y * pm->wid + x];
        const uint64_t pk = pix_key(pm, o, x, y);
        double rand_dis = uni(csor_rng_u01(o->seed, sid, pk, 0), K_DOUBLE_EPS, (double)pm->max_dis);
        pl->p[0] = x; pl->p[1] = y; pl->p[2] = rand_dis;
        double rn[3] = {0.0, 0.0, 1.0};
        double len = 1.0;
        for (int t = 0; t < 32; ++t) {
          for (int k = 0; k < 3; ++k) rn[k] = uni(csor_rng_u01(o->seed, sid, pk, 1 + 3 * t + k), -1.0, 1.0);
          double s = rn[0] * rn[0];
          s += rn[1] * rn[1];
          s += rn[2] * rn[2];
          len = sqrt(s); /* norm(rand_norm, NORM_L2) */
          if (s <= 1.0 && s > 1e-12) break;
        }
        double denom = fmax(len, K_DOUBLE_EPS);
        const double inv = 1. / denom; /* cv::Vec / double multiplies by the reciprocal */
        for (int k = 0; k < 3; ++k) pl->n[k] = rn[k] * inv;
        csor_plane_param(pl->n, pl->p, pl->prm);
        pm->min_cost[v][(size_t)y * pm->wid + x] = plane_cost(pc, x, y, pl, v, o);
      }
    pm->evals += (long long)pm->wid * pm->hei;
  }
}

static inline void try_plane(csor_pm *pm, const csor_pc *pc, const csor_pm_opts *o, int v, int x, int y, const plane_t *cand) {
  const size_t i = (size_t)y * pm->wid + x;
  const double c = plane_cost(pc, x, y, cand, v, o);
  if (c < pm->min_cost[v][i]) {
    pm->min_cost[v][i] = c;
    pm->plane[v][i] = *cand;
  }
}

/* cs_patchmatch.cc:163-216  SpatialPropagation, reference raster order */
static void spatial_raster(csor_pm *pm, int cur_iter, const csor_pc *pc, const csor_pm_opts *o) {
  const int wid = pm->wid, hei = pm->hei;
  int x_st = wid - 2, x_ed = -1, x_inc = -1;
  int y_st = hei - 2, y_ed = -1, y_inc = -1;
  if (cur_iter % 2 == 0) {
    x_st = 1; x_ed = wid; x_inc = 1;
    y_st = 1; y_ed = hei; y_inc = 1;
  }
  for (int v = 0; v < 2; ++v) {
    plane_t *P = pm->plane[v];
    /* first row, y_st - y_inc (:178-186) */
    for (int x = x_st; x != x_ed; x += x_inc) {
      plane_t nx_plane = P[(size_t)(y_st - y_inc) * wid + (x - x_inc)];
      try_plane(pm, pc, o, v, x, y_st - y_inc, &nx_plane);
      pm->evals++;
    }
    for (int y = y_st; y != y_ed; y += y_inc) {
      /* first column, x_st - x_inc (:188-195) */
      plane_t ny0 = P[(size_t)(y - y_inc) * wid + (x_st - x_inc)];
      try_plane(pm, pc, o, v, x_st - x_inc, y, &ny0);
      pm->evals++;
      for (int x = x_st; x != x_ed; x += x_inc) {
        plane_t nx_plane = P[(size_t)y * wid + (x - x_inc)]; /* :198-204 */
        try_plane(pm, pc, o, v, x, y, &nx_plane);
        plane_t ny_plane = P[(size_t)(y - y_inc) * wid + x]; /* :206-212 */
        try_plane(pm, pc, o, v, x, y, &ny_plane);
        pm->evals += 2;
      }
    }
  }
}

/* The same sweep, anti-diagonal by anti-diagonal (csor_pm_opts.wavefront): sweep coordinates xs + ys = k; the in-place serial loop
 * above makes (x,y) depend on (x-inc,y) and (x,y-inc) only -- both on diagonal k-1, both final when diagonal k starts -- so the
 * pixels of a diagonal may run in any order or in parallel.  x-predecessor first, then y-predecessor against the updated minimum
 * (:198-212); the first sweep row has only the former (:178-186), the first column only the latter (:189-195). */
static void spatial_raster_wavefront(csor_pm *pm, int cur_iter, const csor_pc *pc, const csor_pm_opts *o) {
  const int wid = pm->wid, hei = pm->hei;
  const int inc = (cur_iter % 2 == 0) ? 1 : -1;
  for (int v = 0; v < 2; ++v) {
    plane_t *P = pm->plane[v];
    for (int k = 1; k <= wid + hei - 2; ++k) {
      const int ys_lo = k - (wid - 1) > 0 ? k - (wid - 1) : 0, ys_hi = k < hei - 1 ? k : hei - 1;
      long long ev = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : ev)
      for (int ys = ys_lo; ys <= ys_hi; ++ys) {
        const int xs = k - ys;
        const int x = inc > 0 ? xs : wid - 1 - xs, y = inc > 0 ? ys : hei - 1 - ys;
        if (xs > 0) {
          plane_t nx_plane = P[(size_t)y * wid + (x - inc)];
          try_plane(pm, pc, o, v, x, y, &nx_plane);
          ++ev;
        }
        if (ys > 0) {
          plane_t ny_plane = P[(size_t)(y - inc) * wid + x];
          try_plane(pm, pc, o, v, x, y, &ny_plane);
          ++ev;
        }
      }
      pm->evals += ev;
    }
  }
}

/* Device fast-path schedule (DESIGN.md "red-black"): per round two half-steps; in half-step hs
 * the pixels with ((x+y)&1) == ((hs + cur_iter)&1) test the planes of their in-image neighbours
 * in the order (x-inc,y), (x,y-inc), (x+inc,y), (x,y+inc) [first 2 only if rb_neighbours==2],
 * inc=+1 on even iterations, -1 on odd ones (the reference's sweep direction).  Neighbours have
 * the other colour, so a half-step has no intra-step dependencies. */
static void spatial_redblack(csor_pm *pm, int cur_iter, const csor_pc *pc, const csor_pm_opts *o) {
  const int wid = pm->wid, hei = pm->hei;
  const int inc = (cur_iter % 2 == 0) ? 1 : -1;
  const int nb = o->rb_neighbours == 2 ? 2 : 4;
  const int rounds = o->rb_rounds < 1 ? 1 : o->rb_rounds;
  for (int r = 0; r < rounds; ++r)
    for (int hs = 0; hs < 2; ++hs) {
      const int colour = (hs + cur_iter) & 1;
      for (int v = 0; v < 2; ++v) {
        plane_t *P = pm->plane[v];
        long long ev = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : ev)
        for (int y = 0; y < hei; ++y)
          for (int x = 0; x < wid; ++x) {
            if (((x + y) & 1) != colour) continue;
            const int nxs[4] = {x - inc, x, x + inc, x}, nys[4] = {y, y - inc, y, y + inc};
            for (int k = 0; k < nb; ++k) {
              if (nxs[k] < 0 || nxs[k] >= wid || nys[k] < 0 || nys[k] >= hei) continue;
              plane_t cand = P[(size_t)nys[k] * wid + nxs[k]];
              try_plane(pm, pc, o, v, x, y, &cand);
              ++ev;
            }
          }
        pm->evals += ev;
      }
    }
}

void csor_pm_spatial(csor_pm *pm, int cur_iter, const csor_pc *pc, const csor_pm_opts *o) {
  apply_threads(o ? o : &k_default_opts);
  if (!o) o = &k_default_opts;
  if (o->schedule == CSOR_SCHED_REDBLACK) spatial_redblack(pm, cur_iter, pc, o);
  else if (o->wavefront) spatial_raster_wavefront(pm, cur_iter, pc, o);
  else spatial_raster(pm, cur_iter, pc, o);
}

/* cs_patchmatch.cc:229-277  ViewPropagation (serial scatter; the device resolves the same
 * candidates in parallel with a traversal-order tie-break, which is equivalent). */
void csor_pm_view(csor_pm *pm, int cur_iter, const csor_pc *pc, const csor_pm_opts *o) {
  if (!o) o = &k_default_opts;
  const int wid = pm->wid, hei = pm->hei;
  plane_t cor_plane;
  int x_st = wid - 1, x_ed = -1, x_inc = -1;
  int y_st = hei - 1, y_ed = -1, y_inc = -1;
  if (cur_iter % 2 == 0) {
    x_st = 0; x_ed = wid; x_inc = 1;
    y_st = 0; y_ed = hei; y_inc = 1;
  }
  for (int v = 0; v < 2; ++v) {
    const int other_view = 1 - v;
    for (int y = y_st; y != y_ed; y += y_inc)
      for (int x = x_st; x != x_ed; x += x_inc) {
        const plane_t *src = &pm->plane[other_view][(size_t)y * wid + x];
        double disp = src->prm[0] * x + src->prm[1] * y + src->prm[2];
        if (disp < 0.0) disp = 0.0;
        if (disp >= pm->max_dis) disp = pm->max_dis - 1.0;
        int cor_x;
        if (v == CSOR_LEFT) cor_x = csor_handle_border(x + csor_round2int(disp), wid);
        else cor_x = csor_handle_border(x - csor_round2int(disp), wid);
        if (cor_x < 0 || cor_x >= wid) continue; /* max_dis > wid: the reference would index out of range */
        memcpy(cor_plane.n, src->n, sizeof cor_plane.n);
        cor_plane.p[0] = cor_x; cor_plane.p[1] = y; cor_plane.p[2] = disp;
        csor_plane_param(cor_plane.n, cor_plane.p, cor_plane.prm);
        try_plane(pm, pc, o, v, cor_x, y, &cor_plane);
        pm->evals++;
      }
  }
}

/* cs_patchmatch.cc:292-345  PlaneRefinement(max_dis/2.0, kMaxNorm_, kZStopThres_) */
void csor_pm_refine(csor_pm *pm, int cur_iter, const csor_pc *pc, const csor_pm_opts *o) {
  if (!o) o = &k_default_opts;
  apply_threads(o);
  double z_iter = pm->max_dis / 2.0, n_iter = K_MAX_NORM;
  int step = 0;
  while (z_iter >= K_Z_STOP) {
    for (int v = 0; v < 2; ++v) {
      const uint32_t sid = csor_stream_id(1, cur_iter, step, v);
#pragma omp parallel for schedule(dynamic, 1)
      for (int y = 0; y < pm->hei; ++y)
        for (int x = 0; x < pm->wid; ++x) {
          const size_t i = (size_t)y * pm->wid + x;
          const plane_t *cur = &pm->plane[v][i];
          const uint64_t pk = pix_key(pm, o, x, y);
          plane_t dp;
          double disturb_z = cur->prm[0] * x + cur->prm[1] * y + cur->prm[2];
          dp.p[0] = x; dp.p[1] = y;
          dp.p[2] = disturb_z + uni(csor_rng_u01(o->seed, sid, pk, 0), -z_iter, z_iter);
          double dn[3];
          for (int k = 0; k < 3; ++k) dn[k] = cur->n[k] + uni(csor_rng_u01(o->seed, sid, pk, 1 + k), -n_iter, n_iter);
          double s = dn[0] * dn[0];
          s += dn[1] * dn[1];
          s += dn[2] * dn[2];
          double denom = fmax(sqrt(s), K_DOUBLE_EPS);
          const double inv = 1. / denom;
          for (int k = 0; k < 3; ++k) dp.n[k] = dn[k] * inv;
          csor_plane_param(dp.n, dp.p, dp.prm);
          const double c = plane_cost(pc, x, y, &dp, v, o);
          if (c < pm->min_cost[v][i]) {
            pm->plane[v][i] = dp;
            pm->min_cost[v][i] = c;
          }
        }
      pm->evals += (long long)pm->wid * pm->hei;
    }
    z_iter /= 2.0;
    n_iter /= 2.0;
    ++step;
  }
}

static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline double plane_disp(const plane_t *p, int x, int y) { /* param().dot(Vec3d(x,y,1.0)) */
  double s = p->prm[0] * x;
  s += p->prm[1] * y;
  s += p->prm[2] * 1.0;
  return s;
}

/* cs_patchmatch.cc:590-601 */
void csor_pm_plane_to_disp(csor_pm *pm) {
  for (int v = 0; v < 2; ++v)
    for (int y = 0; y < pm->hei; ++y)
      for (int x = 0; x < pm->wid; ++x) {
        double disp = plane_disp(&pm->plane[v][(size_t)y * pm->wid + x], x, y);
        pm->dis[v][(size_t)y * pm->wid + x] = sat_u8(csor_round2int(disp * pm->dis_scale));
      }
}
void csor_pm_disp_f64(const csor_pm *pm, int view, double *out) {
  for (int y = 0; y < pm->hei; ++y)
    for (int x = 0; x < pm->wid; ++x) out[(size_t)y * pm->wid + x] = plane_disp(&pm->plane[view][(size_t)y * pm->wid + x], x, y);
}

/* cs_patchmatch.cc:347-369 */
static void left_right_check(csor_pm *pm, int **valid) {
  for (int v = 0; v < 2; ++v) {
    int *cur_valid = valid[v];
    for (int y = 0; y < pm->hei; y++) {
      const uint8_t *cur_dis_row = pm->dis[v] + (size_t)y * pm->wid;
      const uint8_t *other_dis_row = pm->dis[1 - v] + (size_t)y * pm->wid;
      for (int x = 0; x < pm->wid; x++) {
        *cur_valid = 0;
        double cur_dis = cur_dis_row[x] * 1.0 / pm->dis_scale;
        int other_x = x + (2 * v - 1) * csor_round2int(cur_dis);
        if (other_x >= 0 && other_x < pm->wid) {
          double other_dis = other_dis_row[other_x] * 1.0 / pm->dis_scale;
          if (fabs(cur_dis - other_dis) <= 0.5 && cur_dis > 0.0) *cur_valid = 1;
        }
        ++cur_valid;
      }
    }
  }
}

/* cs_patchmatch.cc:370-428 */
static void fill_invalid(csor_pm *pm, int **valid) {
  for (int v = 0; v < 2; ++v) {
    int *cur_valid = valid[v];
    for (int y = 0; y < pm->hei; ++y) {
      int *y_valid = valid[v] + (size_t)y * pm->wid;
      uint8_t *dis_data = pm->dis[v] + (size_t)y * pm->wid;
      const plane_t *prow = pm->plane[v] + (size_t)y * pm->wid;
      for (int x = 0; x < pm->wid; ++x) {
        if (*cur_valid == 0) {
          int l_first = x, l_find = 0;
          while (l_first >= 0) { if (y_valid[l_first]) { l_find = 1; break; } --l_first; }
          int r_find = 0, r_first = x;
          while (r_first < pm->wid) { if (y_valid[r_first]) { r_find = 1; break; } ++r_first; }
          if (l_find && r_find) {
            double l_d = plane_disp(&prow[l_first], x, y), r_d = plane_disp(&prow[r_first], x, y);
            if (l_d <= r_d) dis_data[x] = sat_u8(pm->dis_scale * csor_round2int(l_d));
            else dis_data[x] = sat_u8(pm->dis_scale * csor_round2int(r_d));
          } else if (l_find) {
            dis_data[x] = sat_u8(pm->dis_scale * csor_round2int(plane_disp(&prow[l_first], x, y)));
          } else if (r_find) {
            dis_data[x] = sat_u8(pm->dis_scale * csor_round2int(plane_disp(&prow[r_first], x, y)));
          }
        }
        ++cur_valid;
      }
    }
  }
}

/* cs_patchmatch.cc:430-506 */
static void weighted_median(csor_pm *pm, int **valid, int wnd_size, double gamma) {
  const int half_wnd = wnd_size / 2;
  double lookup_exp[1000], disp_hist[256];
  csor_exp_lut(lookup_exp, gamma);
  for (int v = 0; v < 2; ++v) {
    int *cur_valid = valid[v];
    for (int y = 0; y < pm->hei; y++) {
      uint8_t *cur_dis = pm->dis[v] + (size_t)y * pm->wid;
      const uint8_t *pL = pm->img[v] + (size_t)y * pm->wid * 3;
      for (int x = 0; x < pm->wid; x++) {
        if (*cur_valid == 0) {
          const uint8_t *pL_x = pL + 3 * x;
          for (int d = 0; d < 256; ++d) disp_hist[d] = 0.0;
          double sum_wgt = 0.0;
          for (int wy = -half_wnd; wy <= half_wnd; wy++) {
            const int qy = y + wy;
            if (qy >= 0 && qy < pm->hei) {
              const int *qLValid = valid[v] + (size_t)qy * pm->wid;
              const uint8_t *qL = pm->img[v] + (size_t)qy * pm->wid * 3;
              const uint8_t *q_dis_data = pm->dis[v] + (size_t)qy * pm->wid;
              for (int wx = -half_wnd; wx <= half_wnd; wx++) {
                const int qx = x + wx;
                if (qx >= 0 && qx < pm->wid && qLValid[qx]) {
                  const int q_disp = q_dis_data[qx];
                  const uint8_t *qL_x = qL + 3 * qx;
                  int clr_diff = abs(pL_x[0] - qL_x[0]) + abs(pL_x[1] - qL_x[1]) + abs(pL_x[2] - qL_x[2]);
                  double wgt = lookup_exp[clr_diff];
                  disp_hist[q_disp] += wgt;
                  sum_wgt += wgt;
                }
              }
            }
          }
          double median_wgt = sum_wgt / 2.0;
          sum_wgt = 0.0;
          int median_disp = 0;
          for (int d = 0; d < 256; d++) {
            sum_wgt += disp_hist[d];
            if (sum_wgt >= median_wgt) { median_disp = d; break; }
          }
          if (median_wgt > 0.0) cur_dis[x] = (uint8_t)median_disp;
        }
        cur_valid++;
      }
    }
  }
}

/* cs_patchmatch.cc:508-588 */
void csor_pm_postprocess(csor_pm *pm) {
  int *valid[2];
  for (int v = 0; v < 2; ++v) valid[v] = (int *)calloc((size_t)pm->wid * pm->hei, sizeof(int));
  left_right_check(pm, valid);
  fill_invalid(pm, valid);
  weighted_median(pm, valid, 35, WMF_GAMMA);
  free(valid[0]);
  free(valid[1]);
}

/* cs_patchmatch.cc:51-109 */
void csor_pm_run(csor_pm *pm, int iter_num, const csor_pc *pc, int use_pp, const csor_pm_opts *o) {
  if (!o) o = &k_default_opts;
#ifdef _OPENMP
  if (o->threads > 0) omp_set_num_threads(o->threads);
#endif
  csor_pm_init(pm, pc, o);
  for (int i = 0; i < iter_num; ++i) {
    csor_pm_spatial(pm, i, pc, o);
    csor_pm_view(pm, i, pc, o);
    csor_pm_refine(pm, i, pc, o);
  }
  csor_pm_plane_to_disp(pm);
  if (use_pp) csor_pm_postprocess(pm);
}
