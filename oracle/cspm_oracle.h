/*
 * cspm_oracle.h -- CPU restatement (the parity ORACLE) of the PatchMatch-stereo hot path of
 * rookiepig/CrossScalePatchMatch.  TEST INFRASTRUCTURE ONLY.
 *
 *   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *   The product (libcspm_hip.so and everything under crossscalepatchmatch_amd/) never includes,
 *   links or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference cannot be compiled in this image: every translation unit
 * includes commfunc.h, which needs OpenCV 2.4.x and gflags (CSPM/commfunc.h:9,17); neither is
 * installed and stand-in headers are not an acceptable build.  The reference ships no tests,
 * golden images or known-answer vectors.  This oracle is therefore a
 * line-by-line restatement of the reference sources (each function cites the file:line it
 * follows) plus a restatement of the documented OpenCV 2.4 contracts it relies on (pyrDown,
 * RGB2GRAY on 32F, Sobel ksize=1, Vec3d::dot, Mat::inv); it is pinned only by hand-derived
 * known answers and by independent re-derivations in tests/, not by reference output.
 *
 * All reference citations are relative to /root/reference/CSPM/.
 */
#ifndef CSPM_ORACLE_H
#define CSPM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CSOR_LEFT = 0, CSOR_RIGHT = 1 };           /* commfunc.h:29  enum RefView */

/* summation order inside GetPlaneCost */
enum {
  CSOR_SUM_SERIAL = 0,  /* reference order: dy outer, dx inner, one accumulator (pre_cs_pc.cc:151-181) */
  CSOR_SUM_DEVICE = 1   /* device order "ROWTREE7": per window row, tap dx goes to partial sum S[dx % 7] in dx order; row total
                           = ((((((S0+S1)+S2)+S3)+S4)+S5)+S6); level sum = balanced binary tree over the 64 (zero-padded) row totals.
                           Multiply-adds are contracted (fma) at the sites DESIGN.md section 3.2 lists: the tap's disparity, the
                           last step of a GRD cell, the interpolation between the two cells, the accumulation */
};

/* propagation schedule of SpatialPropagation */
enum {
  CSOR_SCHED_RASTER   = 0, /* reference: in-place serial raster sweep (cs_patchmatch.cc:163-216) */
  CSOR_SCHED_REDBLACK = 1  /* device fast path: checkerboard half-steps */
};

/* rng flags */
enum {
  CSOR_RNG_PER_PIXEL  = 0, /* counter-based stream keyed by (seed, phase, iter, step, view, y, x) */
  CSOR_RNG_ROW_SHARED = 1  /* reproduces the USE_OMP quirk: every row re-seeds identically
                              (cs_patchmatch.cc:129-131,308-310) => key omits y */
};

/* ---------------- primitives (commfunc.h, plane.h) ---------------- */
int    csor_round2int(double d);                       /* commfunc.h:117-121 */
int    csor_handle_border(int loc, int size);          /* commfunc.h:129-145 */
void   csor_plane_param(const double n[3], const double p[3], double prm[3]); /* plane.h:25-34 */
void   csor_exp_lut(double *lut1000, double gamma);    /* pre_cs_pc.cc:111-114 */
int    csor_scale_weights(int scale_num, double reg_lambda, double *w); /* pre_cs_pc.cc:86-109 */
uint64_t csor_rng_u64(uint64_t seed, uint32_t stream, uint64_t pix, uint32_t draw);
double csor_rng_u01(uint64_t seed, uint32_t stream, uint64_t pix, uint32_t draw);
uint32_t csor_stream_id(int phase, int iter, int step, int view);

/* ---------------- image pyramid / cost computation ---------------- */
/* OpenCV 2.4 pyrDown on 8UC3 (pre_cs_pc.cc:45); dst is ((w+1)/2) x ((h+1)/2), packed 3 B/px */
void   csor_pyrdown_bgr8(const uint8_t *src, int w, int h, uint8_t *dst);
/* GrdCC::buildCV / buildRightCV (cc/grd_cc.cpp:60-154). l,r: h*w*3 doubles RGB 0..255.
 * vol: maxDis slabs of h*w doubles, d-major. */
void   csor_grd_build_cv(const double *l_rgb, const double *r_rgb, int w, int h, int maxDis, double *vol);
void   csor_grd_build_right_cv(const double *l_rgb, const double *r_rgb, int w, int h, int maxDis, double *vol);
/* gray (f32) and x-gradient (f64) helpers used by the two functions above (grd_cc.cpp:70-77) */
void   csor_rgb2gray_f32(const double *rgb, int w, int h, float *gray);
void   csor_sobel_x_ks1(const float *gray, int w, int h, double *grd);

/* CenCC::buildCV / buildRightCV (cc/cen_cc.cc:4-137): 9x9 census (80 bits, wrap-around border, gray via the 8U
 * RGB2GRAY fixed-point contract), Hamming distance, 80 where the other view is outside the image */
void   csor_cen_build_cv(const double *l_rgb, const double *r_rgb, int w, int h, int maxDis, double *vol);
void   csor_cen_build_right_cv(const double *l_rgb, const double *r_rgb, int w, int h, int maxDis, double *vol);
enum { CSOR_CC_GRD = 0, CSOR_CC_CEN = 1,  /* main.cc:39-55 GetCCType("GRD" / "CEN") */
       CSOR_CC_IMG = 2 };                   /* no CCMethod at all: GrdPC / CSPC (plane_cost/grd_pc.cc, cspc.cc) */

/* ---------------- plane cost objects: PreSSPC / PreCSPC ---------------- */
typedef struct csor_pc csor_pc;
/* scale_num == 0  -> PreSSPC (pre_ss_pc.cc:12-65), single level, GetPlaneCost uses plane.param()
 * scale_num >= 1  -> PreCSPC (pre_cs_pc.cc:12-115) with that many levels.
 * l_bgr/r_bgr: packed 8UC3 BGR, h rows of w*3 bytes.  cost function: GRD. */
csor_pc *csor_pc_create(const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h,
                        int max_disp, int wnd_size, int scale_num, double reg_lambda);
/* cc_kind == CSOR_CC_IMG: scale_num == 0 -> GrdPC (plane_cost/grd_pc.cc:11-66, 72-176), scale_num >= 1 -> CSPC
 * (plane_cost/cspc.cc:11-93, 107-183): the volume-free IPlaneCost variants that interpolate the other view's colour and
 * gradient at the real-valued position x -+ q_disp instead of interpolating pre-computed cells.  csor_pc_volume() is NULL. */
csor_pc *csor_pc_create_cc(const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h,
                           int max_disp, int wnd_size, int scale_num, double reg_lambda, int cc_kind);
void     csor_pc_destroy(csor_pc *pc);
int      csor_pc_levels(const csor_pc *pc);
void     csor_pc_level_dims(const csor_pc *pc, int s, int *w, int *h, int *max_disp);
const uint8_t *csor_pc_image(const csor_pc *pc, int view, int s);     /* packed BGR of level s */
double  *csor_pc_volume(csor_pc *pc, int view, int s);                /* (max_disp_s+1) slabs */
double   csor_pc_max_cost(const csor_pc *pc, int view, int s);
/* what a CSOR_SUM_DEVICE evaluation reads: GRD cells with the contracted last step and their max; the plain volumes
 * for census and after csor_pc_refresh_max_cost (a foreign CCMethod's cells are taken as they are) */
double  *csor_pc_volume_dev(csor_pc *pc, int view, int s);
double   csor_pc_max_cost_dev(const csor_pc *pc, int view, int s);
void     csor_pc_refresh_max_cost(csor_pc *pc);   /* after a test overwrote volumes (foreign CCMethod) */
const double *csor_pc_scale_wgt(const csor_pc *pc);
/* IPlaneCost::GetPlaneCost (i_plane_cost.h:28-33).  norm/point/param as in class Plane. */
double   csor_pc_cost(const csor_pc *pc, int x, int y, const double norm[3], const double param[3],
                      int view, int sum_order);
/* same with a reject threshold: returns +inf as soon as the partial sum proves cost >= thresh
 * (result-preserving: all terms are >= 0 when volumes and scale weights are >= 0). taps (may be
 * NULL) receives the number of window taps actually evaluated. */
double   csor_pc_cost_thresh(const csor_pc *pc, int x, int y, const double norm[3],
                             const double param[3], int view, int sum_order, double thresh,
                             long long *taps);
/* unweighted per-level sums of one evaluation (a study hook, tools/lb_exit_study.py); returns the number of levels */
int      csor_pc_level_costs(const csor_pc *pc, int x, int y, const double norm[3], const double param[3], int view,
                             int sum_order, double *level_out);
/* exact number of in-image window taps of one evaluation at (x,y) */
long long csor_pc_taps(const csor_pc *pc, int x, int y);

/* ---------------- CSPatchMatch ---------------- */
typedef struct csor_pm csor_pm;
typedef struct {
  uint64_t seed;
  int rng_mode;      /* CSOR_RNG_* */
  int schedule;      /* CSOR_SCHED_* */
  int sum_order;     /* CSOR_SUM_* */
  int rb_rounds;     /* red-black rounds per iteration (>=1); ignored for raster */
  int rb_neighbours; /* 2 or 4 */
  int threads;       /* OpenMP threads for init/refinement rows (0 = default) */
  int wavefront;     /* 0: the raster sweep is the reference's serial double loop (cs_patchmatch.cc:163-216).  1: the same sweep walked
                        anti-diagonal by anti-diagonal with the pixels of a diagonal in parallel -- pixel (x,y) reads only (x-inc,y) and
                        (x,y-inc), both on the previous diagonal, so the result is the serial loop's bit for bit (asserted by
                        tests/test_oracle_primitives.py); a speed knob for the whole-KITTI-pair parity test, nothing else */
} csor_pm_opts;

csor_pm *csor_pm_create(const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h,
                        int max_dis, int dis_scale);            /* cs_patchmatch.cc:3-34 */
void     csor_pm_destroy(csor_pm *pm);
/* CSPatchMatch::PatchMatch (cs_patchmatch.cc:51-109) */
void     csor_pm_run(csor_pm *pm, int iter_num, const csor_pc *pc, int use_pp, const csor_pm_opts *o);
/* individual phases, for phase-by-phase parity tests */
void     csor_pm_init(csor_pm *pm, const csor_pc *pc, const csor_pm_opts *o);
void     csor_pm_spatial(csor_pm *pm, int iter, const csor_pc *pc, const csor_pm_opts *o);
void     csor_pm_view(csor_pm *pm, int iter, const csor_pc *pc, const csor_pm_opts *o);
void     csor_pm_refine(csor_pm *pm, int iter, const csor_pc *pc, const csor_pm_opts *o);
void     csor_pm_plane_to_disp(csor_pm *pm);
void     csor_pm_postprocess(csor_pm *pm);
/* state access: planes as 9 doubles per pixel (norm[3], point[3], param[3]) */
const uint8_t *csor_pm_dis(const csor_pm *pm, int view);
double  *csor_pm_planes(csor_pm *pm, int view);
double  *csor_pm_min_cost(csor_pm *pm, int view);
void     csor_pm_disp_f64(const csor_pm *pm, int view, double *out);
long long csor_pm_evals(const csor_pm *pm);
int      csor_refine_steps(int max_dis);

#ifdef __cplusplus
}
#endif
#endif
