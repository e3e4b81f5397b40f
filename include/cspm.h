/*
 * cspm.h -- C ABI of libcspm_hip.so: the MI355X (gfx950) implementation of the PatchMatch-stereo
 * hot path of rookiepig/CrossScalePatchMatch (GRD plane cost, single-scale and cross-scale).
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference's
 * CSPM/ directory).  Conventions:
 *   - extern "C", plain pointers and sizes, no C++/torch types; int status: 0 = OK, < 0 = error
 *     (cspm_last_error() gives the text); no exceptions cross the boundary.
 *   - a cspm_ctx owns all device memory of ONE stereo pair on ONE GPU and ONE HIP stream; it is not
 *     thread-safe; use one ctx per host thread / stream.  Host buffers are caller-owned.
 *   - "view": 0 = left (kLeft), 1 = right (kRight)  (commfunc.h:29).
 *   - images are packed 8UC3 BGR, the layout cv::imread(CV_LOAD_IMAGE_COLOR) returns (main.cc:68-69).
 *   - the library has no CPU fallback: every call fails with CSPM_ERR_HIP when no gfx950 device is
 *     usable.
 */
#ifndef CSPM_H
#define CSPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSPM_OK 0
#define CSPM_ERR_ARG (-1)   /* bad argument (the reference CV_Asserts or crashes) */
#define CSPM_ERR_HIP (-2)   /* HIP runtime error, no device, out of memory */
#define CSPM_ERR_STATE (-3) /* call order violated (e.g. patchmatch before a cost is built) */

#define CSPM_MAX_LEVELS 8

typedef struct cspm_ctx cspm_ctx;

/* SpatialPropagation schedule (cs_patchmatch.cc:163-216) */
#define CSPM_SCHED_RASTER 0   /* reference order: in-place raster sweep, run as an anti-diagonal wavefront */
#define CSPM_SCHED_REDBLACK 1 /* checkerboard half-steps (option: fewer dependencies, clearly lower quality) */

/* rng flags */
#define CSPM_RNG_PER_PIXEL 0
#define CSPM_RNG_ROW_SHARED 1 /* the USE_OMP per-row re-seeding quirk (cs_patchmatch.cc:129-131,308-310) */

typedef struct cspm_pm_params {
  uint64_t seed;     /* replaces RNG(time(NULL)), cs_patchmatch.cc:32 */
  int schedule;      /* CSPM_SCHED_* */
  int rb_rounds;     /* red-black rounds per iteration (>= 1) */
  int rb_neighbours; /* 2 or 4 */
  int rng_mode;      /* CSPM_RNG_* */
  int early_exit;    /* 1: stop a plane evaluation once its partial sum proves cost >= min_cost
                        (result-preserving; ignored when a scale weight or max_cost is negative) */
} cspm_pm_params;

/* ---- lifetime ------------------------------------------------------------------------------ */
int cspm_device_count(void);
int cspm_create(cspm_ctx **out, int device);
void cspm_destroy(cspm_ctx *ctx);
const char *cspm_last_error(const cspm_ctx *ctx); /* ctx may be NULL: error of a failed cspm_create */
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream */
int cspm_set_stream(cspm_ctx *ctx, void *hip_stream);
/* the hipStream_t the ctx enqueues on (its own non-blocking stream unless cspm_set_stream replaced it): for callers that order their
 * own work against it (events, torch.cuda.ExternalStream).  The batch driver keeps one ctx-owned stream per pair in flight: streams a
 * framework hands out may share a hardware queue, and two pairs on one queue run one after the other. */
int cspm_get_stream(cspm_ctx *ctx, void **hip_stream_out);
int cspm_synchronize(cspm_ctx *ctx);

/* ---- images: PreSSPC/PreCSPC/CSPatchMatch constructors' (l_img, r_img) ----------------------
 * pre_ss_pc.cc:12-30, pre_cs_pc.cc:12-30, cs_patchmatch.cc:3-11.  stride in bytes (>= 3*w). */
int cspm_set_images(cspm_ctx *ctx, const uint8_t *l_bgr, const uint8_t *r_bgr, int w, int h, size_t stride);
/* same, from device memory already resident in HBM (bench / batch driver).  The copy kernels are enqueued on the ctx
 * stream and nothing waits for the host: the caller orders the PRODUCER of the buffers before this call (same stream,
 * an event the ctx stream waits on, or a synchronisation) and keeps them alive until the stream has passed the call. */
int cspm_set_images_device(cspm_ctx *ctx, const void *d_l_bgr, const void *d_r_bgr, int w, int h, size_t stride);

/* ---- plane cost construction ------------------------------------------------------------------
 * cspm_build_cost_grd = `new PreSSPC(l,r,max_dis,wnd,new GrdCC)` when scale_num == 0
 * (pre_ss_pc.cc:12-65) and `new PreCSPC(l,r,max_dis,wnd,scale_num,new GrdCC,reg_lambda)` when
 * scale_num >= 1 (pre_cs_pc.cc:12-115): pyramid, per-level GRD cost volumes of both views
 * (cc/grd_cc.cpp:60-154), max_cost, scale weights, exp LUT -- all on the device. */
int cspm_build_cost_grd(cspm_ctx *ctx, int max_dis, int wnd_size, int scale_num, double reg_lambda);
/* options (set before cspm_build_cost_grd).
 * CSPM_OPT_GRD_VOLUMES (applies to the GRD and the census cost): 0 (default) = cell costs are recomputed on the fly from the images and
 * gradients inside the PatchMatch kernels (bit-identical to reading GrdCC's volumes, no 1-30 GB cost
 * volume in HBM); 1 = materialise the d-major f64 volumes exactly as PreCSPC does (pre_cs_pc.cc:50-73)
 * and read them. */
#define CSPM_OPT_GRD_VOLUMES 1
/* CSPM_OPT_RASTER_LAUNCHES: 0 (default) = the reference-order raster sweep runs as ONE persistent launch whose
 * workgroups hand pixels over through per-pixel data-tagged granules (the final plane, polled directly); 1 = one launch per anti-diagonal (W+H-2 launches
 * per sweep; same results, kept as a cross-check). */
#define CSPM_OPT_RASTER_LAUNCHES 2
/* CSPM_OPT_SWEEP_TIMEOUT_MS (default 3000; 0 = every wait fails at once, for tests): how long a workgroup of the persistent sweep
 * waits for a predecessor pixel before the sweep gives up.  A timeout is slowness (shared GPU, profiler, many contexts in
 * flight), never a wrong result: when exactly one whole cspm_patchmatch ran since the last synchronising call, that call
 * repeats it with per-diagonal launches (identical planes) and reports success; otherwise it reports CSPM_ERR_HIP.
 * CSPM_OPT_SWEEP_FALLBACKS (read only): how many times that happened on this context. */
#define CSPM_OPT_SWEEP_TIMEOUT_MS 3
#define CSPM_OPT_SWEEP_FALLBACKS 4
/* CSPM_OPT_SWEEP_PAIRS (set before cspm_build_cost_grd; GRD with fused cells only; default 0): 1 = when they fit the context's
 * budget (4 GiB; a KITTI-size pair needs 2.2 GB, a 3000x2000 D=256 pair would need 56 GB and does without), the cost
 * constructor also materialises the GRD cells as PAIRS {cell(d), cell(d+1)} per (d, y, x) -- the two f64 cells a tap interpolates
 * between (pre_cs_pc.cc:171-176) -- and the raster sweep (SpatialPropagation, whose taps are gathers) reads one pair per tap instead
 * of recomputing two cells from three image gathers; every other kernel keeps the fused cells.  Same cells, same order: identical
 * planes.  Measured on MI355X: no faster than the fused sweep (20.4 vs 20.2 ms per sweep of a KITTI-size pair; the 16-byte gathers
 * into a 1 GB volume cost the L1 return path more than the three 12-byte image gathers they replace), hence off by default.
 * CSPM_OPT_SWEEP_PAIRS_ACTIVE (read only): 1 when the current cost object carries the pairs. */
#define CSPM_OPT_SWEEP_PAIRS 5
#define CSPM_OPT_SWEEP_PAIRS_ACTIVE 6
/* CSPM_OPT_TABLE_VOLUMES (set before cspm_build_cost_grd; GRD with fused cells only; default 1): when they fit the context's budget
 * (48 GiB -- the device has 288 GB; a KITTI-size pair needs 1.2 GB, a 3000x2000 D=256 pair 30 GB) the cost constructor also keeps the GRD
 * cells as d-major f64 volumes -- what PreCSPC keeps (pre_cs_pc.cc:50-73) -- and the row kernels (InitRandomPlane, ViewPropagation,
 * PlaneRefinement) fill their per-row cell tables from them by LDS-DMA instead of recomputing the cells, wherever a wave's lanes
 * agree on a narrow disparity range; everything else still computes cells on the fly.  Same cells: identical planes.  0 = never.
 * CSPM_OPT_TABLE_VOLUMES_ACTIVE (read only): 1 when the current cost object carries them.
 * Both kinds of optional volume (this one and CSPM_OPT_SWEEP_PAIRS) are accelerators, never a reason for a pair to fail: beyond the
 * per-context budgets they are taken only out of memory that hipMemGetInfo reports FREE when the cost object is allocated (at most
 * half of it, env CSPM_VOLUMES_MEM_FRACTION), and when a hipMalloc for one of them fails all the same the constructor releases them
 * and goes on with computed tables / the fused sweep -- identical results, as on a device with less HBM than MI355X's 288 GB.
 * CSPM_OPT_VOLUME_FALLBACKS (read only): how many times such an allocation failed on this context. */
#define CSPM_OPT_TABLE_VOLUMES 7
#define CSPM_OPT_TABLE_VOLUMES_ACTIVE 8
#define CSPM_OPT_VOLUME_FALLBACKS 9
/* CSPM_OPT_SWEEP_PACKED (set before cspm_build_cost_grd; GRD with fused cells only; default 0): 1 = the raster sweep (SpatialPropagation,
 * whose window taps are gathers) reads the level images as PACKED 8-byte pixels {36-bit fixed-point x-gradient, 24-bit colour} --
 * lossless: the gradient of an 8-bit image's f32 gray values (grd_cc.cpp:70-77) is a multiple of 2^-27 below 256 -- so that a tap's two
 * adjacent other-view pixels arrive with one 16-byte gather and its own pixel with one 8-byte gather (2 gathers / 24 B instead of
 * 3 / 36 B per tap through the CU's L1).  Same cells, same order: identical planes.  0 = the 12-byte pixels every other kernel reads.
 * Measured on MI355X: 8 % SLOWER (65.3 against 60.5 ms of sweeps per KITTI-size pair) although the microbenchmark confirms the L1 path
 * does a third less work -- the sweep is bound by the latency of its dependent steps, and unpacking adds instructions to each: off.
 * CSPM_OPT_SWEEP_PACKED_ACTIVE (read only): 1 when the current cost object carries the packed pixels.
 * CSPM_OPT_SWEEP_PACKED_BAD (read only; synchronises): pixels the packer could not represent exactly -- 0 by construction; counted per
 * cost object, and a non-zero count fails the next synchronising call with CSPM_ERR_HIP (the sweep would have read wrong cells). */
/* CSPM_OPT_SWEEP_FLOW (default 0; measured on MI355X: 1 is 35 % slower, 26.9 against 20.0 ms per sweep of a KITTI-size pair): how the
 * persistent raster sweep (CSPM_OPT_RASTER_LAUNCHES = 0) hands out its pixels.  1 = by dataflow:
 * every pixel counts its final predecessors and the workgroup that completes the count continues with it at once (the other ready
 * successor goes to a queue idle workgroups pop); 0 = workgroups claim pixels in diagonal-major order and wait for their predecessors.
 * The dependencies, hence the planes, are the same: the reference's in-place raster order (cs_patchmatch.cc:163-216). */
#define CSPM_OPT_SWEEP_FLOW 13
/* CSPM_OPT_SWEEP_WG (default 0 = the library chooses): workgroups of the persistent raster sweep launched per CU.  The library's choice is 2 --
 * a KITTI-size sweep is bound by its dependency chain (one: 29 instead of 20 ms per sweep; three: no faster) and every resident workgroup
 * holds registers another pair's kernels would use -- and 3 for images whose anti-diagonals are many times wider than the resident
 * workgroups (2 * min(w, h) >= 4 * CUs: 1242 x 600 27 instead of 31 ms per sweep, 3000 x 2000 215 instead of 275) unless CSPM_OPT_SWEEP_FOLD says that the GPU is
 * shared.  Any value gives identical planes. */
#define CSPM_OPT_SWEEP_WG 14
/* CSPM_OPT_VOLUME_RETRY_PAIRS (default 16; 0 = never): a cost object that wanted optional volumes and runs without them (free-memory veto
 * or a failed hipMalloc, see CSPM_OPT_TABLE_VOLUMES) asks for them again after this many pairs have reused it.
 * CSPM_OPT_FAULT_VOLUME_ALLOC (write only, TEST HOOK): the n-th optional-volume allocation of this context from now on fails as if
 * hipMalloc had returned out-of-memory; a set_option call, never the environment, so that no deployment can switch it on by accident. */
#define CSPM_OPT_VOLUME_RETRY_PAIRS 15
/* CSPM_OPT_VIEW_SORT (default 1): ViewPropagation (cs_patchmatch.cc:229-277) evaluates the proposals of a row in the order of their TARGET
 * column instead of their source column, so that the 64 proposals of a wavefront land next to each other in the target view even where
 * the source disparities jump (a depth edge); 0 = source order.  The accept rule (smallest cost, earliest in the reference's traversal
 * among equals, :256-272) is applied afterwards per target pixel and does not depend on the order of evaluation: identical planes. */
#define CSPM_OPT_VIEW_SORT 17
/* CSPM_OPT_SWEEP_FOLD (default 0; cross-scale costs with 4 or more levels): 1 = the raster sweep's workgroups have one wavefront FEWER than
 * the cost has pyramid levels -- four, one per SIMD, for the reference's five levels (main.cc:100) -- and the coarsest level is evaluated by
 * the wavefronts of levels 1.. after their own; 0 = one wavefront per level.  Same taps and sums: identical planes.  For callers that keep
 * TWO OR MORE pairs in flight on one GPU (contexts on separate streams): a CU that holds two four-wavefront sweep workgroups has room for
 * two workgroups of another pair's refinement where two five-wavefront ones leave room for one (of three), so the refinement beside a
 * sweep runs at 2/3 instead of 1/3 of its speed -- measured with 3 pairs in flight: 137.9 against 142.8 ms per KITTI-size pair.  A pair
 * that has the GPU to itself is slower folded (a sweep takes 23.0 instead of 20.2 ms: three wavefronts walk 5 window passes instead of 4).
 * With it, CSPM_OPT_SWEEP_WG stays at its default (2) also for three pairs in flight. */
#define CSPM_OPT_SWEEP_FOLD 18
#define CSPM_OPT_FAULT_VOLUME_ALLOC 16
#define CSPM_OPT_SWEEP_PACKED 10
#define CSPM_OPT_SWEEP_PACKED_ACTIVE 11
#define CSPM_OPT_SWEEP_PACKED_BAD 12
int cspm_get_option(cspm_ctx *ctx, int key, long long *value);
int cspm_set_option(cspm_ctx *ctx, int key, long long value);
/* The same constructors with `new CenCC` (main.cc:43-45; cc/cen_cc.cc:4-137): 9x9 census codes of every level built on
 * the device; Hamming cells are computed on the fly from the codes (default) or materialised as f64 volumes
 * (CSPM_OPT_GRD_VOLUMES = 1), exactly like the GRD cost. */
int cspm_build_cost_cen(cspm_ctx *ctx, int max_dis, int wnd_size, int scale_num, double reg_lambda);
/* The two volume-free IPlaneCost implementations the reference also ships (not instantiated by its main.cc):
 *   scale_num == 0 -> `new GrdPC(l_img, r_img, max_disp, wnd_size)`                          plane_cost/grd_pc.h:27-29, grd_pc.cc:11-66
 *   scale_num >= 1 -> `new CSPC(l_img, r_img, max_disp, wnd_size, scale_num, reg_lambda)`    plane_cost/cspc.h:21-23,  cspc.cc:11-93
 * GetPlaneCost (grd_pc.cc:72-176, cspc.cc:107-183) interpolates the other view's colour and 8U-gray x-gradient at the
 * real-valued column x -+ q_disp (wrap-around HandleBorder) instead of interpolating pre-computed cells; the "impossible
 * disparity" cost is the constant COST_ALPHA*TAU_CLR + (1-COST_ALPHA)*TAU_GRD.  cspm_get_cost_slab is an error for these. */
int cspm_build_cost_img(cspm_ctx *ctx, int max_dis, int wnd_size, int scale_num, double reg_lambda);
/* Foreign CCMethod plugins (cc_method.h:31-32): allocate like the constructors above, then upload
 * the host volumes the plugin filled slab by slab, then finalize (max_cost reduction). */
int cspm_begin_cost(cspm_ctx *ctx, int max_dis, int wnd_size, int scale_num, double reg_lambda);
int cspm_upload_cost_slab(cspm_ctx *ctx, int view, int level, int d, const double *slab, size_t stride_elems);
int cspm_finish_cost(cspm_ctx *ctx);
/* introspection of what the constructor built (parity hooks) */
int cspm_get_levels(const cspm_ctx *ctx);
int cspm_get_level_dims(const cspm_ctx *ctx, int level, int *w, int *h, int *max_disp);
int cspm_get_level_image(cspm_ctx *ctx, int view, int level, uint8_t *bgr_out);     /* packed w*h*3 */
int cspm_get_cost_slab(cspm_ctx *ctx, int view, int level, int d, double *slab_out); /* packed w*h */
int cspm_get_max_cost(cspm_ctx *ctx, int view, int level, double *out);
int cspm_get_scale_weights(const cspm_ctx *ctx, double *out /* levels */);

/* CCMethod::buildCV / buildRightCV on host buffers (cc_method.h:31-32, cc/grd_cc.cpp:60-154):
 * l_rgb/r_rgb are h*w*3 doubles (CV_64FC3, RGB, 0..255), vol_out receives maxDis slabs of h*w. */
int cspm_grd_build_cv_host(int device, const double *l_rgb, const double *r_rgb, int w, int h, int maxDis,
                           int right_view, double *vol_out);

/* CenCC::buildCV / buildRightCV on host buffers (cc/cen_cc.cc:4-70, 72-137), same contract as above */
int cspm_cen_build_cv_host(int device, const double *l_rgb, const double *r_rgb, int w, int h, int maxDis,
                           int right_view, double *vol_out);

/* ---- IPlaneCost::GetPlaneCost, batched (plane_cost/i_plane_cost.h:28-33) ----------------------
 * xy: 2 ints per item; plane: 6 doubles per item = Plane::norm() then Plane::param().
 * The cost is computed in the DEVICE ORDER: the same terms as the reference's serial sum in the
 * "ROWTREE7" association, with five multiply-adds per tap contracted into fmas (disparity, the last step of a GRD cell x2,
 * interpolation, accumulation); differs from the reference's SSE2 arithmetic by rounding only, <= 1e-12 relative. */
int cspm_plane_cost_batch(cspm_ctx *ctx, int view, int n, const int *xy, const double *norm_param, double *cost_out);

/* ---- CSPatchMatch ------------------------------------------------------------------------------
 * cspm_patchmatch = CSPatchMatch::PatchMatch(iter_num, plane_cost, false) without PlaneToDisp
 * (cs_patchmatch.cc:51-102); max_dis / images are those of the ctx.  params == NULL: cspm_pm_default_params (raster
 * schedule).  ASYNCHRONOUS: the kernels are enqueued on the ctx stream and the call returns; an error inside the run
 * (a raster sweep whose inter-workgroup hand-off timed out) is reported by the next call that synchronises with the
 * host: cspm_synchronize or any cspm_get_* / cspm_postprocess. */
int cspm_pm_default_params(cspm_pm_params *p);
int cspm_patchmatch(cspm_ctx *ctx, int iter_num, const cspm_pm_params *p);
/* single phases (cs_patchmatch.cc:115-148, 163-216, 229-277, 292-345) for phase-by-phase parity */
int cspm_pm_init(cspm_ctx *ctx, const cspm_pm_params *p);
int cspm_pm_spatial(cspm_ctx *ctx, int iter, const cspm_pm_params *p);
int cspm_pm_view(cspm_ctx *ctx, int iter, const cspm_pm_params *p);
int cspm_pm_refine(cspm_ctx *ctx, int iter, const cspm_pm_params *p);
/* plane field in/out: 6 doubles per pixel (norm, param), row-major h*w; min_cost h*w doubles */
int cspm_get_planes(cspm_ctx *ctx, int view, double *norm_param_out, double *min_cost_out);
int cspm_set_planes(cspm_ctx *ctx, int view, const double *norm_param, const double *min_cost);
/* PlaneToDisp + dis() (cs_patchmatch.cc:590-601, 111-113): saturate_u8(Round2Int(d*dis_scale)) */
int cspm_get_disparity_u8(cspm_ctx *ctx, int view, int dis_scale, uint8_t *out, size_t stride);
int cspm_get_disparity_f64(cspm_ctx *ctx, int view, double *out); /* unquantised a*x+b*y+c */
/* device-resident result (u8, packed w*h) for the batch driver.  Asynchronous; when the PatchMatch run in front of it is repeated
 * after a sweep timeout (CSPM_OPT_SWEEP_TIMEOUT_MS), the map is written again from the repeated run's planes before the
 * synchronising call returns success.
 * CONTRACT for every asynchronous output (this call and cspm_postprocess_device): the buffer must stay allocated, and its contents
 * must not be consumed, until a synchronising cspm_* call on this ctx (cspm_synchronize, any cspm_get_*, cspm_postprocess) has
 * returned CSPM_OK after the request.  Synchronising the stream or an event of your own is NOT enough: only the cspm_* call looks at
 * the sweep's error word, and if the sweep timed out the map that stream-side synchronisation sees was written from the aborted run
 * and is rewritten inside that cspm_* call.  Repeated requests for the same (buffer, kind) between two synchronising calls are
 * recorded once. */
int cspm_disparity_u8_device(cspm_ctx *ctx, int view, int dis_scale, void *d_out);
/* PostProcessing (cs_patchmatch.cc:508-588) on the 8-bit maps */
int cspm_postprocess(cspm_ctx *ctx, int dis_scale, uint8_t *l_out, uint8_t *r_out, size_t stride);
/* the same with device-resident outputs (u8, packed w*h each); asynchronous on the ctx stream like cspm_patchmatch --
 * PatchMatch(iter_num, plane_cost, use_pp = true) without leaving the device (cs_patchmatch.cc:103-107) */
int cspm_postprocess_device(cspm_ctx *ctx, int dis_scale, void *d_l_out, void *d_r_out);

/* ---- CSPatchMatch::PatchMatch over a FOREIGN IPlaneCost (plane_cost/i_plane_cost.h:28-33) ------------------------------
 * Any object with a GetPlaneCost(x, y, plane, view) that is not one of this library's device costs: the reference drives it
 * through the virtual call (call sites cs_patchmatch.cc:144,181,191,200,208,269,334).  Here the device keeps the plane field,
 * draws every candidate from the same random streams and applies the reference's accept rules; the CALLER evaluates the
 * candidates with its cost function -- the plugin contract, batched:
 *     cspm_fpm_begin(ctx, w, h, max_dis);                       plane field, no cost object, no images needed
 *     per batch:  cspm_fpm_candidates(...) -> n candidates {xy, view, plane};  cost[i] = GetPlaneCost(...) for xy[2i] >= 0;
 *                 cspm_fpm_commit(ctx, cost)
 *   phase CSPM_FPM_INIT    (iter, step ignored): InitRandomPlane, 2*w*h candidates                (:115-148)
 *   phase CSPM_FPM_SPATIAL (step = anti-diagonal 0 .. w+h-2 of the raster sweep of iteration iter): 2 per pixel, x- then
 *                           y-predecessor's plane, xy[2i] = -1 where the pixel has no such predecessor (:163-216)
 *   phase CSPM_FPM_VIEW    (step = target view): w*h candidates, one per source pixel of the other view (:229-277)
 *   phase CSPM_FPM_REFINE  (step = halving step 0 ..): 2*w*h candidates                            (:292-345)
 * Arrays must hold max(2*w*h, 4*min(w,h)) candidates.  Synchronous.  Afterwards cspm_get_planes / cspm_get_disparity_* / cspm_postprocess
 * (which needs cspm_set_images for the weighted median) read the result as usual. */
#define CSPM_FPM_INIT 0
#define CSPM_FPM_SPATIAL 1
#define CSPM_FPM_VIEW 2
#define CSPM_FPM_REFINE 3
int cspm_fpm_begin(cspm_ctx *ctx, int w, int h, int max_dis);
int cspm_fpm_candidates(cspm_ctx *ctx, int phase, int iter, int step, const cspm_pm_params *p, int *n_out, int *xy_out, int *view_out,
                        double *plane_out);
int cspm_fpm_commit(cspm_ctx *ctx, const double *cost);

/* ---- measurement --------------------------------------------------------------------------------
 * When enabled, every kernel launch is bracketed by hipEvents on the ctx stream. */
#define CSPM_K_GRD 0      /* cost-volume construction kernels */
#define CSPM_K_INIT 1     /* plane cost evaluation: random init */
#define CSPM_K_SPATIAL 2  /* plane cost evaluation: spatial propagation */
#define CSPM_K_VIEW 3     /* plane cost evaluation: view propagation */
#define CSPM_K_REFINE 4   /* plane cost evaluation: plane refinement (the dominant kernel) */
#define CSPM_K_MISC 5     /* pyramid, resolve, disparity, ... */
#define CSPM_K_POST 6     /* PostProcessing: left-right check, fill, weighted median */
#define CSPM_K_COUNT 7
int cspm_enable_timing(cspm_ctx *ctx, int on);
int cspm_reset_timing(cspm_ctx *ctx);
/* launches, summed milliseconds and summed evaluated candidate planes of a kernel class */
int cspm_get_timing(cspm_ctx *ctx, int kclass, long long *launches, double *total_ms, long long *evals);
/* exact in-image window taps of ONE evaluation of every pixel of one view (sum over pixels, levels) */
long long cspm_taps_per_view_pass(const cspm_ctx *ctx);
/* lane-taps the row engine executes for the same pass (masked window columns and tail lanes included): the denominator of
 * "executed vs algorithmic taps" */
long long cspm_row_engine_taps_per_view_pass(const cspm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
