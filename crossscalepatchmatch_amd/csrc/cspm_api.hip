// cspm_api.hip -- host side of libcspm_hip.so: the C ABI declared in include/cspm.h.
// Owns device memory, builds the plane-cost object (PreSSPC / PreCSPC) on the device and drives the
// PatchMatch kernels.  No CPU fallback anywhere: without a usable gfx950 device every call fails.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cspm_kernels.h"

using namespace cspm;

namespace {

thread_local std::string g_create_error;  // error of a failed cspm_create, per calling thread (cspm_last_error(NULL))

struct TimingRec {
  int kclass;
  hipEvent_t a, b;
  long long evals;
};

}  // namespace

// every extern "C" entry that touches HIP runs on the ctx's device and leaves the caller's current device as it found it
struct DevGuard {
  int prev = -1;
  bool ok = true;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;
  }
  ~DevGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// what the buffers of a cost object were sized for: an identical request reuses them (no hipMalloc / hipFree per pair)
struct CostKey {
  int W = 0, H = 0, max_dis = 0, wnd = 0, scale_num = -1, with_vol = 0, kind = -1, with_pairs = 0, with_cvol = 0, with_px8 = 0;
  bool operator==(const CostKey &o) const {
    return W == o.W && H == o.H && max_dis == o.max_dis && wnd == o.wnd && scale_num == o.scale_num && with_vol == o.with_vol && kind == o.kind &&
           with_pairs == o.with_pairs && with_cvol == o.with_cvol && with_px8 == o.with_px8;
  }
};
enum { kKindForeign = 0, kKindGrd = 1, kKindCen = 2, kKindImg = 3 };

struct cspm_ctx {
  int device = 0, ncu = 256;
  // persistent sweep: workgroups launched per CU (env CSPM_SWEEP_WG).  2..6 take the same time when the pair is alone (the sweep
  // is bound by its dependency chain; 1 is 43 % slower); the resident workgroups mostly wait, and every one of them holds
  // registers another pair's refinement could use: with three pairs in flight 2 gives 222.7 ms per pair, 3 gives 228.1.
  int sweep_wg_per_cu = 0;           // 0 = the default, 2: a CU evaluates two sweep pixels at full speed -- with three resident AND computing (the dataflow sweep) a pixel takes 23 us instead of 9
  int sweep_bands = 1;               // row bands of the persistent sweep (env CSPM_SWEEP_BANDS, up to 8; 1 = one queue for the whole image: the default,
                                     // bands help only the paired-cell volumes)
  int sweep_bands_built = 0;         // what d_sweep_start was filled for
  int refine_chunk = 64;  // PlaneRefinement halving steps per launch (tuning knob, env CSPM_REFINE_CHUNK)
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;
  // images
  int W = 0, H = 0;
  uint32_t *img0[2] = {nullptr, nullptr};
  uint8_t *stage = nullptr;  // host-image staging buffer (cspm_set_images), 3*W*H bytes, kept
  size_t stage_bytes = 0;
  // cost object
  bool cost_alloc = false, cost_ready = false;
  Cost cost{};
  int max_dis = 0, wnd = 0;
  double scale_wgt[CSPM_MAX_LEVELS] = {0};
  double host_max_cost[2 * CSPM_MAX_LEVELS] = {0};
  std::vector<void *> cost_allocs;
  CostKey cost_key;
  bool max_cost_fetched = false;  // host_max_cost mirrors d_maxcost
  int *d_early_ok = nullptr;      // device flag: every max_cost (and, for uploaded volumes, every min) is >= 0
  uint8_t *cen_gray[2][CSPM_MAX_LEVELS] = {{nullptr}};
  double *d_lut = nullptr, *d_lut_a = nullptr, *d_maxcost = nullptr;
  bool is_grd = false;           // cost built by cspm_build_cost_grd (gradients present)
  bool is_cen = false;           // cost built by cspm_build_cost_cen (census codes present)
  bool is_img = false;           // cost built by cspm_build_cost_img (GrdPC / CSPC: no cells, no volumes)
  const uint32_t *cen_code[2][CSPM_MAX_LEVELS] = {{nullptr}};
  long long opt_grd_volumes = 0; // CSPM_OPT_GRD_VOLUMES
  long long opt_sweep_pairs = 0;   // CSPM_OPT_SWEEP_PAIRS: 0 = never (default: measured no faster), 1 = when they fit
  long long sweep_pairs_limit = 4LL << 30;  // bytes of paired-cell volumes a context may hold (env CSPM_SWEEP_PAIRS_MAX_MB)
  bool sweep_pairs = false;      // this cost object carries Level::vol2: the raster sweep reads paired cells (kSrcVol2)
  long long opt_table_volumes = 1;          // CSPM_OPT_TABLE_VOLUMES: device-cell volumes for the row engine's DMA-filled tables, when they fit
  long long table_volumes_limit = 48LL << 30; // bytes of such volumes a context may hold (env CSPM_TABLE_VOLUMES_MAX_MB): 288 GB of HBM per GPU, a few contexts in flight
  long long opt_sweep_packed = 0;           // CSPM_OPT_SWEEP_PACKED: 1 = the raster sweep of a fused GRD cost reads packed 8-byte elements (kSrcGrd8); 0 (default: measured
                                            // 8 % slower -- the sweep is latency-bound and the unpacking adds VALU work to every step) = the 12-byte elements
  bool sweep_packed = false;                // this cost object carries Level::px8
  unsigned int *d_px8_bad = nullptr;        // device counter: gradients k_make_px8 could not pack (must stay 0)
  double volumes_mem_fraction = 0.5;        // of the memory hipMemGetInfo reports free when a cost object is allocated, the share the optional volumes (cvol, vol2) may take (env CSPM_VOLUMES_MEM_FRACTION)
  long long optional_volume_fallbacks = 0;  // times a hipMalloc of an optional volume failed and the pair went on without (CSPM_OPT_VOLUME_FALLBACKS)
  bool optional_missing = false;            // the current cost object wanted optional volumes and runs without them
  long long optional_reuses = 0;            // pairs that reused it since
  long long volume_retry_pairs = 16;        // CSPM_OPT_VOLUME_RETRY_PAIRS: ask again for the volumes every so many reuses (0 = never)
  int fault_volume_alloc = 0;               // fault injection for the tests: the n-th optional-volume allocation of this context fails (CSPM_OPT_FAULT_VOLUME_ALLOC, a test hook: nothing in the environment reaches it)
  unsigned long long *d_maxkeys = nullptr;
  int row_claim = -1;  // row kernels: -1 = claimed column bands for launches of several rounds (default), 0 / 1 = never / always (env CSPM_ROW_CLAIM, tests)
  unsigned int *d_rowq = nullptr;  // row kernels: the eight claim counters of a launch that claims its items (cspm_rows.h row_item)
  // plane field
  bool field_alloc = false;
  bool field_consistent = false;  // every min_cost was computed from the stored plane by this cost object (not by cspm_set_planes)
  double *field_mem = nullptr;
  Field f[2]{};
  ViewCand vc{nullptr, nullptr, nullptr, nullptr};
  long long opt_sweep_fold = 0;  // CSPM_OPT_SWEEP_FOLD: cross-scale sweep workgroups of levels - 1 waves, the last level folded onto them (for contexts that share their GPU)
  long long opt_view_sort = 1;  // CSPM_OPT_VIEW_SORT: view propagation evaluates a row's proposals in the order of their target column
  uint8_t *d_dis[2] = {nullptr, nullptr};
  uint8_t *d_valid[2] = {nullptr, nullptr};  // post-processing: left-right consistency flags
  unsigned int *d_todo = nullptr;            // post-processing: per view n indices of inconsistent pixels, then the two counts
  // persistent raster sweep (k_spatial_sweep)
  unsigned int *d_sweep_ctrl = nullptr, *d_sweep_start = nullptr;
  unsigned long long *d_sweep_gran = nullptr;  // persistent sweep: 12 data-tagged granules per pixel and view (cspm_chain.h)
  unsigned int *d_sweep_ready = nullptr, *d_sweep_qctl = nullptr;  // dataflow sweep (k_spatial_flow): final predecessors per pixel; queue control words
  unsigned long long *d_sweep_queue = nullptr;                      // ... and its queue of ready pixels
  long long opt_sweep_flow = 0;  // CSPM_OPT_SWEEP_FLOW: 1 = the persistent raster sweep is scheduled by dataflow (k_spatial_flow), 0 (default: measured faster) = by ordered claims (k_spatial_sweep)
  unsigned int sweep_epoch = 0;
  long long opt_raster_launches = 0;  // CSPM_OPT_RASTER_LAUNCHES
  long long sweep_timeout_ms = 3000;  // CSPM_OPT_SWEEP_TIMEOUT_MS (env CSPM_SWEEP_TIMEOUT_MS): bound of one wait for a predecessor pixel
  bool sweep_pending = false;         // a sweep's error word has not been checked yet
  // what ran since the sweep error word was last looked at: exactly one whole cspm_patchmatch (on inputs that are still in
  // place) can be repeated with per-diagonal launches when its persistent sweep timed out
  int pm_runs_unchecked = 0;
  bool phases_unchecked = false;      // single phases / cspm_set_planes / new inputs since then: no transparent retry
  int last_iters = 0;
  cspm_pm_params last_params{};
  long long sweep_fallbacks = 0;      // how often that happened (cspm_get_option)
  // asynchronous outputs (cspm_disparity_u8_device / cspm_postprocess_device) enqueued behind a run whose sweep has not been checked
  // yet: when that run is repeated after a timeout they are produced again from the repeated run's planes
  struct OutReq { int post, view, dis_scale; void *o0, *o1; };
  std::vector<OutReq> out_reqs;
  // a later request for the same kind of map into the same buffer replaces the earlier one (the buffer ends up holding the later map)
  void remember_output(const OutReq &q) {
    for (auto it = out_reqs.begin(); it != out_reqs.end(); ++it)
      if (it->post == q.post && it->o0 == q.o0 && it->o1 == q.o1 && (q.post || it->view == q.view)) { out_reqs.erase(it); break; }
    out_reqs.push_back(q);
  }
  // CSPatchMatch over a foreign IPlaneCost (cspm_fpm_*): candidate buffers and what the pending batch was
  FpmCand fpm{nullptr, nullptr, nullptr, nullptr};
  long long fpm_cap = 0;
  int fpm_phase = -1, fpm_iter = 0, fpm_step = 0, fpm_inc = 1;
  long long fpm_count = 0;
  cspm_pm_params fpm_params{};
  // timing
  bool timing = false;
  std::vector<TimingRec> recs;
  std::vector<hipEvent_t> pool;
  double acc_ms[CSPM_K_COUNT] = {0};
  long long acc_launch[CSPM_K_COUNT] = {0}, acc_evals[CSPM_K_COUNT] = {0};
};

namespace {

int fail(cspm_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIPCHK(ctx, expr)                                                                            \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return fail(ctx, CSPM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
  } while (0)

template <class T>
int dalloc(cspm_ctx *c, T **p, size_t n, std::vector<void *> *track) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : sizeof(T));
  if (e != hipSuccess) return fail(c, CSPM_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  *p = (T *)q;
  if (track) track->push_back(q);
  return CSPM_OK;
}

hipEvent_t get_event(cspm_ctx *c) {
  if (!c->pool.empty()) {
    hipEvent_t e = c->pool.back();
    c->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

// bracket a launch with events when timing is on
struct Timed {
  cspm_ctx *c;
  TimingRec r{};
  bool on;
  Timed(cspm_ctx *ctx, int kclass, long long evals) : c(ctx), on(ctx->timing) {
    if (!on) return;
    r.kclass = kclass;
    r.evals = evals;
    r.a = get_event(c);
    r.b = get_event(c);
    (void)hipEventRecord(r.a, c->stream);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(r.b, c->stream);
    c->recs.push_back(r);
  }
};

int drain_timing(cspm_ctx *c) {
  if (c->recs.empty()) return CSPM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto &r : c->recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    c->acc_ms[r.kclass] += ms;
    c->acc_launch[r.kclass] += 1;
    c->acc_evals[r.kclass] += r.evals;
    c->pool.push_back(r.a);
    c->pool.push_back(r.b);
  }
  c->recs.clear();
  return CSPM_OK;
}

// row engine: one wave per 64-pixel run of an image row, kRowWaves waves per workgroup, grid a multiple of 8 (XCD bands)
// a row-kernel launch: claimed column bands when it runs for several rounds of resident waves, interleaved row blocks otherwise
// (cspm_rows.h row_item); workgroups of kRowWaves waves, grid a multiple of 8
inline bool row_claimed_w(const cspm_ctx *c, int W, int views) {
  const long long items = row_items(W, c->H, views);
  if (items >= (1LL << 31)) return false;
  if (c->row_claim >= 0) return c->row_claim != 0;
  return items >= 2LL * c->ncu * 12;  // two rounds of the 12 waves a CU holds (a KITTI-size pair: every row kernel; a 450 x 375 pair: none)
}
inline unsigned row_grid_w(const cspm_ctx *c, int W, int views) {
  const bool claimed = row_claimed_w(c, W, views);
  long long per_xcd = (row_items_per_xcd(W, c->H, views, claimed) + kRowWaves - 1) / kRowWaves;
  if (claimed) per_xcd += per_xcd / 4 + 1;  // the surplus workgroups of the XCDs that finish first take over the others' bands
  return (unsigned)(per_xcd * 8);
}
inline bool row_claimed(const cspm_ctx *c, int views) { return row_claimed_w(c, c->W, views); }
inline unsigned row_grid(const cspm_ctx *c, int views) { return row_grid_w(c, c->W, views); }
inline int row_cap(const cspm_ctx *c) { return strip_capacity(c->max_dis, c->cost.half); }
inline int row_ocap(const cspm_ctx *c) { return own_capacity(c->cost.half); }
inline size_t row_shmem(const cspm_ctx *c) { return sizeof(LutMem) + (size_t)kRowWaves * wave_lds_bytes(row_cap(c), row_ocap(c)); }

// the claim counters of the next row-kernel launch, zeroed on the stream right before it (launches of a context are serial on its
// stream: one set suffices); none for a launch of interleaved row blocks
inline RowQueue next_row_queue(cspm_ctx *c, int views) {
  if (!row_claimed(c, views)) return RowQueue{nullptr};
  (void)hipMemsetAsync(c->d_rowq, 0, 8 * sizeof(unsigned int), c->stream);
  return RowQueue{c->d_rowq};
}

inline unsigned eval_grid(long long items) {
  long long nb = (items + (kEvalBlock / kWave) - 1) / (kEvalBlock / kWave);
  nb = (nb + 7) / 8 * 8;
  if (nb < 8) nb = 8;
  return (unsigned)nb;
}
inline unsigned ew_grid(long long n, int block = 256) { return (unsigned)((n + block - 1) / block); }
inline unsigned stride_grid(long long n, int block = 256) { return (unsigned)std::min<long long>((n + block - 1) / block, 256 * 16); }

void free_cost(cspm_ctx *c) {
  for (void *p : c->cost_allocs) (void)hipFree(p);
  c->cost_allocs.clear();
  c->cost_alloc = c->cost_ready = false;
  c->cost_key = CostKey{};
  memset(&c->cost, 0, sizeof c->cost);
}
void free_field(cspm_ctx *c) {
  if (c->field_mem) (void)hipFree(c->field_mem);
  if (c->vc.cost) (void)hipFree(c->vc.cost);
  if (c->vc.c) (void)hipFree(c->vc.c);
  if (c->vc.cx) (void)hipFree(c->vc.cx);
  if (c->vc.perm) (void)hipFree(c->vc.perm);
  for (int v = 0; v < 2; ++v) {
    if (c->d_dis[v]) (void)hipFree(c->d_dis[v]);
    if (c->d_valid[v]) (void)hipFree(c->d_valid[v]);
    c->d_dis[v] = nullptr;
    c->d_valid[v] = nullptr;
  }
  if (c->d_todo) (void)hipFree(c->d_todo);
  c->d_todo = nullptr;
  if (c->d_rowq) (void)hipFree(c->d_rowq);
  c->d_rowq = nullptr;
  if (c->fpm.xy) (void)hipFree(c->fpm.xy);
  if (c->fpm.view) (void)hipFree(c->fpm.view);
  if (c->fpm.plane) (void)hipFree(c->fpm.plane);
  if (c->fpm.cost) (void)hipFree(c->fpm.cost);
  c->fpm = FpmCand{nullptr, nullptr, nullptr, nullptr};
  c->fpm_cap = 0;
  c->fpm_phase = -1;
  if (c->d_sweep_ctrl) (void)hipFree(c->d_sweep_ctrl);
  if (c->d_sweep_gran) (void)hipFree(c->d_sweep_gran);
  if (c->d_sweep_start) (void)hipFree(c->d_sweep_start);
  if (c->d_sweep_ready) (void)hipFree(c->d_sweep_ready);
  if (c->d_sweep_qctl) (void)hipFree(c->d_sweep_qctl);
  if (c->d_sweep_queue) (void)hipFree(c->d_sweep_queue);
  c->d_sweep_ready = c->d_sweep_qctl = nullptr;
  c->d_sweep_queue = nullptr;
  c->d_sweep_ctrl = c->d_sweep_start = nullptr;
  c->d_sweep_gran = nullptr;
  c->field_mem = nullptr;
  c->vc = ViewCand{nullptr, nullptr, nullptr, nullptr};
  c->field_alloc = false;
}
void free_images(cspm_ctx *c) {
  for (int v = 0; v < 2; ++v) {
    if (c->img0[v]) (void)hipFree(c->img0[v]);
    c->img0[v] = nullptr;
  }
  if (c->stage) (void)hipFree(c->stage);
  c->stage = nullptr;
  c->stage_bytes = 0;
  c->W = c->H = 0;
}

void small_inverse_row0(int S, const double A[CSPM_MAX_LEVELS][CSPM_MAX_LEVELS], double *w);
// pre_cs_pc.cc:86-109: scale_wgt[s] = inv(tridiag(lambda))(0,s); Mat::inv() = cv::invert(DECOMP_LU): closed form for
// n <= 3, otherwise LU with partial pivoting and reciprocal pivots (OpenCV 2.4 LUImpl), identity right-hand side.
int scale_weights(int S, double lambda, double *w) {
  if (S < 1 || S > CSPM_MAX_LEVELS) return -1;
  double A[CSPM_MAX_LEVELS][CSPM_MAX_LEVELS] = {{0}}, B[CSPM_MAX_LEVELS][CSPM_MAX_LEVELS] = {{0}};
  for (int s = 0; s < S; ++s) {
    B[s][s] = 1.0;
    if (S == 1) { A[0][0] = 1 + lambda; break; }
    if (s == 0) { A[s][s] = 1 + lambda; A[s][s + 1] = -lambda; }
    else if (s == S - 1) { A[s][s] = 1 + lambda; A[s][s - 1] = -lambda; }
    else { A[s][s] = 1 + 2 * lambda; A[s][s - 1] = -lambda; A[s][s + 1] = -lambda; }
  }
  if (S <= 3) {  // cv::invert's closed-form path
    small_inverse_row0(S, A, w);
    return 0;
  }
  const double eps = DBL_EPSILON * 100;
  for (int i = 0; i < S; ++i) {
    int k = i;
    for (int j = i + 1; j < S; ++j)
      if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
    if (std::fabs(A[k][i]) < eps) return -2;
    if (k != i) {
      for (int j = i; j < S; ++j) std::swap(A[i][j], A[k][j]);
      for (int j = 0; j < S; ++j) std::swap(B[i][j], B[k][j]);
    }
    const double d = -1 / A[i][i];
    for (int j = i + 1; j < S; ++j) {
      const double alpha = A[j][i] * d;
      for (int q = i + 1; q < S; ++q) A[j][q] += alpha * A[i][q];
      for (int q = 0; q < S; ++q) B[j][q] += alpha * B[i][q];
    }
    A[i][i] = -d;
  }
  for (int i = S - 1; i >= 0; --i)
    for (int j = 0; j < S; ++j) {
      double s = B[i][j];
      for (int q = i + 1; q < S; ++q) s -= A[i][q] * B[q][j];
      B[i][j] = s * A[i][i];
    }
  for (int s = 0; s < S; ++s) w[s] = B[0][s];
  return 0;
}

// cv::invert(DECOMP_LU) of OpenCV 2.4 takes a closed-form path for n <= 3 (det2 / det3 and cofactors times 1/det);
// only the first row of the inverse is needed (pre_cs_pc.cc:105-108).  Restated from the OpenCV 2.4 sources
// from memory -- not verifiable in this image.
void small_inverse_row0(int S, const double A[CSPM_MAX_LEVELS][CSPM_MAX_LEVELS], double *w) {
  if (S == 1) {
    w[0] = 1. / A[0][0];
  } else if (S == 2) {
    double d = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    d = 1. / d;
    w[0] = A[1][1] * d;
    w[1] = -A[0][1] * d;
  } else {
    double d = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
               A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    d = 1. / d;
    w[0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) * d;
    w[1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * d;
    w[2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * d;
  }
}

// (re)run the pyramid of both views into the level images (pre_cs_pc.cc:36-55)
void launch_pyramid(cspm_ctx *c) {
  Cost &cd = c->cost;
  for (int s = 0; s < cd.levels; ++s) {
    Level &L = cd.lv[s];
    for (int v = 0; v < 2; ++v) {
      Timed t(c, CSPM_K_MISC, 0);
      uint32_t *img = const_cast<uint32_t *>(L.pix[v]);
      if (s == 0) {
        hipLaunchKernelGGL(k_pad_u32, dim3(ew_grid((long long)L.Wp * L.H)), dim3(256), 0, c->stream, c->img0[v], L.W, L.H, L.Wp, L.pad, img);
      } else {
        const Level &P = cd.lv[s - 1];
        hipLaunchKernelGGL(k_pyrdown, dim3(ew_grid((long long)L.Wp * L.H)), dim3(256), 0, c->stream, P.pix[v], P.W, P.H, P.Wp, P.pad,
                           img, L.W, L.H, L.Wp, L.pad);
      }
    }
  }
}

// allocate the (padded) pyramid images, the per-kind side arrays (gradients / census codes) and, when `with_vol`, the
// cost volumes; fill Cost (everything except gradients / volume contents / max_cost).  An identical request (same
// image size, max_dis, window, levels, kind, volumes) reuses every buffer: no allocator call, no host synchronisation.
int alloc_cost(cspm_ctx *c, int max_dis, int wnd_size, int scale_num, double reg_lambda, bool with_vol, int kind, bool want_pairs = false,
               bool want_cvol = false) {
  if (!c->img0[0]) return fail(c, CSPM_ERR_STATE, "cspm_set_images must precede cost construction");
  if (max_dis < 1 || wnd_size < 1 || wnd_size > kMaxWnd || scale_num < 0 || scale_num > CSPM_MAX_LEVELS)
    return fail(c, CSPM_ERR_ARG, "bad max_dis / wnd_size / scale_num");
  // paired-cell volumes for the raster sweep (kSrcVol2): only when every level's indices fit the sweep's 28-bit element offsets
  // and 24-bit slab size and the whole set stays under the context's budget (C3: 2.2 GB; C5 would need 56 GB and keeps the fused sweep)
  bool with_pairs = false;
  long long pairs_bytes = 0, cvol_bytes = 0;
  if (want_pairs) {
    long long bytes = 0;
    bool fits = true;
    int W = c->W, H = c->H, D = max_dis;
    for (int s = 0; s < (scale_num > 0 ? scale_num : 1); ++s) {
      if (s > 0) { H = (H + 1) / 2; W = (W + 1) / 2; D = D / 2; }
      const long long slab = (long long)W * H;
      if (slab >= (1LL << 24) || slab * std::max(D, 1) >= (1LL << 28)) fits = false;
      bytes += 2 * slab * std::max(D, 2) * 16;
    }
    with_pairs = fits && bytes <= c->sweep_pairs_limit;
    pairs_bytes = bytes;
  }
  // device-cell volumes for the row engine's DMA-filled tables: C3 1.2 GB, a 3000 x 2000 D = 256 pair 30 GB (each level's volume
  // must stay below 4 GiB per 16 disparities: the DMA's 32-bit offsets span the slabs of one table)
  bool with_cvol = false;
  const int cvpad_all = wnd_size / 2 + 2;
  if (want_cvol) {
    long long bytes = 0;
    bool fits32 = true;
    int W = c->W, H = c->H, D = max_dis;
    for (int s = 0; s < (scale_num > 0 ? scale_num : 1); ++s) {
      if (s > 0) { H = (H + 1) / 2; W = (W + 1) / 2; D = D / 2; }
      bytes += 2LL * (D + 1) * H * (W + 2 * cvpad_all) * 8;
      if ((long long)H * (W + 2 * cvpad_all) * 8 * std::min(D + 1, 64) >= (1LL << 32)) fits32 = false;  // the usual tables of <= 64 slabs within 32-bit offsets (the device checks every table's real span: cspm_rows.h span32)
    }
    with_cvol = fits32 && bytes <= c->table_volumes_limit;
    cvol_bytes = bytes;
  }
  CostKey key;
  key.W = c->W; key.H = c->H; key.max_dis = max_dis; key.wnd = wnd_size; key.scale_num = scale_num; key.with_vol = with_vol; key.kind = kind;
  key.with_pairs = with_pairs;
  key.with_cvol = with_cvol;
  const bool with_px8 = kind == kKindGrd && !with_vol && c->opt_sweep_packed != 0;
  key.with_px8 = with_px8;
  bool reuse = c->cost_alloc && key == c->cost_key;
  // A cost object that wanted optional volumes and did not get them (free-memory veto, failed hipMalloc) is reused as it is, but not for
  // ever: every volume_retry_pairs-th reuse allocates afresh and asks again, so that one transient shortage -- another context was
  // building its own volumes at that moment -- does not leave this context on the slower path until its geometry changes.
  if (reuse && c->optional_missing && c->volume_retry_pairs > 0 && ++c->optional_reuses >= c->volume_retry_pairs) reuse = false;
  if (!reuse) {
    free_cost(c);
    c->optional_reuses = 0;
    c->optional_missing = false;
  }
  Cost &cd = c->cost;
  c->cost_ready = false;
  c->max_cost_fetched = false;
  c->field_consistent = false;  // stored min_costs belong to the previous cost object: only InitRandomPlane re-establishes them
  if (c->pm_runs_unchecked) c->phases_unchecked = true;  // an unchecked run's inputs are being replaced: no transparent retry for it
  int rc;
  if (!reuse) {
    cd.cs = scale_num > 0;
    cd.levels = cd.cs ? scale_num : 1;
    cd.half = wnd_size / 2;
    cd.n = 2 * cd.half + 1;
    cd.T = cd.n * cd.n;
    c->max_dis = max_dis;
    c->wnd = wnd_size;
    // pre_cs_pc.cc:36-55
    int W = c->W, H = c->H, D = max_dis;
    for (int s = 0; s < cd.levels; ++s) {
      if (s > 0) { H = (H + 1) / 2; W = (W + 1) / 2; D = D / 2; }
      Level &L = cd.lv[s];
      L.W = W; L.H = H; L.D = D;
      L.pad = D + cd.half + 8;  // cspm_device.h: window overrun and disparity range stay inside the padding
      L.Wp = W + 2 * L.pad;
      if ((long long)L.Wp * H * 12 >= (1LL << 31) || L.Wp * 12 >= (1 << 23))
        return fail(c, CSPM_ERR_ARG, "image too large for the 32-bit / 24-bit element offsets of the tap engine");
      const size_t px = (size_t)W * H, ppx = (size_t)L.Wp * H;
      for (int v = 0; v < 2; ++v) {
        uint32_t *img;
        if ((rc = dalloc(c, &img, ppx, &c->cost_allocs))) return rc;
        PixG *pxg;
        if ((rc = dalloc(c, &pxg, ppx, &c->cost_allocs))) return rc;
        L.px[v] = pxg;
        L.px16[v] = nullptr;
        L.px8[v] = nullptr;
        L.pc[v] = nullptr;
        L.pix[v] = img;
        L.grd[v] = nullptr;
        L.vol[v] = nullptr;
        L.vol2[v] = nullptr;
        L.cvol[v] = nullptr;
        L.cvpad = cvpad_all;
        L.cvW = W + 2 * cvpad_all;
        if (with_vol) {
          double *vol;
          if ((rc = dalloc(c, &vol, (size_t)(D + 2) * px, &c->cost_allocs))) return rc;  // D+1 slabs and one guard slab (clamped taps of a level with D < 2)
          L.vol[v] = vol;
        }
        if (kind == kKindGrd) {
          double *g;
          if ((rc = dalloc(c, &g, ppx + 64, &c->cost_allocs))) return rc;  // + slack: a strip's last DMA piece may start inside the last row
          L.grd[v] = g;
          uint4 *p16;
          if ((rc = dalloc(c, &p16, ppx + 64, &c->cost_allocs))) return rc;
          L.px16[v] = p16;
          if (with_px8) {
            Pix8 *p8;
            if ((rc = dalloc(c, &p8, ppx + 64, &c->cost_allocs))) return rc;  // + slack: a pair load at the last element reads one element beyond
            L.px8[v] = p8;
          }
        } else if (kind == kKindImg) {
          uint8_t *gray;
          if ((rc = dalloc(c, &gray, px, &c->cost_allocs))) return rc;
          c->cen_gray[v][s] = gray;
        } else if (kind == kKindCen) {
          uint8_t *gray;
          uint32_t *code;
          PixC *pc;
          if ((rc = dalloc(c, &gray, px, &c->cost_allocs)) || (rc = dalloc(c, &code, px * 3, &c->cost_allocs)) ||
              (rc = dalloc(c, &pc, ppx, &c->cost_allocs)))
            return rc;
          c->cen_gray[v][s] = gray;
          c->cen_code[v][s] = code;
          L.pc[v] = pc;
        }
      }
    }
    // The OPTIONAL volumes come last: device-cell volumes for the DMA-filled tables and paired-cell volumes for the sweep are accelerators,
    // never a reason to fail a pair.  They are taken only from memory that is free NOW (at most `volumes_mem_fraction` of it: the
    // plane field, the sweep granules and the other contexts of the process still have to fit), and a hipMalloc that fails all the
    // same releases what this pass took and the pair runs with computed tables / the fused sweep -- the results are identical.
    if (with_cvol || with_pairs) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const long long field_b = (long long)c->W * c->H * 2 * (14 + 3 + kGranPerPixel + 2) * 8;  // ensure_field's arrays, if not there yet
        long long avail = (long long)((double)free_b * c->volumes_mem_fraction) - (c->field_alloc ? 0 : field_b);
        if (with_cvol && cvol_bytes > avail) with_cvol = false;
        if (with_cvol) avail -= cvol_bytes;
        if (with_pairs && pairs_bytes > avail) with_pairs = false;
      }
    }
    const size_t mandatory = c->cost_allocs.size();
    auto drop_optional = [&]() {  // give back whatever this pass allocated and run without either kind of volume
      for (size_t i = mandatory; i < c->cost_allocs.size(); ++i) (void)hipFree(c->cost_allocs[i]);
      c->cost_allocs.resize(mandatory);
      for (int s = 0; s < cd.levels; ++s)
        for (int v = 0; v < 2; ++v) cd.lv[s].cvol[v] = nullptr, cd.lv[s].vol2[v] = nullptr;
      (void)hipGetLastError();  // the out-of-memory error is handled: do not leave it as the runtime's sticky last error
      c->err.clear();
      c->optional_volume_fallbacks++;
    };
    for (int s = 0; s < cd.levels && (with_cvol || with_pairs); ++s) {
      Level &L = cd.lv[s];
      const size_t px = (size_t)L.W * L.H;
      for (int v = 0; v < 2 && (with_cvol || with_pairs); ++v) {
        if (with_cvol) {
          double *cv;
          const size_t ncv = (size_t)(L.D + 1) * L.H * L.cvW + 64;  // + slack: the last DMA piece of the last row may run 8 bytes over
          if ((c->fault_volume_alloc > 0 && --c->fault_volume_alloc == 0) || dalloc(c, &cv, ncv, &c->cost_allocs) != CSPM_OK) { drop_optional(); with_cvol = with_pairs = false; break; }
          HIPCHK(c, hipMemsetAsync(cv, 0, ncv * sizeof(double), c->stream));  // the pad columns stay 0.0; the image columns are rewritten per pair
          L.cvol[v] = cv;
        }
        if (with_pairs) {
          double2 *v2;  // slabs 0 .. D-1; a level with D < 2 is only ever addressed (slab 1), never used
          if ((c->fault_volume_alloc > 0 && --c->fault_volume_alloc == 0) || dalloc(c, &v2, (size_t)std::max(L.D, 2) * px, &c->cost_allocs) != CSPM_OK) { drop_optional(); with_cvol = with_pairs = false; break; }
          L.vol2[v] = v2;
        }
      }
    }
    c->optional_missing = (key.with_cvol && !with_cvol) || (key.with_pairs && !with_pairs);  // wanted, not held: asked for again later
    double lut[2 * kLutSize];
    for (int i = 0; i < kLutSize; ++i) {
      lut[i] = std::exp(-i * 1.0 / 10.0);  // WGT_GAMMA, pre_cs_pc.h:16
      double clrDiff = (double)i;          // sum of three |lC-rC|, an exact integer
      clrDiff *= 0.3333333333;
      clrDiff = clrDiff > 10.0 ? 10.0 : clrDiff;  // TAU_CLR
      lut[kLutSize + i] = 0.1 * clrDiff;   // ALPHA * clrDiff
    }
    if ((rc = dalloc(c, &c->d_lut, 2 * kLutSize, &c->cost_allocs))) return rc;
    c->d_lut_a = c->d_lut + kLutSize;
    if ((rc = dalloc(c, &c->d_maxcost, 2 * CSPM_MAX_LEVELS, &c->cost_allocs))) return rc;
    if ((rc = dalloc(c, &c->d_early_ok, 1, &c->cost_allocs))) return rc;
    if ((rc = dalloc(c, &c->d_px8_bad, 1, &c->cost_allocs))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_px8_bad, 0, sizeof(unsigned int), c->stream));
    if ((rc = dalloc(c, &c->d_maxkeys, 4 * CSPM_MAX_LEVELS, &c->cost_allocs))) return rc;
    HIPCHK(c, hipMemcpy(c->d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    cd.lut = c->d_lut;
    cd.lut_a = c->d_lut_a;
    cd.max_cost = c->d_maxcost;
    cd.early_ok = c->d_early_ok;
    c->cost_key = key;
    c->cost_alloc = true;
  }
  launch_pyramid(c);
  HIPCHK(c, hipMemsetAsync(c->d_px8_bad, 0, sizeof(unsigned int), c->stream));  // per cost object, not per allocation
  // scale weights (pre_cs_pc.cc:86-109): host-side, per call (reg_lambda is not part of the buffer key)
  if (cd.cs) {
    if (scale_weights(cd.levels, reg_lambda, c->scale_wgt)) return fail(c, CSPM_ERR_ARG, "singular regularisation matrix");
  } else {
    c->scale_wgt[0] = 1.0;
  }
  for (int s = 0; s < cd.levels; ++s) cd.lv[s].wgt = c->scale_wgt[s];
  // max keys [0, 2L) start at 0 (= below every key), min keys [2L, 4L) at all-ones
  HIPCHK(c, hipMemsetAsync(c->d_maxkeys, 0, sizeof(unsigned long long) * 2 * CSPM_MAX_LEVELS, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_maxkeys + 2 * CSPM_MAX_LEVELS, 0xFF, sizeof(unsigned long long) * 2 * CSPM_MAX_LEVELS, c->stream));
  cd.fused = kSrcVolume;
  c->is_grd = false;
  c->is_cen = false;
  c->is_img = false;
  c->sweep_pairs = false;
  c->sweep_packed = false;
  return CSPM_OK;
}

// max_cost of every level/view from the reduced keys and the early-exit licence, all on the device: no host round trip.
// The early-exit proof (cspm_kernels.h level_cost) needs every term of the sum to be >= 0: scale weights (checked here),
// cell costs (GRD and census cells are >= 0 by construction; an uploaded volume is checked through its reduced MIN).
int finish_cost(cspm_ctx *c, bool check_min, double floor_val = -1.0) {
  Cost &cd = c->cost;
  int wgt_ok = 1;
  for (int s = 0; s < cd.levels; ++s)
    if (!(c->scale_wgt[s] >= 0.0)) wgt_ok = 0;
  hipLaunchKernelGGL(k_finish_cost, dim3(1), dim3(64), 0, c->stream, c->d_maxkeys, c->d_maxcost, 2 * CSPM_MAX_LEVELS, cd.levels, floor_val,
                     wgt_ok, check_min ? 1 : 0, c->d_early_ok);
  HIPCHK(c, hipGetLastError());
  c->cost_ready = true;
  return CSPM_OK;
}

int fetch_max_cost(cspm_ctx *c) {
  if (c->max_cost_fetched) return CSPM_OK;
  HIPCHK(c, hipMemcpyAsync(c->host_max_cost, c->d_maxcost, sizeof c->host_max_cost, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->max_cost_fetched = true;
  return CSPM_OK;
}

int ensure_field(cspm_ctx *c) {
  if (c->field_alloc) return CSPM_OK;
  const size_t n = (size_t)c->W * c->H;
  int rc;
  if ((rc = dalloc(c, &c->field_mem, 14 * n, nullptr))) return rc;
  for (int v = 0; v < 2; ++v) {
    double *base = c->field_mem + (size_t)v * 7 * n;
    c->f[v] = Field{base, base + n, base + 2 * n, base + 3 * n, base + 4 * n, base + 5 * n, base + 6 * n};
  }
  if ((rc = dalloc(c, &c->vc.cost, n, nullptr))) return rc;
  if ((rc = dalloc(c, &c->vc.c, n, nullptr))) return rc;
  if ((rc = dalloc(c, &c->vc.cx, n, nullptr))) return rc;
  if ((rc = dalloc(c, &c->vc.perm, n, nullptr))) return rc;
  for (int v = 0; v < 2; ++v) {
    if ((rc = dalloc(c, &c->d_dis[v], n, nullptr))) return rc;
    if ((rc = dalloc(c, &c->d_valid[v], n, nullptr))) return rc;
  }
  if ((rc = dalloc(c, &c->d_todo, 2 * n + 2, nullptr))) return rc;
  if ((rc = dalloc(c, &c->d_rowq, 8, nullptr))) return rc;
  // persistent sweep state: control words, per-pixel granules (tag zero = never written), diagonal start table
  if ((rc = dalloc(c, &c->d_sweep_ctrl, 2 + kSweepMaxBands, nullptr))) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_sweep_ctrl, 0, (2 + kSweepMaxBands) * sizeof(unsigned int), c->stream));
  if ((rc = dalloc(c, &c->d_sweep_gran, 2 * n * kGranPerPixel, nullptr))) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_sweep_gran, 0, sizeof(unsigned long long) * 2 * n * kGranPerPixel, c->stream));
  c->sweep_epoch = 0;
  {
    // per row band b (sweep rows [b*H/nb, (b+1)*H/nb)): start[k] = the band's items (both views) on anti-diagonals < k
    const int nd = c->W + c->H - 1;
    const int nb = std::max(1, std::min(std::min(c->sweep_bands, kSweepMaxBands), c->H));
    std::vector<unsigned int> start((size_t)nb * (nd + 1), 0);
    for (int b = 0; b < nb; ++b) {
      const int y0 = (int)((long long)b * c->H / nb), y1 = (int)((long long)(b + 1) * c->H / nb);
      unsigned int *st = start.data() + (size_t)b * (nd + 1);
      for (int k = 0; k < nd; ++k) {
        const int cnt = std::min(y1 - 1, k) - std::max(y0, k - (c->W - 1)) + 1;
        st[k + 1] = st[k] + 2u * (unsigned)std::max(cnt, 0);
      }
    }
    if ((rc = dalloc(c, &c->d_sweep_start, start.size(), nullptr))) return rc;
    HIPCHK(c, hipMemcpy(c->d_sweep_start, start.data(), start.size() * sizeof(unsigned int), hipMemcpyHostToDevice));
    c->sweep_bands_built = nb;
  }
  c->field_alloc = true;
  return CSPM_OK;
}

// dataflow sweep (CSPM_OPT_SWEEP_FLOW, off by default): predecessor counters (zeroed before every sweep), the queue of ready pixels
// (epoch-tagged entries: never cleared; at most one entry per pixel plus one reserved slot per waiting workgroup) and its control
// words -- 24 bytes per pixel that the default sweep never touches, so they are allocated by the first sweep that wants them
// (free_field releases them).
int ensure_flow(cspm_ctx *c) {
  if (c->d_sweep_ready) return CSPM_OK;
  const size_t n = (size_t)c->W * c->H;
  int rc;
  if ((rc = dalloc(c, &c->d_sweep_ready, 2 * n, nullptr))) return rc;
  if ((rc = dalloc(c, &c->d_sweep_queue, 2 * n + 65536, nullptr))) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_sweep_queue, 0, sizeof(unsigned long long) * (2 * n + 65536), c->stream));
  if ((rc = dalloc(c, &c->d_sweep_qctl, 4, nullptr))) return rc;
  return CSPM_OK;
}

Pm make_pm(cspm_ctx *c, const cspm_pm_params *p) {
  Pm pm{};
  pm.W = c->W; pm.H = c->H; pm.max_dis = c->max_dis;
  pm.seed = p->seed;
  pm.rng_row_shared = p->rng_mode == CSPM_RNG_ROW_SHARED;
  pm.trust_cost = c->field_consistent ? 1 : 0;
  pm.use_thresh = p->early_exit ? 1 : 0;  // and-ed with the cost object's device-side licence (Cost::early_ok) in the kernels
  pm.f[0] = c->f[0];
  pm.f[1] = c->f[1];
  return pm;
}

// A persistent sweep's bounded spins raise the STICKY error word ctrl[1] instead of hanging (later sweeps then drain at
// once).  It is looked at by every call that synchronises with the host anyway (cspm_synchronize, the getters, the
// single-phase entry cspm_pm_spatial): cspm_patchmatch itself stays asynchronous.
int run_patchmatch(cspm_ctx *c, int iter_num, const cspm_pm_params *p);
int enqueue_disp_u8(cspm_ctx *c, int view, int dis_scale, void *d_out);
int enqueue_postprocess_device(cspm_ctx *c, int dis_scale, void *d_l_out, void *d_r_out);

int check_sweep(cspm_ctx *c) {
  if (!c->sweep_pending) return CSPM_OK;
  unsigned int ctrl[2] = {0, 0}, px8_bad = 0;
  HIPCHK(c, hipMemcpyAsync(ctrl, c->d_sweep_ctrl, sizeof ctrl, hipMemcpyDeviceToHost, c->stream));
  const bool packed = c->sweep_packed && c->cost_alloc && c->d_px8_bad;
  if (packed) HIPCHK(c, hipMemcpyAsync(&px8_bad, c->d_px8_bad, sizeof px8_bad, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->sweep_pending = false;
  // CSPM_OPT_SWEEP_PACKED: a gradient the 36-bit fixed-point field cannot hold was packed as 0 and the sweep read a wrong cell.  Cannot
  // happen for 8-bit images (cspm.h); if it ever does, the planes are wrong and the caller must hear about it, not only a counter.
  if (packed && px8_bad) {
    c->pm_runs_unchecked = 0;
    c->phases_unchecked = false;
    c->out_reqs.clear();
    return fail(c, CSPM_ERR_HIP, "CSPM_OPT_SWEEP_PACKED: " + std::to_string(px8_bad) + " gradients could not be packed exactly; the raster sweep's planes are not valid");
  }
  const int runs = c->pm_runs_unchecked;
  const bool phases = c->phases_unchecked;
  c->pm_runs_unchecked = 0;
  c->phases_unchecked = false;
  const std::vector<cspm_ctx::OutReq> reqs = c->out_reqs;
  c->out_reqs.clear();
  if (ctrl[1]) {
    (void)hipMemsetAsync(c->d_sweep_ctrl, 0, 2 * sizeof(unsigned int), c->stream);
    // A timeout is slowness (a shared or oversubscribed GPU, a profiler attached, many contexts in flight), not a wrong
    // result waiting to happen: the per-diagonal sweep needs no inter-workgroup hand-off and gives the same planes bit for
    // bit.  When exactly one whole PatchMatch ran since the last check -- its inputs are still in place -- it is repeated
    // that way and the caller never sees the hiccup; anything else (several pairs enqueued, single phases) is an error.
    if (runs == 1 && !phases && c->cost_ready) {
      const long long keep = c->opt_raster_launches;
      c->opt_raster_launches = 1;
      int rc = run_patchmatch(c, c->last_iters, &c->last_params);
      c->opt_raster_launches = keep;
      c->pm_runs_unchecked = 0;
      // the maps that were enqueued behind the aborted run were computed from its planes: produce them again
      for (const auto &q : reqs) {
        if (rc != CSPM_OK) break;
        rc = q.post ? enqueue_postprocess_device(c, q.dis_scale, q.o0, q.o1) : enqueue_disp_u8(c, q.view, q.dis_scale, q.o0);
      }
      if (rc == CSPM_OK) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ++c->sweep_fallbacks;
        return CSPM_OK;
      }
      return rc;
    }
    return fail(c, CSPM_ERR_HIP, "raster sweep timed out waiting for a predecessor pixel (inter-workgroup hand-off)");
  }
  return CSPM_OK;
}

const cspm_pm_params kDefaultParams = {12345ULL, CSPM_SCHED_RASTER, 1, 4, CSPM_RNG_PER_PIXEL, 1};

int check_pm(cspm_ctx *c, const cspm_pm_params **p) {
  if (!c) return CSPM_ERR_ARG;
  if (!c->cost_ready) return fail(c, CSPM_ERR_STATE, "no plane cost built (cspm_build_cost_grd / cspm_finish_cost)");
  if (!*p) *p = &kDefaultParams;
  if ((*p)->schedule != CSPM_SCHED_RASTER && (*p)->schedule != CSPM_SCHED_REDBLACK) return fail(c, CSPM_ERR_ARG, "bad schedule");
  if ((*p)->rb_neighbours != 2 && (*p)->rb_neighbours != 4) return fail(c, CSPM_ERR_ARG, "rb_neighbours must be 2 or 4");
  if ((*p)->rb_rounds < 1) return fail(c, CSPM_ERR_ARG, "rb_rounds must be >= 1");
  return ensure_field(c);
}

// A launch gets 64 KB of dynamic LDS without asking; a wide disparity range (max_dis > ~290: two strip sets of 384 slots per wave)
// needs a little more -- gfx950 has 160 KB per CU -- and a kernel has to opt in once per instantiation.
template <class K>
inline void allow_lds(K kern, size_t shmem) {
  // always the device's whole 160 KB: the attribute belongs to the kernel, not to the launch, and two host threads (two contexts)
  // that set two different sizes for the same instantiation would race between one's attribute call and its launch
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
#define LAUNCH_ONE(K, grid, block, shmem, ...)                                \
  do {                                                                       \
    allow_lds(K, shmem);                                                     \
    hipLaunchKernelGGL(K, grid, block, shmem, c->stream, __VA_ARGS__);       \
  } while (0)
#define LAUNCH_CS(kern, grid, block, shmem, ...)                                                                  \
  do {                                                                                                            \
    const int src_ = c->cost.fused;                                                                               \
    if (c->cost.cs) {                                                                                             \
      if (src_ == kSrcGrd) LAUNCH_ONE((kern<true, kSrcGrd>), grid, block, shmem, __VA_ARGS__);                    \
      else if (src_ == kSrcCen) LAUNCH_ONE((kern<true, kSrcCen>), grid, block, shmem, __VA_ARGS__);               \
      else if (src_ == kSrcImg) LAUNCH_ONE((kern<true, kSrcImg>), grid, block, shmem, __VA_ARGS__);               \
      else LAUNCH_ONE((kern<true, kSrcVolume>), grid, block, shmem, __VA_ARGS__);                                 \
    } else {                                                                                                      \
      if (src_ == kSrcGrd) LAUNCH_ONE((kern<false, kSrcGrd>), grid, block, shmem, __VA_ARGS__);                   \
      else if (src_ == kSrcCen) LAUNCH_ONE((kern<false, kSrcCen>), grid, block, shmem, __VA_ARGS__);              \
      else if (src_ == kSrcImg) LAUNCH_ONE((kern<false, kSrcImg>), grid, block, shmem, __VA_ARGS__);              \
      else LAUNCH_ONE((kern<false, kSrcVolume>), grid, block, shmem, __VA_ARGS__);                                \
    }                                                                                                             \
  } while (0)

// the raster sweep reads the paired-cell volumes when the cost object carries them (GRD, fused): same cells, same order, same bits
#define LAUNCH_SWEEP(kern, grid, block, shmem, ...)                                                              \
  do {                                                                                                            \
    if (c->sweep_pairs && c->cost.fused == kSrcGrd) {                                                             \
      if (c->cost.cs) LAUNCH_ONE((kern<true, kSrcVol2>), grid, block, shmem, __VA_ARGS__);                        \
      else LAUNCH_ONE((kern<false, kSrcVol2>), grid, block, shmem, __VA_ARGS__);                                  \
    } else if (c->sweep_packed && c->cost.fused == kSrcGrd) {                                                     \
      if (c->cost.cs) LAUNCH_ONE((kern<true, kSrcGrd8>), grid, block, shmem, __VA_ARGS__);                        \
      else LAUNCH_ONE((kern<false, kSrcGrd8>), grid, block, shmem, __VA_ARGS__);                                  \
    } else {                                                                                                      \
      LAUNCH_CS(kern, grid, block, shmem, __VA_ARGS__);                                                           \
    }                                                                                                             \
  } while (0)

int do_init(cspm_ctx *c, const cspm_pm_params *p) {
  const long long items = 2LL * c->W * c->H;
  Pm pm = make_pm(c, p);
  {
    Timed t(c, CSPM_K_INIT, items);
    const RowQueue rq = next_row_queue(c, 2);
    LAUNCH_CS(k_init, dim3(row_grid(c, 2)), dim3(kRowBlock), row_shmem(c), c->cost, pm, rq, row_cap(c), row_ocap(c));
  }
  HIPCHK(c, hipGetLastError());
  c->field_consistent = true;
  return CSPM_OK;
}

// waves of a sweep workgroup: cross-scale -> one per pyramid level; single-scale -> one per chain pass of a full window
inline bool sweep_folded(const cspm_ctx *c) { return c->cost.cs && kSweepWpl == 1 && c->opt_sweep_fold != 0 && c->cost.levels >= 4; }
inline unsigned sweep_waves(const cspm_ctx *c) {
  if (sweep_folded(c)) return (unsigned)(c->cost.levels - 1);  // the last level is folded onto the waves of levels 1 .. (cspm_chain.h eval_pixel_pair)
  return c->cost.cs ? (unsigned)(c->cost.levels * kSweepWpl) : (unsigned)((c->cost.n + kChainRows - 1) / kChainRows);
}
inline size_t sweep_lds(const cspm_ctx *c) { return sweep_shared_bytes((int)sweep_waves(c) + (sweep_folded(c) ? 1 : 0)); }

int do_spatial(cspm_ctx *c, int iter, const cspm_pm_params *p) {
  Pm pm = make_pm(c, p);
  const int inc = (iter % 2 == 0) ? 1 : -1;
  if (p->schedule == CSPM_SCHED_REDBLACK) {
    const long long items = 2LL * ((c->W + 1) / 2) * c->H;
    for (int r = 0; r < p->rb_rounds; ++r)
      for (int hs = 0; hs < 2; ++hs) {
        Timed t(c, CSPM_K_SPATIAL, items * p->rb_neighbours);
        LAUNCH_CS(k_spatial_rb, dim3(eval_grid(items)), dim3(kEvalBlock), 0, c->cost, pm, (hs + iter) & 1, inc, p->rb_neighbours);
      }
  } else if (!c->opt_raster_launches) {
    // one persistent launch per sweep: 2 workgroups of 8 waves per CU pull pixels in diagonal-major order
    Sweep sw{};
    sw.ctrl = c->d_sweep_ctrl;
    sw.gran[0] = c->d_sweep_gran;
    sw.gran[1] = c->d_sweep_gran + (size_t)c->W * c->H * kGranPerPixel;
    sw.start = c->d_sweep_start;
    sw.nbands = c->sweep_bands_built;
    sw.epoch = ++c->sweep_epoch;
    sw.total = 2u * (unsigned)c->W * (unsigned)c->H;
    sw.timeout_ticks = c->sweep_timeout_ms * 100000LL;  // the constant clock ticks at 100 MHz
    sw.trace = nullptr;
#ifdef CSPM_SWEEP_TRACE
    {
      static long long *d_trace = nullptr;
      if (!d_trace) { (void)hipMalloc((void **)&d_trace, sizeof(long long) * kTraceSlots * (size_t)sw.total); (void)hipMemset(d_trace, 0, sizeof(long long) * kTraceSlots * (size_t)sw.total); }
      sw.trace = d_trace;
      if (const char *path = getenv("CSPM_SWEEP_TRACE_FILE")) {
        static int sweep_no = 0;
        const int dump_after = getenv("CSPM_SWEEP_TRACE_SWEEP") ? atoi(getenv("CSPM_SWEEP_TRACE_SWEEP")) : 1;  // dump sweep n when sweep n+1 is about to start
        if (sweep_no++ == dump_after) {
          std::vector<long long> h((size_t)kTraceSlots * sw.total);
          (void)hipStreamSynchronize(c->stream);
          (void)hipMemcpy(h.data(), d_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
          if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), sizeof(long long), h.size(), fp); fclose(fp); }
        }
      }
    }
#endif
    HIPCHK(c, hipMemsetAsync(c->d_sweep_ctrl + 2, 0, kSweepMaxBands * sizeof(unsigned int), c->stream));  // the claim counters; ctrl[1] is sticky
    const bool flow = c->opt_sweep_flow != 0 && (long long)c->W * c->H < (1LL << 30);
    if (flow) {
      int frc = ensure_flow(c);
      if (frc) return frc;
      sw.ready[0] = c->d_sweep_ready;
      sw.ready[1] = c->d_sweep_ready + (size_t)c->W * c->H;
      sw.queue = c->d_sweep_queue;
      sw.qctl = c->d_sweep_qctl;
      HIPCHK(c, hipMemsetAsync(c->d_sweep_ready, 0, sizeof(unsigned int) * 2 * (size_t)c->W * c->H, c->stream));
      HIPCHK(c, hipMemsetAsync(c->d_sweep_qctl, 0, 4 * sizeof(unsigned int), c->stream));
    }
    int ncu = c->ncu;
    // Workgroups per CU when the caller has not chosen (CSPM_OPT_SWEEP_WG = 0): 2 for a KITTI-size sweep -- bound by its dependency chain,
    // 19.3 / 18.5 ms with 2 / 3, while every resident workgroup holds registers other pairs' kernels want -- and 3 (the most five-wave
    // workgroups of 88 VGPRs a CU holds) once the anti-diagonals are wide enough for THROUGHPUT to bind: 1242 x 600 31.1 -> 27.1 ms,
    // 1600 x 1000 72.0 -> 58.5, 3000 x 2000 274.7 -> 214.8 (tools/exp_wg_threshold.py, profiles/r06_c5/).  Not for folded sweeps: their
    // caller shares the GPU, and with pairs in flight 2 per CU measured best.
    const bool wide = 2LL * std::min(c->W, c->H) >= 4LL * c->ncu && !sweep_folded(c);
    const int wg_per_cu = c->sweep_wg_per_cu > 0 ? c->sweep_wg_per_cu : (wide ? 3 : 2);
    // more workgroups than fit is harmless (unclaimed work is all a late workgroup needs); at least one per band, a multiple of
    // the bands so that every band gets the same number
    unsigned grid = (unsigned)std::max<long long>(sw.nbands, std::min<long long>((long long)sw.total, (long long)ncu * wg_per_cu));
    grid = (grid + (unsigned)sw.nbands - 1) / (unsigned)sw.nbands * (unsigned)sw.nbands;
    const unsigned waves = sweep_waves(c);
    {
      Timed t(c, CSPM_K_SPATIAL, (long long)sw.total * 2);
      if (flow) LAUNCH_SWEEP(k_spatial_flow, dim3(grid), dim3(waves * kWave), sweep_lds(c), c->cost, pm, sw, inc);
      else LAUNCH_SWEEP(k_spatial_sweep, dim3(grid), dim3(waves * kWave), sweep_lds(c), c->cost, pm, sw, inc);
    }
    c->sweep_pending = true;
  } else {
    for (int k = 1; k <= c->W + c->H - 2; ++k) {
      const int ys_lo = std::max(0, k - (c->W - 1)), ys_hi = std::min(c->H - 1, k);
      const long long items = 2LL * (ys_hi - ys_lo + 1);
      Timed t(c, CSPM_K_SPATIAL, items * 2);
      LAUNCH_SWEEP(k_spatial_diag, dim3((unsigned)items), dim3(sweep_waves(c) * kWave), sweep_lds(c), c->cost, pm, k, inc);
    }
  }
  HIPCHK(c, hipGetLastError());
  return CSPM_OK;
}

int do_view(cspm_ctx *c, int iter, const cspm_pm_params *p) {
  Pm pm = make_pm(c, p);
  const long long items = (long long)c->W * c->H;
  const size_t shmem = (size_t)c->W * (sizeof(unsigned long long) + sizeof(unsigned int));
  if (shmem > 160 * 1024) return fail(c, CSPM_ERR_ARG, "image too wide for the view-propagation row resolver");
  ViewCand vc = c->vc;
  const size_t sort_shmem = (size_t)(c->W + 1 + 256) * sizeof(unsigned int);
  if (!c->opt_view_sort || sort_shmem > 160 * 1024) vc.perm = nullptr;
  for (int v = 0; v < 2; ++v) {
    if (vc.perm) {
      Timed t(c, CSPM_K_MISC, 0);
      LAUNCH_ONE(k_view_sort, dim3(c->H), dim3(256), sort_shmem, pm, v, vc);
    }
    {
      Timed t(c, CSPM_K_VIEW, items);
      const RowQueue rq = next_row_queue(c, 1);
      LAUNCH_CS(k_view_eval, dim3(row_grid(c, 1)), dim3(kRowBlock), row_shmem(c), c->cost, pm, rq, v, vc, row_cap(c), row_ocap(c));
    }
    {
      Timed t(c, CSPM_K_MISC, 0);
      LAUNCH_ONE(k_view_resolve, dim3(c->H), dim3(256), shmem, pm, v, iter % 2 == 0 ? 0 : 1, c->vc);
    }
  }
  HIPCHK(c, hipGetLastError());
  return CSPM_OK;
}

int do_refine(cspm_ctx *c, int iter, const cspm_pm_params *p) {
  Pm pm = make_pm(c, p);
  const long long items = 2LL * c->W * c->H;
  const double z_iter = c->max_dis / 2.0, n_iter = 1.0;  // cs_patchmatch.cc:95, cs_patchmatch.h:145
  int steps = 0;
  for (double z = z_iter; z >= 0.1; z /= 2.0) ++steps;   // kZStopThres_, cs_patchmatch.h:146
  // several halving steps per launch: a pixel's steps depend only on its own earlier steps, so the lane keeps its plane in
  // registers between them (c->refine_chunk steps per launch; one launch for all of them by default)
  double z = z_iter, nn = n_iter;
  for (int first = 0; first < steps; first += c->refine_chunk) {
    const int cnt = std::min(c->refine_chunk, steps - first);
    Timed t(c, CSPM_K_REFINE, items * cnt);
    const RowQueue rq = next_row_queue(c, 2);
    LAUNCH_CS(k_refine, dim3(row_grid(c, 2)), dim3(kRowBlock), row_shmem(c), c->cost, pm, rq, iter, first, cnt, z, nn, row_cap(c), row_ocap(c));
    for (int k = 0; k < cnt; ++k) { z /= 2.0; nn /= 2.0; }
  }
  HIPCHK(c, hipGetLastError());
  return CSPM_OK;
}

int run_patchmatch(cspm_ctx *c, int iter_num, const cspm_pm_params *p) {
  int rc;
  if ((rc = do_init(c, p))) return rc;                 // cs_patchmatch.cc:55
  for (int i = 0; i < iter_num; ++i) {                 // :65-102
    if ((rc = do_spatial(c, i, p))) return rc;
    if ((rc = do_view(c, i, p))) return rc;
    if ((rc = do_refine(c, i, p))) return rc;
  }
  return CSPM_OK;
}

// PlaneToDisp + PostProcessing (cs_patchmatch.cc:103-107, 508-588) enqueued on the ctx stream; results in c->d_dis[v]
int postprocess_enqueue(cspm_ctx *c, int dis_scale) {
  Pm pm{};
  pm.W = c->W; pm.H = c->H; pm.f[0] = c->f[0]; pm.f[1] = c->f[1];
  const long long n = (long long)c->W * c->H;
  const Level &L0 = c->cost.lv[0];
  Timed t(c, CSPM_K_POST, 0);
  for (int v = 0; v < 2; ++v)  // PlaneToDisp (cs_patchmatch.cc:103)
    hipLaunchKernelGGL(k_plane_to_disp_u8, dim3(ew_grid(n)), dim3(256), 0, c->stream, pm, v, dis_scale, c->d_dis[v], (size_t)c->W);
  unsigned int *todo_cnt = c->d_todo + 2 * (size_t)n;
  HIPCHK(c, hipMemsetAsync(todo_cnt, 0, 2 * sizeof(unsigned int), c->stream));
  // LeftRightCheck of both views (:516) on the maps as PlaneToDisp left them
  hipLaunchKernelGGL(k_lr_check, dim3(ew_grid(2 * n)), dim3(256), 0, c->stream, c->d_dis[0], c->d_dis[1], c->W, c->H, dis_scale, c->d_valid[0],
                     c->d_valid[1]);
  // FillInvalid (:545): a workgroup per row and view
  if (fill_rows_shmem(c->W) > 160 * 1024) return fail(c, CSPM_ERR_ARG, "image too wide for the row scan of FillInvalid");
  LAUNCH_ONE(k_fill_rows, dim3(2u * (unsigned)c->H), dim3(kFillBlock), fill_rows_shmem(c->W), pm, dis_scale, c->d_valid[0], c->d_valid[1], c->d_dis[0],
             c->d_dis[1], c->d_todo, todo_cnt);
  // WeightedMedian(valid, 35, WMF_GAMMA) (:571-573): a wavefront per listed pixel; exp(-i/10) is the plane-cost LUT
  hipLaunchKernelGGL(k_weighted_median, dim3((unsigned)c->ncu * 8u), dim3(kMedianBlock), 0, c->stream, L0.pix[0], L0.pix[1], L0.Wp, L0.pad, c->W,
                     c->H, c->d_valid[0], c->d_valid[1], c->d_lut, c->d_dis[0], c->d_dis[1], c->d_todo, todo_cnt, 35 / 2);
  HIPCHK(c, hipGetLastError());
  return CSPM_OK;
}

// PlaneToDisp of one view into a device buffer (u8, packed W*H), enqueued on the ctx stream
int enqueue_disp_u8(cspm_ctx *c, int view, int dis_scale, void *d_out) {
  Pm pm{};
  pm.W = c->W; pm.H = c->H; pm.f[0] = c->f[0]; pm.f[1] = c->f[1];
  {
    Timed t(c, CSPM_K_MISC, 0);
    hipLaunchKernelGGL(k_plane_to_disp_u8, dim3(ew_grid((long long)c->W * c->H)), dim3(256), 0, c->stream, pm, view, dis_scale,
                       (uint8_t *)d_out, (size_t)c->W);
  }
  HIPCHK(c, hipGetLastError());
  return CSPM_OK;
}
int enqueue_postprocess_device(cspm_ctx *c, int dis_scale, void *d_l_out, void *d_r_out) {
  int rc = postprocess_enqueue(c, dis_scale);
  if (rc) return rc;
  void *outs[2] = {d_l_out, d_r_out};
  for (int v = 0; v < 2; ++v)
    HIPCHK(c, hipMemcpyAsync(outs[v], c->d_dis[v], (size_t)c->W * c->H, hipMemcpyDeviceToDevice, c->stream));
  return CSPM_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int cspm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int cspm_create(cspm_ctx **out, int device) {
  if (!out) return CSPM_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, CSPM_ERR_HIP, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count=0"));
  if (device < 0 || device >= n) return fail(nullptr, CSPM_ERR_ARG, "device index out of range");
  DevGuard guard_(device);
  if (!guard_.ok) return fail(nullptr, CSPM_ERR_HIP, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail(nullptr, CSPM_ERR_HIP, hipGetErrorString(e));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, CSPM_ERR_HIP, std::string("libcspm_hip is built for gfx950 only, device is ") + prop.gcnArchName);
  cspm_ctx *c = new cspm_ctx();
  c->device = device;
  c->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char *e = getenv("CSPM_REFINE_CHUNK")) c->refine_chunk = std::max(1, atoi(e));
  if (const char *e = getenv("CSPM_SWEEP_WG")) c->sweep_wg_per_cu = std::max(1, atoi(e));
  if (const char *e = getenv("CSPM_ROW_CLAIM")) c->row_claim = atoi(e) < 0 ? -1 : (atoi(e) ? 1 : 0);
  if (const char *e = getenv("CSPM_SWEEP_BANDS")) c->sweep_bands = std::max(1, std::min(kSweepMaxBands, atoi(e)));
  if (const char *e = getenv("CSPM_SWEEP_PAIRS")) c->opt_sweep_pairs = atoi(e) ? 1 : 0;
  if (const char *e = getenv("CSPM_TABLE_VOLUMES")) c->opt_table_volumes = atoi(e) ? 1 : 0;
  if (const char *e = getenv("CSPM_TABLE_VOLUMES_MAX_MB")) c->table_volumes_limit = std::max(0LL, atoll(e)) << 20;
  if (const char *e = getenv("CSPM_SWEEP_PACKED")) c->opt_sweep_packed = atoi(e) ? 1 : 0;
  if (const char *e = getenv("CSPM_SWEEP_FLOW")) c->opt_sweep_flow = atoi(e) ? 1 : 0;
  if (const char *e = getenv("CSPM_VIEW_SORT")) c->opt_view_sort = atoi(e) ? 1 : 0;
  if (const char *e = getenv("CSPM_SWEEP_FOLD")) c->opt_sweep_fold = atoi(e) ? 1 : 0;
  if (const char *e = getenv("CSPM_VOLUMES_MEM_FRACTION")) c->volumes_mem_fraction = std::min(1.0, std::max(0.0, atof(e)));
  if (const char *e = getenv("CSPM_SWEEP_PAIRS_MAX_MB")) c->sweep_pairs_limit = std::max(0LL, atoll(e)) << 20;
  if (const char *e = getenv("CSPM_SWEEP_TIMEOUT_MS")) c->sweep_timeout_ms = std::min(3600000LL, std::max(0LL, atoll(e)));
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    delete c;
    return fail(nullptr, CSPM_ERR_HIP, hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  *out = c;
  return CSPM_OK;
}

void cspm_destroy(cspm_ctx *c) {
  if (!c) return;
  DevGuard guard_(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto &r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  free_cost(c);
  free_field(c);
  free_images(c);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char *cspm_last_error(const cspm_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int cspm_set_stream(cspm_ctx *c, void *s) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return CSPM_OK;
}

int cspm_get_stream(cspm_ctx *c, void **out) {
  if (!c || !out) return CSPM_ERR_ARG;
  *out = (void *)c->stream;
  return CSPM_OK;
}

int cspm_synchronize(cspm_ctx *c) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = check_sweep(c);  // also surfaces a timed-out raster sweep of an asynchronous cspm_patchmatch
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CSPM_OK;
}

// The work is enqueued on the ctx stream.  Device sources: the caller orders its producer before this call on the same
// stream (or synchronises); nothing here waits for the host.  Host sources: rows are gathered with a 2-D copy (a padded
// row is never read past 3*w bytes) into a staging buffer the ctx keeps.
static int set_images_impl(cspm_ctx *c, const void *l, const void *r, int w, int h, size_t stride, bool on_device) {
  if (!c) return CSPM_ERR_ARG;
  if (!l || !r || w < 1 || h < 1 || stride < (size_t)w * 3) return fail(c, CSPM_ERR_ARG, "bad image arguments");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  if (w != c->W || h != c->H) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_cost(c);
    free_field(c);
    free_images(c);
    int rc;
    for (int v = 0; v < 2; ++v)
      if ((rc = dalloc(c, &c->img0[v], (size_t)w * h, nullptr))) return rc;
    c->W = w; c->H = h;
  } else {
    c->cost_ready = false;
  }
  c->field_consistent = false;
  if (c->pm_runs_unchecked) c->phases_unchecked = true;
  const void *src[2] = {l, r};
  const size_t row = (size_t)w * 3;
  if (!on_device && c->stage_bytes < 2 * row * h) {
    if (c->stage) (void)hipFree(c->stage);
    c->stage = nullptr;
    c->stage_bytes = 0;
    HIPCHK(c, hipMalloc((void **)&c->stage, 2 * row * h));
    c->stage_bytes = 2 * row * h;
  }
  for (int v = 0; v < 2; ++v) {
    const uint8_t *d_src = (const uint8_t *)src[v];
    size_t d_stride = stride;
    if (!on_device) {
      uint8_t *dst = c->stage + (size_t)v * row * h;
      HIPCHK(c, hipMemcpy2DAsync(dst, row, src[v], stride, row, h, hipMemcpyHostToDevice, c->stream));
      d_src = dst;
      d_stride = row;
    }
    Timed t(c, CSPM_K_MISC, 0);
    hipLaunchKernelGGL(k_pack_bgr, dim3(ew_grid((long long)w * h)), dim3(256), 0, c->stream, d_src, d_stride, w, h, w, 0, c->img0[v]);
  }
  HIPCHK(c, hipGetLastError());
  if (!on_device) HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller may reuse its host buffers
  return CSPM_OK;
}

int cspm_set_images(cspm_ctx *c, const uint8_t *l, const uint8_t *r, int w, int h, size_t stride) {
  return set_images_impl(c, l, r, w, h, stride, false);
}
int cspm_set_images_device(cspm_ctx *c, const void *l, const void *r, int w, int h, size_t stride) {
  return set_images_impl(c, l, r, w, h, stride, true);
}

int cspm_set_option(cspm_ctx *c, int key, long long value) {
  if (!c) return CSPM_ERR_ARG;
  switch (key) {
    case CSPM_OPT_GRD_VOLUMES: c->opt_grd_volumes = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_RASTER_LAUNCHES: c->opt_raster_launches = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_PAIRS: c->opt_sweep_pairs = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_TABLE_VOLUMES: c->opt_table_volumes = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_PACKED: c->opt_sweep_packed = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_FLOW: c->opt_sweep_flow = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_WG: c->sweep_wg_per_cu = value < 0 ? 0 : (value > 16 ? 16 : (int)value); return CSPM_OK;
    case CSPM_OPT_VIEW_SORT: c->opt_view_sort = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_FOLD: c->opt_sweep_fold = value ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_FAULT_VOLUME_ALLOC: c->fault_volume_alloc = value < 0 ? 0 : (int)value; return CSPM_OK;
    case CSPM_OPT_VOLUME_RETRY_PAIRS: c->volume_retry_pairs = value < 0 ? 0 : value; return CSPM_OK;
    case CSPM_OPT_SWEEP_TIMEOUT_MS:
      if (value < 0 || value > 3600000) return fail(c, CSPM_ERR_ARG, "sweep timeout out of range");
      c->sweep_timeout_ms = value;
      return CSPM_OK;
    default: return fail(c, CSPM_ERR_ARG, "unknown option");
  }
}

int cspm_get_option(cspm_ctx *c, int key, long long *value) {
  if (!c || !value) return CSPM_ERR_ARG;
  switch (key) {
    case CSPM_OPT_GRD_VOLUMES: *value = c->opt_grd_volumes; return CSPM_OK;
    case CSPM_OPT_RASTER_LAUNCHES: *value = c->opt_raster_launches; return CSPM_OK;
    case CSPM_OPT_SWEEP_TIMEOUT_MS: *value = c->sweep_timeout_ms; return CSPM_OK;
    case CSPM_OPT_SWEEP_FALLBACKS: *value = c->sweep_fallbacks; return CSPM_OK;
    case CSPM_OPT_SWEEP_PAIRS: *value = c->opt_sweep_pairs; return CSPM_OK;
    case CSPM_OPT_TABLE_VOLUMES: *value = c->opt_table_volumes; return CSPM_OK;
    case CSPM_OPT_TABLE_VOLUMES_ACTIVE: *value = (c->cost_alloc && c->cost.lv[0].cvol[0]) ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_PAIRS_ACTIVE: *value = c->sweep_pairs ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_VOLUME_FALLBACKS: *value = c->optional_volume_fallbacks; return CSPM_OK;
    case CSPM_OPT_VOLUME_RETRY_PAIRS: *value = c->volume_retry_pairs; return CSPM_OK;
    case CSPM_OPT_VIEW_SORT: *value = c->opt_view_sort; return CSPM_OK;
    case CSPM_OPT_SWEEP_FOLD: *value = c->opt_sweep_fold; return CSPM_OK;
    case CSPM_OPT_SWEEP_PACKED: *value = c->opt_sweep_packed; return CSPM_OK;
    case CSPM_OPT_SWEEP_FLOW: *value = c->opt_sweep_flow; return CSPM_OK;
    case CSPM_OPT_SWEEP_WG: *value = c->sweep_wg_per_cu; return CSPM_OK;
    case CSPM_OPT_SWEEP_PACKED_ACTIVE: *value = c->sweep_packed ? 1 : 0; return CSPM_OK;
    case CSPM_OPT_SWEEP_PACKED_BAD: {  // synchronises: gradients the packer could not represent (always 0 for 8-bit images)
      if (!c->cost_alloc || !c->d_px8_bad) { *value = 0; return CSPM_OK; }
      DevGuard guard_(c->device);
      unsigned int bad = 0;
      HIPCHK(c, hipMemcpyAsync(&bad, c->d_px8_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      *value = bad;
      return CSPM_OK;
    }
    default: return fail(c, CSPM_ERR_ARG, "unknown option");
  }
}

int cspm_build_cost_grd(cspm_ctx *c, int max_dis, int wnd_size, int scale_num, double reg_lambda) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  const bool with_vol = c->opt_grd_volumes != 0;
  int rc = alloc_cost(c, max_dis, wnd_size, scale_num, reg_lambda, with_vol, kKindGrd, !with_vol && c->opt_sweep_pairs != 0,
                      !with_vol && c->opt_table_volumes != 0);
  if (rc) return rc;
  Cost &cd = c->cost;
  // gradients of both views per level (grd_cc.cpp:70-77); then the GRD cells of both views
  // (pre_cs_pc.cc:57-84): stored as volumes when CSPM_OPT_GRD_VOLUMES, otherwise only their max is
  // reduced (pre_cs_pc.cc:75-82) and the PatchMatch kernels recompute cells on the fly.
  for (int s = 0; s < cd.levels; ++s) {
    Level &L = cd.lv[s];
    const long long ppx = (long long)L.Wp * L.H;
    for (int v = 0; v < 2; ++v) {
      double *g = const_cast<double *>(L.grd[v]);
      Timed t(c, CSPM_K_GRD, 0);
      hipLaunchKernelGGL(k_gradient<SrcU32>, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, SrcU32{L.pix[v], L.Wp, L.pad}, L.W, L.H,
                         L.Wp, L.pad, g);
      hipLaunchKernelGGL(k_make_aos, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, L.pix[v], (const double *)g, ppx, (PixG *)L.px[v]);
      // image v is the other view of view 1-v: the left view (0) reads the right image at x-f, x-f-1; the right view the left image at x+f, x+f+1
      hipLaunchKernelGGL(k_make_px16, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, L.pix[v], (const double *)g, L.Wp, L.H, v == 1 ? -1 : 1,
                         (uint4 *)L.px16[v]);
      if (L.px8[v])
        hipLaunchKernelGGL(k_make_px8, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, L.pix[v], (const double *)g, ppx, (Pix8 *)L.px8[v], c->d_px8_bad);
    }
    const long long cells = (long long)L.W * L.H * (L.D + 1);
    for (int v = 0; v < 2; ++v) {
      Timed t(c, CSPM_K_GRD, 0);
      hipLaunchKernelGGL((k_grd_volume<SrcU32, true>), dim3(stride_grid(cells)), dim3(256), 0, c->stream, SrcU32{L.pix[0], L.Wp, L.pad},
                         SrcU32{L.pix[1], L.Wp, L.pad}, L.grd[0], L.grd[1], L.Wp, L.pad, L.W, L.H, 0, L.D + 1, v,
                         (double *)L.vol[v], c->d_maxkeys + v * CSPM_MAX_LEVELS + s, (double2 *)L.vol2[v], (double *)L.cvol[v], L.cvW, L.cvpad);
    }
  }
  HIPCHK(c, hipGetLastError());
  cd.fused = with_vol ? kSrcVolume : kSrcGrd;
  c->sweep_pairs = cd.lv[0].vol2[0] != nullptr;
  c->sweep_packed = !with_vol && cd.lv[0].px8[0] != nullptr;
  c->is_grd = true;
  return finish_cost(c, false);
}

// `new PreSSPC/PreCSPC(l, r, max_dis, wnd, [scale_num,] new CenCC, [reg_lambda])`: census volumes of every level
// built on the device (cc/cen_cc.cc:4-137), then read by the PatchMatch kernels like any CCMethod's volumes.
int cspm_build_cost_cen(cspm_ctx *c, int max_dis, int wnd_size, int scale_num, double reg_lambda) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  const bool with_vol = c->opt_grd_volumes != 0;
  int rc = alloc_cost(c, max_dis, wnd_size, scale_num, reg_lambda, with_vol, kKindCen);
  if (rc) return rc;
  Cost &cd = c->cost;
  for (int s = 0; s < cd.levels; ++s) {
    Level &L = cd.lv[s];
    const long long px = (long long)L.W * L.H, ppx = (long long)L.Wp * L.H;
    for (int v = 0; v < 2; ++v) {
      uint8_t *gray = c->cen_gray[v][s];
      uint32_t *code = const_cast<uint32_t *>(c->cen_code[v][s]);
      Timed t(c, CSPM_K_GRD, 0);
      hipLaunchKernelGGL(k_gray8<SrcU32>, dim3(ew_grid(px)), dim3(256), 0, c->stream, SrcU32{L.pix[v], L.Wp, L.pad}, L.W, L.H, gray);
      hipLaunchKernelGGL(k_census, dim3(ew_grid(px)), dim3(256), 0, c->stream, gray, L.W, L.H, code);
      hipLaunchKernelGGL(k_make_aos_cen, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, L.pix[v], code, L.W, L.H, L.Wp, L.pad, (PixC *)L.pc[v]);
      hipLaunchKernelGGL(k_make_aos, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, L.pix[v], (const double *)nullptr, ppx, (PixG *)L.px[v]);
    }
    for (int v = 0; v < 2; ++v) {  // volumes when asked for, their max (pre_cs_pc.cc:75-82) always
      Timed t(c, CSPM_K_GRD, 0);
      hipLaunchKernelGGL(k_cen_volume, dim3(stride_grid(px * (L.D + 1))), dim3(256), 0, c->stream, c->cen_code[0][s], c->cen_code[1][s], L.W, L.H,
                         0, L.D + 1, v, (double *)L.vol[v], c->d_maxkeys + v * CSPM_MAX_LEVELS + s);
    }
  }
  HIPCHK(c, hipGetLastError());
  cd.fused = with_vol ? kSrcVolume : kSrcCen;
  c->is_cen = true;
  return finish_cost(c, false);
}

// `new GrdPC(l, r, max_dis, wnd)` (scale_num == 0; plane_cost/grd_pc.cc:11-66) / `new CSPC(l, r, max_dis, wnd, scale_num,
// reg_lambda)` (cspc.cc:11-93): pyramid, 8U gray and its x-gradient per level; no volumes, no CCMethod.  max_cost_ has no
// counterpart in these classes: the "impossible disparity" cost is a constant (grd_pc.cc:131-132, cspc.cc:150-152).
int cspm_build_cost_img(cspm_ctx *c, int max_dis, int wnd_size, int scale_num, double reg_lambda) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = alloc_cost(c, max_dis, wnd_size, scale_num, reg_lambda, false, kKindImg);
  if (rc) return rc;
  Cost &cd = c->cost;
  for (int s = 0; s < cd.levels; ++s) {
    Level &L = cd.lv[s];
    const long long px = (long long)L.W * L.H, ppx = (long long)L.Wp * L.H;
    for (int v = 0; v < 2; ++v) {
      uint8_t *gray = c->cen_gray[v][s];
      Timed t(c, CSPM_K_GRD, 0);
      hipLaunchKernelGGL(k_gray8_u8, dim3(ew_grid(px)), dim3(256), 0, c->stream, L.pix[v], L.W, L.H, L.Wp, L.pad, gray);
      hipLaunchKernelGGL(k_make_aos_img, dim3(ew_grid(ppx)), dim3(256), 0, c->stream, L.pix[v], (const uint8_t *)gray, L.W, L.H, L.Wp, L.pad,
                         (PixG *)L.px[v]);
    }
  }
  HIPCHK(c, hipGetLastError());
  cd.fused = kSrcImg;
  c->is_img = true;
  const double alpha = 0.1, tau_clr = 10.0, tau_grd = 2.0;
  return finish_cost(c, false, alpha * tau_clr + (1 - alpha) * tau_grd);
}

// CenCC::buildCV / buildRightCV on host buffers (cc_method.h:31-32, cc/cen_cc.cc:4-137)
int cspm_cen_build_cv_host(int device, const double *l_rgb, const double *r_rgb, int w, int h, int maxDis, int right_view, double *vol_out) {
  if (!l_rgb || !r_rgb || !vol_out || w < 1 || h < 1 || maxDis < 1) return fail(nullptr, CSPM_ERR_ARG, "bad arguments");
  cspm_ctx *c = nullptr;
  int rc = cspm_create(&c, device);
  if (rc) return rc;
  const size_t px = (size_t)w * h;
  double *d[2] = {nullptr, nullptr}, *vol = nullptr;
  uint8_t *gray[2];
  uint32_t *code[2];
  std::vector<void *> tmp;
  auto done = [&](int code_) {
    if (code_) g_create_error = c->err;
    for (void *p : tmp) (void)hipFree(p);
    cspm_destroy(c);
    return code_;
  };
  const double *src[2] = {l_rgb, r_rgb};
  for (int v = 0; v < 2; ++v) {
    if ((rc = dalloc(c, &d[v], px * 3, &tmp)) || (rc = dalloc(c, &gray[v], px, &tmp)) || (rc = dalloc(c, &code[v], px * 3, &tmp))) return done(rc);
    if (hipMemcpyAsync(d[v], src[v], sizeof(double) * px * 3, hipMemcpyHostToDevice, c->stream) != hipSuccess) return done(fail(c, CSPM_ERR_HIP, "upload failed"));
    hipLaunchKernelGGL(k_gray8<SrcF64>, dim3(ew_grid((long long)px)), dim3(256), 0, c->stream, SrcF64{d[v], w}, w, h, gray[v]);
    hipLaunchKernelGGL(k_census, dim3(ew_grid((long long)px)), dim3(256), 0, c->stream, gray[v], w, h, code[v]);
  }
  if ((rc = dalloc(c, &vol, px * maxDis, &tmp))) return done(rc);
  hipLaunchKernelGGL(k_cen_volume, dim3(stride_grid((long long)px * maxDis)), dim3(256), 0, c->stream, code[0], code[1], w, h, 0, maxDis, right_view,
                     vol, (unsigned long long *)nullptr);
  if (hipMemcpyAsync(vol_out, vol, sizeof(double) * px * maxDis, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess)
    return done(fail(c, CSPM_ERR_HIP, "census volume kernel failed"));
  return done(CSPM_OK);
}

int cspm_begin_cost(cspm_ctx *c, int max_dis, int wnd_size, int scale_num, double reg_lambda) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = alloc_cost(c, max_dis, wnd_size, scale_num, reg_lambda, true, kKindForeign);
  if (rc) return rc;
  for (int s = 0; s < c->cost.levels; ++s)
    for (int v = 0; v < 2; ++v) {
      const Level &L = c->cost.lv[s];
      HIPCHK(c, hipMemsetAsync((void *)L.vol[v], 0, sizeof(double) * (size_t)(L.D + 1) * L.W * L.H, c->stream));  // Mat::zeros, pre_cs_pc.cc:52
      hipLaunchKernelGGL(k_make_aos, dim3(ew_grid((long long)L.Wp * L.H)), dim3(256), 0, c->stream, L.pix[v], (const double *)nullptr,
                         (long long)L.Wp * L.H, (PixG *)L.px[v]);
    }
  return CSPM_OK;
}

int cspm_upload_cost_slab(cspm_ctx *c, int view, int level, int d, const double *slab, size_t stride_elems) {
  if (!c) return CSPM_ERR_ARG;
  if (!c->cost_alloc || c->is_grd || c->is_cen || c->is_img || !c->cost.lv[0].vol[0]) return fail(c, CSPM_ERR_STATE, "cspm_begin_cost first");
  if (view < 0 || view > 1 || level < 0 || level >= c->cost.levels || !slab) return fail(c, CSPM_ERR_ARG, "bad view/level/slab");
  const Level &L = c->cost.lv[level];
  if (d < 0 || d > L.D || stride_elems < (size_t)L.W) return fail(c, CSPM_ERR_ARG, "bad slab index or stride");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  double *dst = (double *)L.vol[view] + (size_t)d * L.W * L.H;
  HIPCHK(c, hipMemcpy2DAsync(dst, sizeof(double) * L.W, slab, sizeof(double) * stride_elems, sizeof(double) * L.W, L.H,
                             hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // caller may reuse the slab buffer
  c->cost_ready = false;
  return CSPM_OK;
}

int cspm_finish_cost(cspm_ctx *c) {
  if (!c) return CSPM_ERR_ARG;
  if (!c->cost_alloc || c->is_grd || c->is_cen || c->is_img || !c->cost.lv[0].vol[0]) return fail(c, CSPM_ERR_STATE, "cspm_begin_cost first");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  HIPCHK(c, hipMemsetAsync(c->d_maxkeys, 0, sizeof(unsigned long long) * 2 * CSPM_MAX_LEVELS, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_maxkeys + 2 * CSPM_MAX_LEVELS, 0xFF, sizeof(unsigned long long) * 2 * CSPM_MAX_LEVELS, c->stream));
  for (int s = 0; s < c->cost.levels; ++s)
    for (int v = 0; v < 2; ++v) {
      const Level &L = c->cost.lv[s];
      const long long cells = (long long)(L.D + 1) * L.W * L.H;
      Timed t(c, CSPM_K_GRD, 0);
      hipLaunchKernelGGL(k_volume_max, dim3(2048), dim3(256), 0, c->stream, L.vol[v], cells, c->d_maxkeys + v * CSPM_MAX_LEVELS + s,
                         c->d_maxkeys + 2 * CSPM_MAX_LEVELS + v * CSPM_MAX_LEVELS + s);
    }
  HIPCHK(c, hipGetLastError());
  c->max_cost_fetched = false;
  return finish_cost(c, true);  // a foreign volume may hold negative cells: the early exit is licensed only when its min is >= 0
}

int cspm_get_levels(const cspm_ctx *c) { return (c && c->cost_alloc) ? c->cost.levels : 0; }

int cspm_get_level_dims(const cspm_ctx *c, int level, int *w, int *h, int *max_disp) {
  if (!c || !c->cost_alloc || level < 0 || level >= c->cost.levels) return CSPM_ERR_ARG;
  if (w) *w = c->cost.lv[level].W;
  if (h) *h = c->cost.lv[level].H;
  if (max_disp) *max_disp = c->cost.lv[level].D;
  return CSPM_OK;
}

int cspm_get_level_image(cspm_ctx *c, int view, int level, uint8_t *out) {
  if (!c || !out) return CSPM_ERR_ARG;
  if (!c->cost_alloc || view < 0 || view > 1 || level < 0 || level >= c->cost.levels) return fail(c, CSPM_ERR_ARG, "bad view/level");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  const Level &L = c->cost.lv[level];
  const size_t px = (size_t)L.W * L.H;
  uint8_t *tmp;
  HIPCHK(c, hipMalloc((void **)&tmp, px * 3));
  hipLaunchKernelGGL(k_unpack_bgr, dim3(ew_grid((long long)px)), dim3(256), 0, c->stream, L.pix[view], L.W, L.H, L.Wp, L.pad, tmp);
  HIPCHK(c, hipMemcpyAsync(out, tmp, px * 3, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(tmp);
  return CSPM_OK;
}

int cspm_get_cost_slab(cspm_ctx *c, int view, int level, int d, double *out) {
  if (!c || !out) return CSPM_ERR_ARG;
  if (!c->cost_alloc || view < 0 || view > 1 || level < 0 || level >= c->cost.levels) return fail(c, CSPM_ERR_ARG, "bad view/level");
  const Level &L = c->cost.lv[level];
  if (d < 0 || d > L.D) return fail(c, CSPM_ERR_ARG, "bad slab index");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  const size_t px = (size_t)L.W * L.H;
  if (L.vol[view]) {
    HIPCHK(c, hipMemcpyAsync(out, L.vol[view] + (size_t)d * px, sizeof(double) * px, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CSPM_OK;
  }
  if (c->is_img) return fail(c, CSPM_ERR_STATE, "GrdPC / CSPC have no cost volumes");
  // fused cost: materialise the requested slab with the volume kernel
  double *tmp;
  HIPCHK(c, hipMalloc((void **)&tmp, sizeof(double) * px));
  if (c->is_cen)
    hipLaunchKernelGGL(k_cen_volume, dim3(stride_grid((long long)px)), dim3(256), 0, c->stream, c->cen_code[0][level], c->cen_code[1][level], L.W,
                       L.H, d, 1, view, tmp, (unsigned long long *)nullptr);
  else
  hipLaunchKernelGGL((k_grd_volume<SrcU32, true>), dim3(stride_grid((long long)px)), dim3(256), 0, c->stream, SrcU32{L.pix[0], L.Wp, L.pad},
                     SrcU32{L.pix[1], L.Wp, L.pad}, L.grd[0], L.grd[1], L.Wp, L.pad, L.W, L.H, d, 1, view, tmp,
                     (unsigned long long *)nullptr);
  hipError_t e = hipMemcpyAsync(out, tmp, sizeof(double) * px, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(c, CSPM_ERR_HIP, hipGetErrorString(e));
  return CSPM_OK;
}

int cspm_get_max_cost(cspm_ctx *c, int view, int level, double *out) {
  if (!c || !out) return CSPM_ERR_ARG;
  if (!c->cost_ready || view < 0 || view > 1 || level < 0 || level >= c->cost.levels) return fail(c, CSPM_ERR_STATE, "cost not ready or bad view/level");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = fetch_max_cost(c);
  if (rc) return rc;
  *out = c->host_max_cost[view * CSPM_MAX_LEVELS + level];
  return CSPM_OK;
}

int cspm_get_scale_weights(const cspm_ctx *c, double *out) {
  if (!c || !out || !c->cost_alloc) return CSPM_ERR_ARG;
  for (int s = 0; s < c->cost.levels; ++s) out[s] = c->scale_wgt[s];
  return CSPM_OK;
}

int cspm_grd_build_cv_host(int device, const double *l_rgb, const double *r_rgb, int w, int h, int maxDis, int right_view, double *vol_out) {
  if (!l_rgb || !r_rgb || !vol_out || w < 1 || h < 1 || maxDis < 1) return fail(nullptr, CSPM_ERR_ARG, "bad arguments");
  cspm_ctx *c = nullptr;
  int rc = cspm_create(&c, device);
  if (rc) return rc;
  const size_t px = (size_t)w * h;
  double *dl = nullptr, *dr = nullptr, *gl = nullptr, *gr = nullptr, *vol = nullptr;
  unsigned long long *key = nullptr;
  std::vector<void *> tmp;
  auto done = [&](int code) {
    if (code) g_create_error = c->err;
    for (void *p : tmp) (void)hipFree(p);
    cspm_destroy(c);
    return code;
  };
  if ((rc = dalloc(c, &dl, px * 3, &tmp)) || (rc = dalloc(c, &dr, px * 3, &tmp)) || (rc = dalloc(c, &gl, px, &tmp)) ||
      (rc = dalloc(c, &gr, px, &tmp)) || (rc = dalloc(c, &vol, px * maxDis, &tmp)) || (rc = dalloc(c, &key, 1, &tmp)))
    return done(rc);
  if (hipMemcpyAsync(dl, l_rgb, sizeof(double) * px * 3, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemcpyAsync(dr, r_rgb, sizeof(double) * px * 3, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemsetAsync(key, 0, sizeof *key, c->stream) != hipSuccess)
    return done(fail(c, CSPM_ERR_HIP, "upload failed"));
  hipLaunchKernelGGL(k_gradient<SrcF64>, dim3(ew_grid((long long)px)), dim3(256), 0, c->stream, SrcF64{dl, w}, w, h, w, 0, gl);
  hipLaunchKernelGGL(k_gradient<SrcF64>, dim3(ew_grid((long long)px)), dim3(256), 0, c->stream, SrcF64{dr, w}, w, h, w, 0, gr);
  hipLaunchKernelGGL((k_grd_volume<SrcF64, false>), dim3(stride_grid((long long)px * maxDis)), dim3(256), 0, c->stream, SrcF64{dl, w}, SrcF64{dr, w},
                     gl, gr, w, 0, w, h, 0, maxDis, right_view, vol, key);
  if (hipMemcpyAsync(vol_out, vol, sizeof(double) * px * maxDis, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess)
    return done(fail(c, CSPM_ERR_HIP, "GRD volume kernel failed"));
  return done(CSPM_OK);
}

int cspm_plane_cost_batch(cspm_ctx *c, int view, int n, const int *xy, const double *np, double *out) {
  if (!c) return CSPM_ERR_ARG;
  if (!c->cost_ready) return fail(c, CSPM_ERR_STATE, "no plane cost built");
  if (view < 0 || view > 1 || n < 0 || (n && (!xy || !np || !out))) return fail(c, CSPM_ERR_ARG, "bad arguments");
  if (n == 0) return CSPM_OK;
  for (int i = 0; i < n; ++i)
    if (xy[2 * i] < 0 || xy[2 * i] >= c->W || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= c->H) return fail(c, CSPM_ERR_ARG, "pixel outside the image");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int *dxy = nullptr;
  double *dnp = nullptr, *dout = nullptr;
  std::vector<void *> tmp;
  int rc;
  if ((rc = dalloc(c, &dxy, (size_t)2 * n, &tmp)) || (rc = dalloc(c, &dnp, (size_t)6 * n, &tmp)) || (rc = dalloc(c, &dout, (size_t)n, &tmp))) {
    for (void *p : tmp) (void)hipFree(p);
    return rc;
  }
  hipError_t e = hipMemcpyAsync(dxy, xy, sizeof(int) * 2 * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dnp, np, sizeof(double) * 6 * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    LAUNCH_CS(k_cost_batch, dim3(eval_grid(n)), dim3(kEvalBlock), 0, c->cost, view, n, dxy, dnp, dout);
    e = hipMemcpyAsync(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipGetLastError();
  for (void *p : tmp) (void)hipFree(p);
  if (e != hipSuccess) return fail(c, CSPM_ERR_HIP, hipGetErrorString(e));
  return CSPM_OK;
}

int cspm_pm_default_params(cspm_pm_params *p) {
  if (!p) return CSPM_ERR_ARG;
  *p = kDefaultParams;
  return CSPM_OK;
}

#define PM_ENTER()                                                            \
  if (!c) return CSPM_ERR_ARG;                                                \
  DevGuard guard_(c->device);                                                 \
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");        \
  int rc = check_pm(c, &p);                                                   \
  if (rc) return rc

int cspm_pm_init(cspm_ctx *c, const cspm_pm_params *p) {
  PM_ENTER();
  c->phases_unchecked = true;
  return do_init(c, p);
}
int cspm_pm_spatial(cspm_ctx *c, int iter, const cspm_pm_params *p) {
  PM_ENTER();
  c->phases_unchecked = true;
  if ((rc = do_spatial(c, iter, p))) return rc;
  return check_sweep(c);
}
int cspm_pm_view(cspm_ctx *c, int iter, const cspm_pm_params *p) {
  PM_ENTER();
  c->phases_unchecked = true;
  return do_view(c, iter, p);
}
int cspm_pm_refine(cspm_ctx *c, int iter, const cspm_pm_params *p) {
  PM_ENTER();
  c->phases_unchecked = true;
  return do_refine(c, iter, p);
}

// Asynchronous: everything is enqueued on the ctx stream and the call returns.  A raster sweep that timed out is
// reported by the next synchronising call (cspm_synchronize or any getter).
int cspm_patchmatch(cspm_ctx *c, int iter_num, const cspm_pm_params *p) {
  PM_ENTER();
  if (iter_num < 0 || iter_num > 15) return fail(c, CSPM_ERR_ARG, "iter_num out of range");
  c->last_iters = iter_num;
  c->last_params = *p;
  c->out_reqs.clear();  // outputs requested behind an earlier run: that run can no longer be repeated (two runs unchecked = an error)
  ++c->pm_runs_unchecked;
  return run_patchmatch(c, iter_num, p);
}

int cspm_get_planes(cspm_ctx *c, int view, double *np_out, double *cost_out) {
  if (!c || view < 0 || view > 1) return CSPM_ERR_ARG;
  if (!c->field_alloc) return fail(c, CSPM_ERR_STATE, "no plane field yet");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  {
    int rc = check_sweep(c);
    if (rc) return rc;
  }
  const size_t n = (size_t)c->W * c->H;
  std::vector<double> h(7 * n);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->f[view].nx, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (np_out)
    for (size_t i = 0; i < n; ++i)
      for (int k = 0; k < 6; ++k) np_out[6 * i + k] = h[k * n + i];
  if (cost_out) memcpy(cost_out, h.data() + 6 * n, sizeof(double) * n);
  return CSPM_OK;
}

int cspm_set_planes(cspm_ctx *c, int view, const double *np, const double *cost) {
  if (!c || view < 0 || view > 1 || !np || !cost) return CSPM_ERR_ARG;
  if (!c->img0[0]) return fail(c, CSPM_ERR_STATE, "cspm_set_images first");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = ensure_field(c);
  if (rc) return rc;
  const size_t n = (size_t)c->W * c->H;
  std::vector<double> h(7 * n);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 6; ++k) h[k * n + i] = np[6 * i + k];
  memcpy(h.data() + 6 * n, cost, sizeof(double) * n);
  HIPCHK(c, hipMemcpyAsync(c->f[view].nx, h.data(), sizeof(double) * 7 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->field_consistent = false;  // min_cost is whatever the caller says: the sweep may not assume cost(plane) == min_cost
  c->phases_unchecked = true;
  return CSPM_OK;
}

int cspm_disparity_u8_device(cspm_ctx *c, int view, int dis_scale, void *d_out) {
  if (!c || view < 0 || view > 1 || !d_out) return CSPM_ERR_ARG;
  if (!c->field_alloc) return fail(c, CSPM_ERR_STATE, "no plane field yet");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  // asynchronous: when the run in front of it has an unchecked sweep, remember the request -- a repeated run (sweep timeout)
  // writes the map again from ITS planes, so the caller never reads a map of the aborted run after a successful check
  if (c->sweep_pending) c->remember_output(cspm_ctx::OutReq{0, view, dis_scale, d_out, nullptr});
  return enqueue_disp_u8(c, view, dis_scale, d_out);
}

int cspm_get_disparity_u8(cspm_ctx *c, int view, int dis_scale, uint8_t *out, size_t stride) {
  if (!c || !out || view < 0 || view > 1 || stride < (size_t)c->W) return CSPM_ERR_ARG;
  if (!c->field_alloc) return fail(c, CSPM_ERR_STATE, "no plane field yet");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = check_sweep(c);  // BEFORE PlaneToDisp: a run repeated after a sweep timeout must be the one the map is computed from
  if (rc) return rc;
  if ((rc = enqueue_disp_u8(c, view, dis_scale, c->d_dis[view]))) return rc;
  HIPCHK(c, hipMemcpy2DAsync(out, stride, c->d_dis[view], c->W, c->W, c->H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CSPM_OK;
}

int cspm_get_disparity_f64(cspm_ctx *c, int view, double *out) {
  if (!c || view < 0 || view > 1 || !out) return CSPM_ERR_ARG;
  if (!c->field_alloc) return fail(c, CSPM_ERR_STATE, "no plane field yet");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  {
    int rc = check_sweep(c);
    if (rc) return rc;
  }
  Pm pm{};
  pm.W = c->W; pm.H = c->H; pm.f[0] = c->f[0]; pm.f[1] = c->f[1];
  const size_t n = (size_t)c->W * c->H;
  hipLaunchKernelGGL(k_plane_to_disp_f64, dim3(ew_grid((long long)n)), dim3(256), 0, c->stream, pm, view, c->vc.cost);
  HIPCHK(c, hipMemcpyAsync(out, c->vc.cost, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CSPM_OK;
}

int cspm_postprocess(cspm_ctx *c, int dis_scale, uint8_t *l_out, uint8_t *r_out, size_t stride) {
  if (!c) return CSPM_ERR_ARG;
  if (!c->field_alloc || !c->cost_alloc) return fail(c, CSPM_ERR_STATE, "cspm_postprocess needs a finished PatchMatch");
  if (dis_scale < 1 || stride < (size_t)c->W || !l_out || !r_out) return fail(c, CSPM_ERR_ARG, "bad dis_scale / stride / outputs");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = check_sweep(c);
  if (rc) return rc;
  if ((rc = postprocess_enqueue(c, dis_scale))) return rc;
  uint8_t *outs[2] = {l_out, r_out};
  for (int v = 0; v < 2; ++v)
    HIPCHK(c, hipMemcpy2DAsync(outs[v], stride, c->d_dis[v], c->W, c->W, c->H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return CSPM_OK;
}

// the same with device-resident outputs (W*H bytes each, packed rows): asynchronous like cspm_patchmatch
int cspm_postprocess_device(cspm_ctx *c, int dis_scale, void *d_l_out, void *d_r_out) {
  if (!c) return CSPM_ERR_ARG;
  if (!c->field_alloc || !c->cost_alloc) return fail(c, CSPM_ERR_STATE, "cspm_postprocess_device needs a finished PatchMatch");
  if (dis_scale < 1 || !d_l_out || !d_r_out) return fail(c, CSPM_ERR_ARG, "bad dis_scale / outputs");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  if (c->sweep_pending) c->remember_output(cspm_ctx::OutReq{1, 0, dis_scale, d_l_out, d_r_out});  // see cspm_disparity_u8_device
  return enqueue_postprocess_device(c, dis_scale, d_l_out, d_r_out);
}

int cspm_enable_timing(cspm_ctx *c, int on) {
  if (!c) return CSPM_ERR_ARG;
  c->timing = on != 0;
  return CSPM_OK;
}
int cspm_reset_timing(cspm_ctx *c) {
  if (!c) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = drain_timing(c);
  for (int k = 0; k < CSPM_K_COUNT; ++k) { c->acc_ms[k] = 0; c->acc_launch[k] = 0; c->acc_evals[k] = 0; }
  return rc;
}
int cspm_get_timing(cspm_ctx *c, int k, long long *launches, double *total_ms, long long *evals) {
  if (!c || k < 0 || k >= CSPM_K_COUNT) return CSPM_ERR_ARG;
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  int rc = drain_timing(c);
  if (launches) *launches = c->acc_launch[k];
  if (total_ms) *total_ms = c->acc_ms[k];
  if (evals) *evals = c->acc_evals[k];
  return rc;
}

long long cspm_taps_per_view_pass(const cspm_ctx *c) {
  if (!c || !c->cost_alloc) return 0;
  long long total = 0;
  const Cost &cd = c->cost;
  for (int s = 0; s < cd.levels; ++s) {
    const Level &L = cd.lv[s];
    long long sx = 0, sy = 0;
    for (int x = 0; x < c->W; ++x) {
      const int cx = x >> s;
      sx += std::min(cx + cd.half, L.W - 1) - std::max(cx - cd.half, 0) + 1;
    }
    for (int y = 0; y < c->H; ++y) {
      const int cy = y >> s;
      sy += std::min(cy + cd.half, L.H - 1) - std::max(cy - cd.half, 0) + 1;
    }
    total += sx * sy;
  }
  return total;
}

// lane-taps the row engine EXECUTES for one evaluation of every pixel of one view: every lane of every 64-pixel wave walks
// all window columns of the window rows that lie inside the image (columns outside the image are executed with weight 0,
// the lanes past the end of an image row shadow its last pixel)
long long cspm_row_engine_taps_per_view_pass(const cspm_ctx *c) {
  if (!c || !c->cost_alloc) return 0;
  long long total = 0;
  const Cost &cd = c->cost;
  const long long lanes_per_row = (long long)((c->W + kWave - 1) / kWave) * kWave;
  for (int s = 0; s < cd.levels; ++s) {
    const Level &L = cd.lv[s];
    long long sy = 0;
    for (int y = 0; y < c->H; ++y) {
      const int cy = y >> s;
      sy += std::min(cy + cd.half, L.H - 1) - std::max(cy - cd.half, 0) + 1;
    }
    total += sy * lanes_per_row * cd.n;
  }
  return total;
}

#ifdef CSPM_COUNT_ALIVE
// debug build only: lanes alive after each pyramid level / lanes evaluated at each level, summed over every row-engine launch
int cspm_debug_alive(unsigned long long *out16, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(cspm::g_alive), sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cspm::g_alive), z, sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  return CSPM_OK;
}
int cspm_debug_alive_hist(unsigned long long *out120, int reset) {  // [3 groups][8 levels][5 buckets]
  static unsigned long long z[120];
  if (hipMemcpyFromSymbol(out120, HIP_SYMBOL(cspm::g_alive_hist), sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cspm::g_alive_hist), z, sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  return CSPM_OK;
}
#endif

// ---- CSPatchMatch over a foreign IPlaneCost: see cspm_foreign.h -----------------------------------------------------------
int cspm_fpm_begin(cspm_ctx *c, int w, int h, int max_dis) {
  if (!c) return CSPM_ERR_ARG;
  if (w < 1 || h < 1 || max_dis < 1) return fail(c, CSPM_ERR_ARG, "bad w / h / max_dis");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  if (w != c->W || h != c->H) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_cost(c);
    free_field(c);
    free_images(c);
    c->W = w; c->H = h;
  }
  if (c->cost_alloc && c->max_dis != max_dis) {  // its levels, strips and init range were sized for the old disparity range
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_cost(c);
  }
  c->max_dis = max_dis;
  c->field_consistent = false;
  c->phases_unchecked = true;
  int rc = ensure_field(c);
  if (rc) return rc;
  const long long need = std::max(2LL * w * h, 4LL * std::min(w, h));  // a diagonal batch holds 4 per pixel (tiny images)
  if (c->fpm_cap < need) {
    if ((rc = dalloc(c, &c->fpm.xy, (size_t)need * 2, nullptr)) || (rc = dalloc(c, &c->fpm.view, (size_t)need, nullptr)) ||
        (rc = dalloc(c, &c->fpm.plane, (size_t)need * 6, nullptr)) || (rc = dalloc(c, &c->fpm.cost, (size_t)need, nullptr)))
      return rc;
    c->fpm_cap = need;
  }
  c->fpm_phase = -1;
  return CSPM_OK;
}

int cspm_fpm_candidates(cspm_ctx *c, int phase, int iter, int step, const cspm_pm_params *p, int *n_out, int *xy_out, int *view_out,
                        double *plane_out) {
  if (!c || !n_out || !xy_out || !view_out || !plane_out) return CSPM_ERR_ARG;
  if (!c->fpm_cap) return fail(c, CSPM_ERR_STATE, "cspm_fpm_begin first");
  if (!p) p = &kDefaultParams;
  if (p->schedule != CSPM_SCHED_RASTER) return fail(c, CSPM_ERR_ARG, "a foreign IPlaneCost runs the reference's raster schedule only");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  Pm pm = make_pm(c, p);
  const long long n = (long long)c->W * c->H;
  long long count = 0;
  int inc = 1;
  switch (phase) {
    case CSPM_FPM_INIT:
    case CSPM_FPM_REFINE: {
      if (phase == CSPM_FPM_REFINE && (step < 0 || step > 64)) return fail(c, CSPM_ERR_ARG, "refinement step out of range");
      double z = c->max_dis / 2.0, nn = 1.0;  // cs_patchmatch.cc:95, cs_patchmatch.h:145; halved once per step (:342-343)
      for (int k = 0; k < step && k < 64; ++k) { z /= 2.0; nn /= 2.0; }
      if (phase == CSPM_FPM_REFINE && z < 0.1) return fail(c, CSPM_ERR_ARG, "refinement step out of range");
      count = 2 * n;
      hipLaunchKernelGGL(k_fpm_point_cand, dim3(ew_grid(count)), dim3(256), 0, c->stream, pm, c->fpm, phase == CSPM_FPM_REFINE ? 1 : 0, iter, step, z, nn);
      break;
    }
    case CSPM_FPM_VIEW:
      if (step < 0 || step > 1) return fail(c, CSPM_ERR_ARG, "view propagation: step = target view, 0 or 1");
      count = n;
      hipLaunchKernelGGL(k_fpm_view_cand, dim3(ew_grid(count)), dim3(256), 0, c->stream, pm, c->fpm, step);
      break;
    case CSPM_FPM_SPATIAL: {
      if (step < 0 || step > c->W + c->H - 2) return fail(c, CSPM_ERR_ARG, "spatial propagation: step = anti-diagonal, 0 .. w+h-2");
      inc = (iter % 2 == 0) ? 1 : -1;
      const int cnt = std::min(c->H - 1, step) - std::max(0, step - (c->W - 1)) + 1;
      count = 4LL * cnt;
      hipLaunchKernelGGL(k_fpm_diag_cand, dim3(ew_grid(2LL * cnt)), dim3(256), 0, c->stream, pm, c->fpm, step, inc);
      break;
    }
    default: return fail(c, CSPM_ERR_ARG, "unknown phase");
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(xy_out, c->fpm.xy, sizeof(int) * 2 * count, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(view_out, c->fpm.view, sizeof(int) * count, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(plane_out, c->fpm.plane, sizeof(double) * 6 * count, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->fpm_phase = phase; c->fpm_iter = iter; c->fpm_step = step; c->fpm_inc = inc; c->fpm_count = count; c->fpm_params = *p;
  *n_out = (int)count;
  return CSPM_OK;
}

int cspm_fpm_commit(cspm_ctx *c, const double *cost) {
  if (!c || !cost) return CSPM_ERR_ARG;
  if (c->fpm_phase < 0) return fail(c, CSPM_ERR_STATE, "no candidate batch pending (cspm_fpm_candidates)");
  DevGuard guard_(c->device);
  if (!guard_.ok) return fail(c, CSPM_ERR_HIP, "hipSetDevice failed");
  Pm pm = make_pm(c, &c->fpm_params);
  const long long count = c->fpm_count;
  HIPCHK(c, hipMemcpyAsync(c->fpm.cost, cost, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  switch (c->fpm_phase) {
    case CSPM_FPM_INIT:
    case CSPM_FPM_REFINE:
      hipLaunchKernelGGL(k_fpm_point_commit, dim3(ew_grid(count)), dim3(256), 0, c->stream, pm, c->fpm, c->fpm_phase == CSPM_FPM_REFINE ? 1 : 0);
      break;
    case CSPM_FPM_VIEW: {
      const size_t shmem = (size_t)c->W * (sizeof(unsigned long long) + sizeof(unsigned int));
      if (shmem > 160 * 1024) return fail(c, CSPM_ERR_ARG, "image too wide for the view-propagation row resolver");
      hipLaunchKernelGGL(k_fpm_view_commit, dim3(ew_grid(count)), dim3(256), 0, c->stream, pm, c->fpm, c->vc);
      LAUNCH_ONE(k_view_resolve, dim3(c->H), dim3(256), shmem, pm, c->fpm_step, c->fpm_iter % 2 == 0 ? 0 : 1, c->vc);
      break;
    }
    case CSPM_FPM_SPATIAL:
      hipLaunchKernelGGL(k_fpm_diag_commit, dim3(ew_grid(count / 2)), dim3(256), 0, c->stream, pm, c->fpm, c->fpm_step, c->fpm_inc);
      break;
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // `cost` is the caller's buffer
  c->fpm_phase = -1;
  return CSPM_OK;
}

#ifdef CSPM_ROW_STATS
// debug build only (tools/row_stats.py): the row-engine statistics of cspm_rows.h g_rowstat
int cspm_debug_unionstat(unsigned long long *out64, int reset) {
  static unsigned long long z[64];
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(cspm::g_unionstat), sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cspm::g_unionstat), z, sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  return CSPM_OK;
}
int cspm_debug_rowtime(unsigned long long *out64, int reset) {
  static unsigned long long z[64];
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(cspm::g_rowtime), sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cspm::g_rowtime), z, sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  return CSPM_OK;
}
int cspm_debug_rangestats(unsigned long long *out1024, int reset) {
  static unsigned long long z[16 * 8 * 8];
  if (hipMemcpyFromSymbol(out1024, HIP_SYMBOL(cspm::g_rangestat), sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cspm::g_rangestat), z, sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  return CSPM_OK;
}
int cspm_debug_rowstats(unsigned long long *out1024, int reset) {
  static unsigned long long z[16 * 8 * 8];
  if (hipMemcpyFromSymbol(out1024, HIP_SYMBOL(cspm::g_rowstat), sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cspm::g_rowstat), z, sizeof z) != hipSuccess) return CSPM_ERR_HIP;
  return CSPM_OK;
}
#endif

}  // extern "C"
