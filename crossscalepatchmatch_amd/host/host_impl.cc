// host_impl.cc -- GrdCC, DevicePlaneCost (PreSSPC / PreCSPC) and CSPatchMatch above the C ABI of
// libcspm_hip.so.  No arithmetic of the hot path happens here.
#include <mutex>
#include <vector>

#include "../../include/cspm.h"
#include "cc/cen_cc.h"
#include "cc/grd_cc.h"
#include "cs_patchmatch.h"
#include "plane_cost/device_plane_cost.h"

namespace {
void check(int rc, cspm_ctx *ctx, const char *what) {
  if (rc != CSPM_OK) throw std::runtime_error(std::string(what) + ": " + cspm_last_error(ctx));
}
// CV_64F copy of a Mat with packed rows
std::vector<double> packed64(const Mat &m) {
  CV_Assert(m.depth() == CV_64F);
  const size_t row = (size_t)m.cols * m.channels();
  std::vector<double> v(row * m.rows);
  for (int y = 0; y < m.rows; ++y) std::memcpy(&v[y * row], m.ptr<double>(y), row * sizeof(double));
  return v;
}
}  // namespace

// ---------------------------------------------------------------- GrdCC (cc/grd_cc.cpp:60-154)
void GrdCC::build(const Mat &lImg, const Mat &rImg, int maxDis, Mat *vol, int right) {
  CV_Assert(lImg.type() == CV_64FC3 && rImg.type() == CV_64FC3);  // grd_cc.cpp:63
  CV_Assert(lImg.rows == rImg.rows && lImg.cols == rImg.cols && maxDis >= 1 && vol);
  const int h = lImg.rows, w = lImg.cols;
  std::vector<double> l = packed64(lImg), r = packed64(rImg), out((size_t)maxDis * h * w);
  check(cspm_grd_build_cv_host(device_ >= 0 ? device_ : DeviceSlot::current().device(), l.data(), r.data(), w, h, maxDis, right, out.data()), NULL, "GrdCC");
  for (int d = 0; d < maxDis; ++d) {
    if (vol[d].rows != h || vol[d].cols != w || vol[d].type() != CV_64FC1) vol[d].create(h, w, CV_64FC1);
    for (int y = 0; y < h; ++y) std::memcpy(vol[d].ptr<double>(y), &out[((size_t)d * h + y) * w], sizeof(double) * w);
  }
}
void GrdCC::buildCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *costVol) { build(lImg, rImg, maxDis, costVol, 0); }
void GrdCC::buildRightCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *rCostVol) { build(lImg, rImg, maxDis, rCostVol, 1); }

// ---------------------------------------------------------------- CenCC (cc/cen_cc.cc:4-137)
void CenCC::build(const Mat &lImg, const Mat &rImg, int maxDis, Mat *vol, int right) {
  CV_Assert(lImg.type() == CV_64FC3 && rImg.type() == CV_64FC3);  // cen_cc.cc:7
  CV_Assert(lImg.rows == rImg.rows && lImg.cols == rImg.cols && maxDis >= 1 && vol);
  const int h = lImg.rows, w = lImg.cols;
  std::vector<double> l = packed64(lImg), r = packed64(rImg), out((size_t)maxDis * h * w);
  check(cspm_cen_build_cv_host(device_ >= 0 ? device_ : DeviceSlot::current().device(), l.data(), r.data(), w, h, maxDis, right, out.data()), NULL, "CenCC");
  for (int d = 0; d < maxDis; ++d) {
    if (vol[d].rows != h || vol[d].cols != w || vol[d].type() != CV_64FC1) vol[d].create(h, w, CV_64FC1);
    for (int y = 0; y < h; ++y) std::memcpy(vol[d].ptr<double>(y), &out[((size_t)d * h + y) * w], sizeof(double) * w);
  }
}
void CenCC::buildCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *costVol) { build(lImg, rImg, maxDis, costVol, 0); }
void CenCC::buildRightCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *rCostVol) { build(lImg, rImg, maxDis, rCostVol, 1); }

// ---------------------------------------------------------------- DeviceSlot, PreSSPC / PreCSPC / GrdPC / CSPC
int DevicePlaneCost::device = 0;
bool DevicePlaneCost::keep_context = false;

namespace {
// the only process-wide state of the host layer: the registry of live contexts and the default slot, both behind one mutex
std::mutex g_host_mutex;
std::vector<cspm_ctx *> g_live;
thread_local DeviceSlot *t_slot = NULL;
DeviceSlot &default_slot() {
  static DeviceSlot slot;
  return slot;
}
void forget(std::vector<cspm_ctx *> &v, const cspm_ctx *c) {
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == c) { v.erase(v.begin() + i); return; }
}
long long option(cspm_ctx *ctx, int key) {
  long long v = 0;
  return cspm_get_option(ctx, key, &v) == CSPM_OK ? v : 0;
}
}  // namespace

DeviceSlot::Use::Use(DeviceSlot &slot) : prev_(t_slot) { t_slot = &slot; }
DeviceSlot::Use::~Use() { t_slot = prev_; }

int DeviceSlot::device_count() { return cspm_device_count(); }

DeviceSlot &DeviceSlot::current() {
  if (t_slot) return *t_slot;
  DeviceSlot &d = default_slot();
  cspm_ctx *stale = NULL;
  {
    std::lock_guard<std::mutex> lock(g_host_mutex);  // the default slot follows the statics (the reference-shaped, single-threaded flow)
    if (d.device_ != DevicePlaneCost::device && d.parked_) { stale = d.parked_; d.parked_ = NULL; forget(g_live, stale); }
    d.device_ = DevicePlaneCost::device;
    d.keep_ = DevicePlaneCost::keep_context;
  }
  if (stale) cspm_destroy(stale);
  return d;
}

cspm_ctx *DeviceSlot::take() {
  std::lock_guard<std::mutex> lock(g_host_mutex);
  cspm_ctx *c = parked_;
  parked_ = NULL;
  return c;
}
bool DeviceSlot::park(cspm_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_host_mutex);
  if (!keep_ || parked_) return false;
  parked_ = ctx;
  return true;
}
void DeviceSlot::release() {
  cspm_ctx *c = take();
  if (!c) return;
  DevicePlaneCost::disown(c);
  cspm_destroy(c);
}

void DevicePlaneCost::adopt(cspm_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_host_mutex);
  g_live.push_back(ctx);
}
void DevicePlaneCost::disown(cspm_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_host_mutex);
  forget(g_live, ctx);
}
bool DevicePlaneCost::is_live(const cspm_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_host_mutex);
  for (size_t i = 0; i < g_live.size(); ++i)
    if (g_live[i] == ctx) return true;
  return false;
}
void DevicePlaneCost::release_kept_context() { default_slot().release(); }

void DevicePlaneCost::open_context(const Mat &l_img, const Mat &r_img) {
  CV_Assert(l_img.type() == CV_8UC3 && r_img.type() == CV_8UC3);  // pre_cs_pc.cc:25, pre_ss_pc.cc:24, grd_pc.cc:22, cspc.cc:24
  CV_Assert(l_img.rows == r_img.rows && l_img.cols == r_img.cols);
  if (!slot_) slot_ = &DeviceSlot::current();
  ctx_ = slot_->take();
  if (!ctx_) {
    check(cspm_create(&ctx_, slot_->device()), NULL, "cspm_create");
    adopt(ctx_);
    if (slot_->shared_gpu() && !std::getenv("CSPM_SWEEP_FOLD"))  // the environment variable, if set, has decided at cspm_create
      check(cspm_set_option(ctx_, CSPM_OPT_SWEEP_FOLD, 1), ctx_, "cspm_set_option");
  }
  base_sweep_fallbacks_ = option(ctx_, CSPM_OPT_SWEEP_FALLBACKS);
  base_volume_fallbacks_ = option(ctx_, CSPM_OPT_VOLUME_FALLBACKS);
  const Mat l = l_img.clone(), r = r_img.clone();  // packed rows
  check(cspm_set_images(ctx_, l.data, r.data, l.cols, l.rows, l.step), ctx_, "cspm_set_images");
}

// a constructor that throws never runs the destructor: hand the context back (or destroy it) before the exception leaves
#define CSPM_CTOR_GUARD(body)        \
  try {                              \
    body                             \
  } catch (...) {                    \
    close_context();                 \
    throw;                           \
  }

void DevicePlaneCost::close_context() {
  if (!ctx_) return;
  if (slot_) {
    const long long ds = option(ctx_, CSPM_OPT_SWEEP_FALLBACKS) - base_sweep_fallbacks_, dv = option(ctx_, CSPM_OPT_VOLUME_FALLBACKS) - base_volume_fallbacks_;
    std::lock_guard<std::mutex> lock(g_host_mutex);  // the default slot is shared by every thread that names none
    slot_->sweep_fallbacks_ += ds;
    slot_->volume_fallbacks_ += dv;
  }
  if (!(slot_ && slot_->park(ctx_))) {
    disown(ctx_);
    cspm_destroy(ctx_);
  }
  ctx_ = NULL;
}

// GrdPC / CSPC
DevicePlaneCost::DevicePlaneCost(const Mat &l_img, const Mat &r_img, int max_disp, int wnd_size, int scale_num, double reg_lambda, DeviceSlot *slot)
    : ctx_(NULL), slot_(slot), base_sweep_fallbacks_(0), base_volume_fallbacks_(0) {
  CSPM_CTOR_GUARD(
    open_context(l_img, r_img);
    check(cspm_build_cost_img(ctx_, max_disp, wnd_size, scale_num, reg_lambda), ctx_, "cspm_build_cost_img");
  )
}

DevicePlaneCost::DevicePlaneCost(const Mat &l_img, const Mat &r_img, int max_disp, int wnd_size, int scale_num,
                                 CCMethod *cc_method, double reg_lambda, DeviceSlot *slot)
    : ctx_(NULL), slot_(slot), base_sweep_fallbacks_(0), base_volume_fallbacks_(0) {
  if (!cc_method) throw std::runtime_error("PreSSPC/PreCSPC: NULL CCMethod (unknown --cc_name)");  // the reference dereferences it
  CSPM_CTOR_GUARD(
    open_context(l_img, r_img);
    if (dynamic_cast<GrdCC *>(cc_method)) {
      // the known cost function: pyramid, gradients, max_cost, scale weights all on the device
      check(cspm_build_cost_grd(ctx_, max_disp, wnd_size, scale_num, reg_lambda), ctx_, "cspm_build_cost_grd");
    } else if (dynamic_cast<CenCC *>(cc_method)) {
      check(cspm_build_cost_cen(ctx_, max_disp, wnd_size, scale_num, reg_lambda), ctx_, "cspm_build_cost_cen");
    } else {
      // a foreign CCMethod: let it fill host volumes level by level exactly as pre_cs_pc.cc:57-74 does
      check(cspm_begin_cost(ctx_, max_disp, wnd_size, scale_num, reg_lambda), ctx_, "cspm_begin_cost");
      const int levels = cspm_get_levels(ctx_);
      for (int s = 0; s < levels; ++s)
        for (int v = 0; v < kViewNum; ++v) upload_foreign(cc_method, v, s);
      check(cspm_finish_cost(ctx_), ctx_, "cspm_finish_cost");
    }
  )
}

void DevicePlaneCost::upload_foreign(CCMethod *cc, int view, int level) {
  int w, h, D;
  check(cspm_get_level_dims(ctx_, level, &w, &h, &D), ctx_, "cspm_get_level_dims");
  Mat rgb[2];
  for (int v = 0; v < 2; ++v) {  // cvtColor(BGR2RGB) + convertTo(CV_64F), pre_cs_pc.cc:60-64
    std::vector<unsigned char> bgr((size_t)w * h * 3);
    check(cspm_get_level_image(ctx_, v, level, bgr.data()), ctx_, "cspm_get_level_image");
    rgb[v].create(h, w, CV_64FC3);
    for (int y = 0; y < h; ++y) {
      double *o = rgb[v].ptr<double>(y);
      for (int x = 0; x < w; ++x)
        for (int c = 0; c < 3; ++c) o[3 * x + c] = bgr[((size_t)y * w + x) * 3 + (2 - c)];
    }
  }
  std::vector<Mat> vol(D + 1);
  for (int d = 0; d <= D; ++d) vol[d] = Mat::zeros(h, w, CV_64FC1);  // pre_cs_pc.cc:50-53
  if (view == kLeft) cc->buildCV(rgb[0], rgb[1], D + 1, vol.data());
  else cc->buildRightCV(rgb[0], rgb[1], D + 1, vol.data());
  for (int d = 0; d <= D; ++d)
    check(cspm_upload_cost_slab(ctx_, view, level, d, vol[d].ptr<double>(0), vol[d].step / sizeof(double)), ctx_, "cspm_upload_cost_slab");
}

DevicePlaneCost::~DevicePlaneCost() { close_context(); }

double DevicePlaneCost::GetPlaneCost(const int &ref_x, const int &ref_y, const Plane &plane, const RefView &view) const {
  const int xy[2] = {ref_x, ref_y};
  const Vec3d n = plane.norm(), p = plane.param();
  const double np[6] = {n[0], n[1], n[2], p[0], p[1], p[2]};
  double cost = 0.0;
  check(cspm_plane_cost_batch(ctx_, view, 1, xy, np, &cost), ctx_, "cspm_plane_cost_batch");
  return cost;
}

// ---------------------------------------------------------------- CSPatchMatch (cs_patchmatch.cc:3-109)
CSPatchMatch::CSPatchMatch(const Mat &l_img, const Mat &r_img, const int &max_dis, const int &dis_scale)
    : max_dis_(max_dis), dis_scale_(dis_scale), seed_(12345), schedule_(CSPM_SCHED_RASTER), rb_rounds_(1), last_ctx_(NULL), own_ctx_(NULL), pending_ctx_(NULL), pending_pp_(false) {
  CV_Assert(l_img.type() == CV_8UC3 && r_img.type() == CV_8UC3);  // cs_patchmatch.cc:8
  img_[kLeft] = l_img.clone();
  img_[kRight] = r_img.clone();
  wid_ = l_img.cols;
  hei_ = l_img.rows;
  for (int v = 0; v < kViewNum; ++v) dis_[v] = Mat::zeros(hei_, wid_, CV_8UC1);
}

// A foreign IPlaneCost (i_plane_cost.h:28-33): the device owns the plane field, the random streams and the accept rules
// (cspm_fpm_*, csrc/cspm_foreign.h); every candidate is priced by the plugin's GetPlaneCost, exactly the calls the reference
// makes (cs_patchmatch.cc:144,181,191,200,208,269,334), batch by batch.  GetPlaneCost is const and re-entrant
// (the reference calls it from OpenMP threads): the batches are evaluated in parallel when this file is built with -fopenmp.
void CSPatchMatch::PatchMatchForeign(int iter_num, const IPlaneCost *plane_cost, bool use_pp) {
  if (!own_ctx_) {
    check(cspm_create(&own_ctx_, DeviceSlot::current().device()), NULL, "cspm_create");  // the calling thread's GPU
    DevicePlaneCost::adopt(own_ctx_);
  }
  cspm_ctx *ctx = own_ctx_;
  const Mat l = img_[kLeft].clone(), r = img_[kRight].clone();
  check(cspm_set_images(ctx, l.data, r.data, l.cols, l.rows, l.step), ctx, "cspm_set_images");
  check(cspm_fpm_begin(ctx, wid_, hei_, max_dis_), ctx, "cspm_fpm_begin");
  cspm_pm_params p;
  cspm_pm_default_params(&p);
  p.seed = seed_;
  if (schedule_ != CSPM_SCHED_RASTER) throw std::runtime_error("CSPatchMatch: a foreign IPlaneCost runs the raster schedule only");
  const size_t cap = (size_t)2 * wid_ * hei_;
  std::vector<int> xy(2 * cap), view(cap);
  std::vector<double> plane(6 * cap), cost(cap);
  auto batch = [&](int phase, int iter, int step) {
    int n = 0;
    check(cspm_fpm_candidates(ctx, phase, iter, step, &p, &n, xy.data(), view.data(), plane.data()), ctx, "cspm_fpm_candidates");
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int i = 0; i < n; ++i) {
      cost[i] = 0.0;
      if (xy[2 * i] < 0) continue;  // no such candidate (first sweep row / column, proposal outside the image)
      const double *q = &plane[6 * (size_t)i];
      Plane pl;
      pl.set_norm(Point3d(q[0], q[1], q[2]));
      pl.set_param(Vec3d(q[3], q[4], q[5]));
      cost[i] = plane_cost->GetPlaneCost(xy[2 * i], xy[2 * i + 1], pl, view[i] == 0 ? kLeft : kRight);
    }
    check(cspm_fpm_commit(ctx, cost.data()), ctx, "cspm_fpm_commit");
  };
  int steps = 0;
  for (double z = max_dis_ / 2.0; z >= 0.1; z /= 2.0) ++steps;  // kZStopThres_ (cs_patchmatch.h:146), :299-301,342
  batch(CSPM_FPM_INIT, 0, 0);                                   // cs_patchmatch.cc:55
  for (int it = 0; it < iter_num; ++it) {                       // :65-102
    for (int k = 1; k <= wid_ + hei_ - 2; ++k) batch(CSPM_FPM_SPATIAL, it, k);
    for (int v = 0; v < kViewNum; ++v) batch(CSPM_FPM_VIEW, it, v);
    for (int st = 0; st < steps; ++st) batch(CSPM_FPM_REFINE, it, st);
  }
  if (use_pp) {  // PostProcessing reads the level-0 images of a cost object: the cheapest one provides them
    check(cspm_build_cost_img(ctx, max_dis_, 35, 0, 0.0), ctx, "cspm_build_cost_img");
    check(cspm_postprocess(ctx, dis_scale_, dis_[kLeft].data, dis_[kRight].data, dis_[kLeft].step), ctx, "cspm_postprocess");
  } else {
    for (int v = 0; v < kViewNum; ++v)
      check(cspm_get_disparity_u8(ctx, v, dis_scale_, dis_[v].data, dis_[v].step), ctx, "cspm_get_disparity_u8");
  }
  last_ctx_ = ctx;
}

CSPatchMatch::~CSPatchMatch() {
  if (own_ctx_) {
    DevicePlaneCost::disown(own_ctx_);
    cspm_destroy(own_ctx_);
  }
}

void CSPatchMatch::PatchMatchBegin(const int &iter_num, const IPlaneCost *plane_cost, const bool &use_pp) {
  if (pending_ctx_) throw std::runtime_error("CSPatchMatch::PatchMatchBegin: the previous run has not been ended");
  const IDevicePlaneCost *dev = dynamic_cast<const IDevicePlaneCost *>(plane_cost);
  if (!dev) {
    PatchMatchForeign(iter_num, plane_cost, use_pp);
    return;
  }
  cspm_ctx *ctx = dev->device_ctx();
  cspm_pm_params p;
  cspm_pm_default_params(&p);
  p.seed = seed_;
  p.schedule = schedule_;
  p.rb_rounds = rb_rounds_;
  check(cspm_patchmatch(ctx, iter_num, &p), ctx, "cspm_patchmatch");  // asynchronous: enqueued on the context's stream
  pending_ctx_ = ctx;
  pending_pp_ = use_pp;
}

void CSPatchMatch::PatchMatchEnd() {
  cspm_ctx *ctx = pending_ctx_;
  if (!ctx) return;  // nothing pending (a foreign IPlaneCost finished inside Begin)
  pending_ctx_ = NULL;
  if (!DevicePlaneCost::is_live(ctx)) throw std::runtime_error("CSPatchMatch::PatchMatchEnd: the plane cost the run was started on has been deleted");
  if (pending_pp_) {  // PostProcessing (cs_patchmatch.cc:105-107)
    check(cspm_postprocess(ctx, dis_scale_, dis_[kLeft].data, dis_[kRight].data, dis_[kLeft].step), ctx, "cspm_postprocess");
  } else {            // PlaneToDisp (cs_patchmatch.cc:103)
    for (int v = 0; v < kViewNum; ++v)
      check(cspm_get_disparity_u8(ctx, v, dis_scale_, dis_[v].data, dis_[v].step), ctx, "cspm_get_disparity_u8");
  }
  last_ctx_ = ctx;
}

void CSPatchMatch::PatchMatch(const int &iter_num, const IPlaneCost *plane_cost, const bool &use_pp) {
  PatchMatchBegin(iter_num, plane_cost, use_pp);
  PatchMatchEnd();
}

void CSPatchMatch::disparity(const RefView &view, std::vector<double> *out) const {
  if (!last_ctx_) throw std::runtime_error("CSPatchMatch::disparity before PatchMatch");
  if (!DevicePlaneCost::is_live(last_ctx_)) throw std::runtime_error("CSPatchMatch::disparity: the plane cost PatchMatch ran on has been deleted");
  out->resize((size_t)wid_ * hei_);
  check(cspm_get_disparity_f64(last_ctx_, view, out->data()), last_ctx_, "cspm_get_disparity_f64");
}

void CSPatchMatch::planes(const RefView &view, std::vector<Plane> *out, std::vector<double> *min_cost) const {
  if (!last_ctx_) throw std::runtime_error("CSPatchMatch::planes before PatchMatch");
  if (!DevicePlaneCost::is_live(last_ctx_)) throw std::runtime_error("CSPatchMatch::planes: the plane cost PatchMatch ran on has been deleted");
  const size_t n = (size_t)wid_ * hei_;
  std::vector<double> np(6 * n), cost(n);
  check(cspm_get_planes(last_ctx_, view, np.data(), cost.data()), last_ctx_, "cspm_get_planes");
  if (out) {
    out->resize(n);
    for (size_t i = 0; i < n; ++i) {
      Plane pl;
      pl.set_norm(Point3d(np[6 * i], np[6 * i + 1], np[6 * i + 2]));
      pl.set_param(Vec3d(np[6 * i + 3], np[6 * i + 4], np[6 * i + 5]));
      (*out)[i] = pl;
    }
  }
  if (min_cost) *min_cost = cost;
}
