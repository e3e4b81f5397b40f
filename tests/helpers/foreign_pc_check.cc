// Test helper (GPU): CSPatchMatch::PatchMatch driven by a FOREIGN IPlaneCost -- a plugin class this host layer knows nothing
// about (not an IDevicePlaneCost): i_plane_cost.h:28-33.  The plugin prices planes with the CPU oracle in the REFERENCE order
// (so the expected result exists: the oracle's own CSPatchMatch over the same cost), the device draws the candidates and takes
// the accept decisions.  Planes, costs and the post-processed 8-bit maps must equal the oracle's bit for bit.
//   foreign_pc_check [w h max_dis scale_num iters use_pp]      exit 0 = identical
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/cspm_oracle.h"
#include "cs_patchmatch.h"

class OracleBackedCost : public IPlaneCost {  // the "third-party" cost function
 public:
  OracleBackedCost(const Mat &l, const Mat &r, int max_disp, int scale_num, double lambda) : calls_(0) {
    pc_ = csor_pc_create(l.data, r.data, l.cols, l.rows, max_disp, 35, scale_num, lambda);
  }
  ~OracleBackedCost() { csor_pc_destroy(pc_); }
  virtual double GetPlaneCost(const int &ref_x, const int &ref_y, const Plane &plane, const RefView &view) const {
    const Vec3d n = plane.norm(), p = plane.param();
    const double nn[3] = {n[0], n[1], n[2]}, pp[3] = {p[0], p[1], p[2]};
#pragma omp atomic
    ++calls_;
    return csor_pc_cost(pc_, ref_x, ref_y, nn, pp, view == kLeft ? CSOR_LEFT : CSOR_RIGHT, CSOR_SUM_SERIAL);
  }
  csor_pc *pc() const { return pc_; }
  long long calls() const { return calls_; }

 private:
  csor_pc *pc_;
  mutable long long calls_;
};

int main(int argc, char **argv) {
  const int w = argc > 1 ? atoi(argv[1]) : 44, h = argc > 2 ? atoi(argv[2]) : 30, D = argc > 3 ? atoi(argv[3]) : 12;
  const int scale_num = argc > 4 ? atoi(argv[4]) : 3, iters = argc > 5 ? atoi(argv[5]) : 2, use_pp = argc > 6 ? atoi(argv[6]) : 1;
  const int dis_scale = 4;
  const uint64_t seed = 4711;
  // a textured pair with a disparity step: right(x) = left(x + d)
  Mat l(h, w, CV_8UC3), r(h, w, CV_8UC3);
  unsigned s = 12345u;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) {
        s = s * 1664525u + 1013904223u;
        l.ptr<unsigned char>(y)[3 * x + c] = (unsigned char)(96 + ((s >> 24) & 63) + 40 * ((x / 5 + y / 7 + c) & 1));
      }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int d = y < h / 2 ? 3 : 6, xs = x + d < w ? x + d : w - 1;
      for (int c = 0; c < 3; ++c) r.ptr<unsigned char>(y)[3 * x + c] = l.ptr<unsigned char>(y)[3 * xs + c];
    }
  OracleBackedCost plugin(l, r, D, scale_num, 0.3);
  CSPatchMatch matcher(l, r, D, dis_scale);
  matcher.set_seed(seed);
  matcher.PatchMatch(iters, &plugin, use_pp != 0);
  // the expected result: the oracle's CSPatchMatch over the same cost object, reference order, same random streams
  csor_pm *pm = csor_pm_create(l.data, r.data, w, h, D, dis_scale);
  csor_pm_opts o;
  memset(&o, 0, sizeof o);
  o.seed = seed; o.rng_mode = CSOR_RNG_PER_PIXEL; o.schedule = CSOR_SCHED_RASTER; o.sum_order = CSOR_SUM_SERIAL; o.rb_rounds = 1; o.rb_neighbours = 4;
  csor_pm_run(pm, iters, plugin.pc(), use_pp, &o);
  long long bad = 0;
  for (int v = 0; v < 2; ++v) {
    std::vector<Plane> planes;
    std::vector<double> cost;
    matcher.planes(v == 0 ? kLeft : kRight, &planes, &cost);
    const double *P = csor_pm_planes(pm, v), *C = csor_pm_min_cost(pm, v);
    const uint8_t *dis = csor_pm_dis(pm, v);
    for (int i = 0; i < w * h; ++i) {
      const Vec3d n = planes[i].norm(), p = planes[i].param();
      const double got[6] = {n[0], n[1], n[2], p[0], p[1], p[2]}, want[6] = {P[9 * i], P[9 * i + 1], P[9 * i + 2], P[9 * i + 6], P[9 * i + 7], P[9 * i + 8]};
      if (memcmp(got, want, sizeof got) != 0 || memcmp(&cost[i], &C[i], sizeof(double)) != 0) ++bad;
      if (matcher.dis(v == 0 ? kLeft : kRight).ptr<unsigned char>(i / w)[i % w] != dis[i]) ++bad;
    }
  }
  std::printf("foreign IPlaneCost: %dx%d D=%d levels=%d iters=%d pp=%d: %lld GetPlaneCost calls, %lld mismatches\n", w, h, D, scale_num, iters, use_pp,
              plugin.calls(), bad);
  csor_pm_destroy(pm);
  return bad == 0 ? 0 : 1;
}
