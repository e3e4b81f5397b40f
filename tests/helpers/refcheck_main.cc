// tests/test_reference_loops.py: the reference's own CSPatchMatch / PreSSPC / PreCSPC / GrdCC (compiled UNMODIFIED from
// /root/reference/CSPM against the test-only stand-ins in tests/helpers/refcheck/) run on a small pair; the plane field, the stored
// costs, the 8-bit maps and sampled GetPlaneCost values are written out for comparison with the oracle in its reference order and
// ROW_SHARED random streams.  Test infrastructure; nothing of the reference is copied into the repository.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <opencv2/opencv.hpp>
// the plane field and the costs are private members of CSPatchMatch: this translation unit (only) reads them
#define private public
#include "cs_patchmatch.h"
#undef private
#include "plane_cost/pre_cs_pc.h"
#include "plane_cost/pre_ss_pc.h"
#include "plane_cost/grd_pc.h"
#include "plane_cost/cspc.h"
#include "cc/grd_cc.h"
#include "cc/cen_cc.h"

int main(int argc, char **argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: refcheck <in.bin> <out.bin>\n"); return 2; }
  std::ifstream in(argv[1], std::ios::binary);
  int hdr[9];  // w, h, max_dis, dis_scale, scale_num, iters, use_pp, wnd, kind (0 GrdCC + PreSSPC/PreCSPC, 1 CenCC + the same, 2 GrdPC / CSPC)
  double lambda;
  unsigned long long seed;
  in.read(reinterpret_cast<char *>(hdr), sizeof hdr);
  in.read(reinterpret_cast<char *>(&lambda), sizeof lambda);
  in.read(reinterpret_cast<char *>(&seed), sizeof seed);
  const int w = hdr[0], h = hdr[1], max_dis = hdr[2], dis_scale = hdr[3], scale_num = hdr[4], iters = hdr[5], use_pp = hdr[6], wnd = hdr[7], kind = hdr[8];
  Mat l(h, w, CV_8UC3), r(h, w, CV_8UC3);
  in.read(reinterpret_cast<char *>(l.data), (std::streamsize)w * h * 3);
  in.read(reinterpret_cast<char *>(r.data), (std::streamsize)w * h * 3);
  int nq = 0;
  in.read(reinterpret_cast<char *>(&nq), sizeof nq);
  std::vector<int> qxyv((size_t)nq * 3);
  std::vector<double> qnp((size_t)nq * 6);  // normal, point z is derived: the query gives normal + (x, y, z)
  in.read(reinterpret_cast<char *>(qxyv.data()), (std::streamsize)qxyv.size() * sizeof(int));
  in.read(reinterpret_cast<char *>(qnp.data()), (std::streamsize)qnp.size() * sizeof(double));
  if (!in) { std::fprintf(stderr, "short input\n"); return 2; }

  std::stringstream quiet;  // the reference prints its progress to cout
  std::streambuf *keep = std::cout.rdbuf(quiet.rdbuf());
  GrdCC grd;
  CenCC cen;
  CCMethod *cc = kind == 1 ? static_cast<CCMethod *>(&cen) : static_cast<CCMethod *>(&grd);
  IPlaneCost *pc;
  if (kind == 2) pc = scale_num > 0 ? static_cast<IPlaneCost *>(new CSPC(l, r, max_dis, wnd, scale_num, lambda)) : static_cast<IPlaneCost *>(new GrdPC(l, r, max_dis, wnd));
  else pc = scale_num > 0 ? static_cast<IPlaneCost *>(new PreCSPC(l, r, max_dis, wnd, scale_num, cc, lambda)) : static_cast<IPlaneCost *>(new PreSSPC(l, r, max_dis, wnd, cc));
  std::vector<double> qcost(nq);
  for (int i = 0; i < nq; ++i) {
    const Plane pl(Vec3d(qnp[6 * i], qnp[6 * i + 1], qnp[6 * i + 2]), Point3d(qnp[6 * i + 3], qnp[6 * i + 4], qnp[6 * i + 5]));
    qcost[i] = pc->GetPlaneCost(qxyv[3 * i], qxyv[3 * i + 1], pl, RefView(qxyv[3 * i + 2]));
  }
  CSPatchMatch pm(l, r, max_dis, dis_scale);
  int steps = 0;
  for (double z = max_dis / 2.0; z >= 0.1; z /= 2.0) ++steps;  // PlaneRefinement(max_dis_ / 2.0, kMaxNorm_, kZStopThres_ = 0.1)
  cv::refcheck::begin(seed, h, steps);
  pm.PatchMatch(iters, pc, use_pp != 0);
  std::cout.rdbuf(keep);

  std::ofstream out(argv[2], std::ios::binary);
  out.write(reinterpret_cast<const char *>(qcost.data()), (std::streamsize)nq * sizeof(double));
  for (int v = 0; v < 2; ++v) {
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const Plane &p = pm.plane_[v][y][x];
        const Vec3d n = p.norm(), prm = p.param();
        const Point3d pt = p.point();
        const double rec[10] = {n[0], n[1], n[2], pt.x, pt.y, pt.z, prm[0], prm[1], prm[2], pm.min_cost_[v][y][x]};
        out.write(reinterpret_cast<const char *>(rec), sizeof rec);
      }
    const Mat &d = pm.dis(RefView(v));
    for (int y = 0; y < h; ++y) out.write(reinterpret_cast<const char *>(d.ptr<uchar>(y)), w);
  }
  delete pc;
  return out ? 0 : 1;
}
