// Test helper (GPU): a CCMethod the library knows nothing about, written against the host layer exactly as a
// plugin author would write it against the reference's cc_method.h, driven through PreSSPC / PreCSPC + CSPatchMatch.
//   foreign_cc_check <l.ppm> <r.ppm> <max_dis> <use_cs 0|1> <out_l.pgm> <out_r.pgm>
// Prints GetPlaneCost of a fixed plane at three pixels (per-call boundary) and writes the 8-bit maps.
#include "cs_patchmatch.h"
#include "plane_cost/pre_cs_pc.h"
#include "plane_cost/pre_ss_pc.h"

// cost = |R_l - R_r| + 0.5 * |B_l - B_r| on the CV_64FC3 RGB inputs, 100 where the other view is outside
class AbsDiffCC : public CCMethod {
 public:
  void build(const Mat &l, const Mat &r, int maxDis, Mat *vol, bool right) {
    CV_Assert(l.type() == CV_64FC3 && r.type() == CV_64FC3);
    for (int d = 0; d < maxDis; ++d)
      for (int y = 0; y < l.rows; ++y) {
        const double *pl = l.ptr<double>(y), *pr = r.ptr<double>(y);
        double *c = vol[d].ptr<double>(y);
        for (int x = 0; x < l.cols; ++x) {
          const int xo = right ? x + d : x - d;
          if (xo < 0 || xo >= l.cols) { c[x] = 100.0; continue; }
          const double *a = right ? pr + 3 * x : pl + 3 * x, *b = right ? pl + 3 * xo : pr + 3 * xo;
          c[x] = std::fabs(a[0] - b[0]) + 0.5 * std::fabs(a[2] - b[2]);
        }
      }
  }
  void buildCV(const Mat &l, const Mat &r, const int maxDis, Mat *vol) { build(l, r, maxDis, vol, false); }
  void buildRightCV(const Mat &l, const Mat &r, const int maxDis, Mat *vol) { build(l, r, maxDis, vol, true); }
};

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  try {
    Mat l = imread(argv[1]), r = imread(argv[2]);
    if (!l.data || !r.data) return 3;
    const int max_dis = std::atoi(argv[3]);
    const bool use_cs = std::atoi(argv[4]) != 0;
    AbsDiffCC cc;
    IPlaneCost *pc = use_cs ? static_cast<IPlaneCost *>(new PreCSPC(l, r, max_dis, 35, 3, &cc, 0.3))
                            : static_cast<IPlaneCost *>(new PreSSPC(l, r, max_dis, 35, &cc));
    const int xs[3] = {0, l.cols / 2, l.cols - 1}, ys[3] = {0, l.rows / 2, l.rows - 1};
    for (int i = 0; i < 3; ++i) {
      Plane p(Vec3d(0.1, -0.2, 0.97), Point3d(xs[i], ys[i], 4.25));
      std::printf("%.17g\n", pc->GetPlaneCost(xs[i], ys[i], p, i == 1 ? kRight : kLeft));
    }
    CSPatchMatch pm(l, r, max_dis, 4);
    pm.set_seed(99);
    pm.PatchMatch(2, pc, false);
    if (!imwrite(argv[5], pm.dis(kLeft)) || !imwrite(argv[6], pm.dis(kRight))) return 4;
    delete pc;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
