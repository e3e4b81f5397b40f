// Test helper (GPU): the reference's volume-free plane costs used the way its (commented-out) main.cc:106-107 would use them:
//   `new GrdPC(l, r, max_dis, wnd)` / `new CSPC(l, r, max_dis, wnd, scale_num, reg_lambda)` -> CSPatchMatch::PatchMatch.
//   img_pc_check <l.ppm> <r.ppm> <max_dis> <use_cs 0|1> <out_l.pgm> <out_r.pgm>
// Prints GetPlaneCost of a fixed plane at three pixels (per-call boundary) and writes the 8-bit maps.
#include "cs_patchmatch.h"
#include "plane_cost/cspc.h"
#include "plane_cost/grd_pc.h"

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  try {
    Mat l = imread(argv[1]), r = imread(argv[2]);
    if (!l.data || !r.data) return 3;
    const int max_dis = std::atoi(argv[3]);
    const bool use_cs = std::atoi(argv[4]) != 0;
    IPlaneCost *pc = use_cs ? static_cast<IPlaneCost *>(new CSPC(l, r, max_dis, 35, 3, 0.3))
                            : static_cast<IPlaneCost *>(new GrdPC(l, r, max_dis, 35));
    const int xs[3] = {0, l.cols / 2, l.cols - 1}, ys[3] = {0, l.rows / 2, l.rows - 1};
    for (int i = 0; i < 3; ++i) {
      Plane p(Vec3d(0.1, -0.2, 0.97), Point3d(xs[i], ys[i], 4.25));
      std::printf("%.17g\n", pc->GetPlaneCost(xs[i], ys[i], p, i == 1 ? kRight : kLeft));
    }
    CSPatchMatch pm(l, r, max_dis, 4);
    pm.set_seed(99);
    pm.PatchMatch(2, pc, true);
    if (!imwrite(argv[5], pm.dis(kLeft)) || !imwrite(argv[6], pm.dis(kRight))) return 4;
    delete pc;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
