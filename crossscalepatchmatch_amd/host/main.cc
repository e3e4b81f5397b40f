// main.cc -- command line with the reference's ten flags and defaults (CSPM/main.cc:23-34) and its flow
// (main.cc:57-139): read the pair, construct the plane cost (timed), run PatchMatch, print "Total Time", write the
// two 8-bit maps.  Runs on the GPU through the host layer.  Extra flags: --seed --schedule --device --iters.
#include "commfunc.h"
#include "cs_patchmatch.h"
#include "get_method.h"
#include "plane_cost/pre_cs_pc.h"
#include "plane_cost/pre_ss_pc.h"

// images
DEFINE_string(l_img_file, "l_img.png", "left input image (8-bit PNG / PPM / PGM)");
DEFINE_string(r_img_file, "r_img.png", "right input image");
DEFINE_string(l_dis_file, "l_dis.png", "left disparity map to write (8-bit)");
DEFINE_string(r_dis_file, "r_dis.png", "right disparity map to write (8-bit)");
// matching
DEFINE_int32(max_dis, 0, "disparity search range");
DEFINE_int32(dis_scale, 0, "factor applied to disparities before 8-bit quantisation");
DEFINE_string(cc_name, "CCName", "matching cost: GRD | CEN");
DEFINE_bool(use_cs, false, "cross-scale aggregation over a 5-level pyramid (PreCSPC) instead of PreSSPC");
DEFINE_bool(use_pp, false, "left-right check, hole filling and weighted median afterwards");
DEFINE_double(reg_lambda, 0.0, "cross-scale regularisation weight");
// not in the reference
DEFINE_int32(seed, 12345, "random seed (the reference uses the wall clock)");
DEFINE_string(schedule, "raster", "spatial propagation: raster (the reference's sweep) | redblack");
DEFINE_int32(device, 0, "GPU index");
DEFINE_int32(iters, 3, "PatchMatch iterations (3 in the reference, main.cc:93)");

namespace {
const int kWindow = 35;  // main.cc:94
const int kScales = 5;   // main.cc:100

int run() {
  const Mat left = imread(FLAGS_l_img_file, CV_LOAD_IMAGE_COLOR), right = imread(FLAGS_r_img_file, CV_LOAD_IMAGE_COLOR);
  if (left.empty() || right.empty()) {
    // the reference waits for a key press here (main.cc:70-75); a batch tool must not
    cout << "Error: can not open image\n";
    return EXIT_FAILURE;
  }
  DevicePlaneCost::device = FLAGS_device;
  CCMethod *cost_fn = GetCCType(FLAGS_cc_name);  // NULL for unknown names, rejected by the cost constructors
  const double t0 = static_cast<double>(getTickCount());
  IPlaneCost *plane_cost =
      FLAGS_use_cs ? static_cast<IPlaneCost *>(new PreCSPC(left, right, FLAGS_max_dis, kWindow, kScales, cost_fn, FLAGS_reg_lambda))
                   : static_cast<IPlaneCost *>(new PreSSPC(left, right, FLAGS_max_dis, kWindow, cost_fn));
  CSPatchMatch matcher(left, right, FLAGS_max_dis, FLAGS_dis_scale);
  matcher.set_seed(static_cast<uint64_t>(FLAGS_seed));
  matcher.set_schedule(FLAGS_schedule == "redblack" ? 1 : 0);
  matcher.PatchMatch(FLAGS_iters, plane_cost, FLAGS_use_pp);
  const double seconds = (static_cast<double>(getTickCount()) - t0) / getTickFrequency();
  cout << "--------------------------------------------------------\n"
       << "Total Time: " << seconds << "\n"
       << "--------------------------------------------------------\n";
  const bool written = imwrite(FLAGS_l_dis_file, matcher.dis(kLeft)) && imwrite(FLAGS_r_dis_file, matcher.dis(kRight));
  delete plane_cost;
  delete cost_fn;
  if (!written) {
    cout << "Error: can not write disparity maps\n";
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
}  // namespace

int main(int argc, char **argv) {
  cout << "PatchMatch Stereo Matching (MI355X)" << endl;
  gflags::ParseCommandLineFlags(&argc, &argv, true);
  cout << "Load Image: " << FLAGS_l_img_file << " " << FLAGS_r_img_file << "\n";
  try {
    return run();
  } catch (const std::exception &e) {
    cout << "Error: " << e.what() << endl;
    return EXIT_FAILURE;
  }
}
