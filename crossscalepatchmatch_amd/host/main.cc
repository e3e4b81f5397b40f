// main.cc -- the command line of the reference (CSPM/main.cc:23-34,57-139): same flags, same defaults, same
// flow (load pair, build the plane cost, PatchMatch, report "Total Time", write the two 8-bit maps), on the GPU.
// Additive flags only: --seed, --schedule, --device, --iters.  A missing image is an error return, not a blocking
// cin.get() (main.cc:70-75).
#include "commfunc.h"
#include "cs_patchmatch.h"
#include "get_method.h"
#include "plane_cost/pre_cs_pc.h"
#include "plane_cost/pre_ss_pc.h"

DEFINE_string(l_img_file, "l_img.png", "input left image file name");
DEFINE_string(r_img_file, "r_img.png", "input right image file name");
DEFINE_string(l_dis_file, "l_dis.png", "output left disparity file name");
DEFINE_string(r_dis_file, "r_dis.png", "output right disparity file name");
DEFINE_int32(max_dis, 0, "max allowed disparity range");
DEFINE_int32(dis_scale, 0, "disparity re-scaling factor");
DEFINE_string(cc_name, "CCName", "cost function name");
DEFINE_bool(use_cs, false, "enable cross-scale cost aggregation");
DEFINE_bool(use_pp, false, "enable post-processing");
DEFINE_double(reg_lambda, 0.0, "regularization lambda");
// additions
DEFINE_int32(seed, 12345, "random seed (the reference seeds from time(NULL))");
DEFINE_string(schedule, "raster", "spatial propagation schedule: raster (reference order) | redblack");
DEFINE_int32(device, 0, "GPU index");
DEFINE_int32(iters, 3, "PatchMatch iterations (main.cc:93 max_iter)");

int main(int argc, char **argv) {
  cout << "PatchMatch Stereo Matching" << endl;
  gflags::ParseCommandLineFlags(&argc, &argv, true);
  cout << "Load Image: " << FLAGS_l_img_file << " " << FLAGS_r_img_file << "\n";
  Mat l_img = imread(FLAGS_l_img_file, CV_LOAD_IMAGE_COLOR);
  Mat r_img = imread(FLAGS_r_img_file, CV_LOAD_IMAGE_COLOR);
  if (!l_img.data || !r_img.data) {
    cout << "Error: can not open image\n";
    return EXIT_FAILURE;
  }
  try {
    DevicePlaneCost::device = FLAGS_device;
    CCMethod *cc_cost = GetCCType(FLAGS_cc_name);  // main.cc:89
    double duration = static_cast<double>(getTickCount());
    const int wnd_size = 35;  // main.cc:94
    IPlaneCost *plane_cost = NULL;
    if (FLAGS_use_cs) {
      const int scale_num = 5;  // main.cc:100
      plane_cost = new PreCSPC(l_img, r_img, FLAGS_max_dis, wnd_size, scale_num, cc_cost, FLAGS_reg_lambda);
    } else {
      plane_cost = new PreSSPC(l_img, r_img, FLAGS_max_dis, wnd_size, cc_cost);
    }
    CSPatchMatch *patch_match = new CSPatchMatch(l_img, r_img, FLAGS_max_dis, FLAGS_dis_scale);
    patch_match->set_seed((uint64_t)FLAGS_seed);
    patch_match->set_schedule(FLAGS_schedule == "redblack" ? 1 : 0);
    patch_match->PatchMatch(FLAGS_iters, plane_cost, FLAGS_use_pp);
    duration = (static_cast<double>(getTickCount()) - duration) / getTickFrequency();
    cout << "--------------------------------------------------------\n";
    cout << "Total Time: " << duration << endl;
    cout << "--------------------------------------------------------\n";
    if (!imwrite(FLAGS_l_dis_file, patch_match->dis(kLeft)) || !imwrite(FLAGS_r_dis_file, patch_match->dis(kRight))) {
      cout << "Error: can not write disparity maps\n";
      return EXIT_FAILURE;
    }
    delete patch_match;
    delete plane_cost;
    delete cc_cost;
  } catch (const std::exception &e) {
    cout << "Error: " << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
