// main.cc -- command line with the reference's ten flags and defaults (CSPM/main.cc:23-34) and its flow
// (main.cc:57-139): read the pair, construct the plane cost (timed), run PatchMatch, print "Total Time", write the
// two 8-bit maps.  Runs on the GPU through the host layer.  Extra flags: --seed --schedule --device --iters.
#include "commfunc.h"
#include "cs_patchmatch.h"
#include "get_method.h"
#include "plane_cost/cspc.h"
#include "plane_cost/grd_pc.h"
#include "plane_cost/pre_cs_pc.h"
#include "plane_cost/pre_ss_pc.h"
#include "pfm_io.h"

#include <fstream>
#include <memory>
#include <sstream>

// images
DEFINE_string(l_img_file, "l_img.png", "left input image (8-bit PNG / PPM / PGM)");
DEFINE_string(r_img_file, "r_img.png", "right input image");
DEFINE_string(l_dis_file, "l_dis.png", "left disparity map to write (8-bit)");
DEFINE_string(r_dis_file, "r_dis.png", "right disparity map to write (8-bit)");
// matching
DEFINE_int32(max_dis, 0, "disparity search range");
DEFINE_int32(dis_scale, 0, "factor applied to disparities before 8-bit quantisation");
DEFINE_string(cc_name, "CCName", "matching cost: GRD | CEN");
DEFINE_string(pc_name, "PRE", "plane cost family: PRE = PreSSPC / PreCSPC over --cc_name's cost volumes (the reference's main.cc); "
                              "IMG = GrdPC / CSPC, the volume-free colour + gradient costs (main.cc:106-107, commented out there)");
DEFINE_bool(use_cs, false, "cross-scale aggregation over a 5-level pyramid (PreCSPC) instead of PreSSPC");
DEFINE_bool(use_pp, false, "left-right check, hole filling and weighted median afterwards");
DEFINE_double(reg_lambda, 0.0, "cross-scale regularisation weight");
// not in the reference
DEFINE_int32(seed, 12345, "random seed (the reference uses the wall clock)");
DEFINE_string(schedule, "raster", "spatial propagation: raster (the reference's sweep) | redblack");
DEFINE_int32(device, 0, "GPU index");
DEFINE_int32(iters, 3, "PatchMatch iterations (3 in the reference, main.cc:93)");
DEFINE_string(l_disp_pfm, "", "also write the left sub-pixel disparity map as float32 PFM: the unquantised plane disparity a*x+b*y+c, "
                              "BEFORE post-processing (--use_pp changes the 8-bit maps only, as in the reference)");
DEFINE_string(r_disp_pfm, "", "also write the right sub-pixel disparity map as float32 PFM (see --l_disp_pfm)");
DEFINE_string(batch_list, "", "text file, one stereo pair per line: l_img r_img l_dis r_dis [l_pfm r_pfm]; all pairs run with the "
                              "matching flags of this command line on one device context (buffers are reused between pairs). A pair "
                              "that fails is reported and the batch goes on; the exit code is non-zero if any pair failed");
DEFINE_bool(batch_skip_existing, false, "with --batch_list: skip the pairs whose output maps already exist (restart an interrupted batch)");
DEFINE_bool(quiet, false, "print errors and the batch summary only");

namespace {
const int kWindow = 35;  // main.cc:94
const int kScales = 5;   // main.cc:100

struct PairFiles {
  string l_img, r_img, l_dis, r_dis, l_pfm, r_pfm;
};

// one stereo pair: the flow of main.cc:57-139
int run_pair(const PairFiles &f, CCMethod *cost_fn) {
  const Mat left = imread(f.l_img, CV_LOAD_IMAGE_COLOR), right = imread(f.r_img, CV_LOAD_IMAGE_COLOR);
  if (left.empty() || right.empty()) {
    // the reference waits for a key press here (main.cc:70-75); a batch tool must not
    cout << "Error: can not open image\n";
    return EXIT_FAILURE;
  }
  const double t0 = static_cast<double>(getTickCount());
  IPlaneCost *plane_cost_raw;
  if (FLAGS_pc_name == "IMG")
    plane_cost_raw = FLAGS_use_cs ? static_cast<IPlaneCost *>(new CSPC(left, right, FLAGS_max_dis, kWindow, kScales, FLAGS_reg_lambda))
                              : static_cast<IPlaneCost *>(new GrdPC(left, right, FLAGS_max_dis, kWindow));
  else
    plane_cost_raw = FLAGS_use_cs ? static_cast<IPlaneCost *>(new PreCSPC(left, right, FLAGS_max_dis, kWindow, kScales, cost_fn, FLAGS_reg_lambda))
                                  : static_cast<IPlaneCost *>(new PreSSPC(left, right, FLAGS_max_dis, kWindow, cost_fn));
  const std::unique_ptr<IPlaneCost> plane_cost_owner(plane_cost_raw);  // released on every path, exceptions included (batch mode goes on)
  IPlaneCost *plane_cost = plane_cost_raw;
  CSPatchMatch matcher(left, right, FLAGS_max_dis, FLAGS_dis_scale);
  matcher.set_seed(static_cast<uint64_t>(FLAGS_seed));
  matcher.set_schedule(FLAGS_schedule == "redblack" ? 1 : 0);
  matcher.PatchMatch(FLAGS_iters, plane_cost, FLAGS_use_pp);
  const double seconds = (static_cast<double>(getTickCount()) - t0) / getTickFrequency();
  if (!FLAGS_quiet)
    cout << "--------------------------------------------------------\n"
         << "Total Time: " << seconds << "\n"
         << "--------------------------------------------------------\n";
  bool written = imwrite(f.l_dis, matcher.dis(kLeft)) && imwrite(f.r_dis, matcher.dis(kRight));
  const string *pfm[2] = {&f.l_pfm, &f.r_pfm};
  for (int v = 0; v < kViewNum && written; ++v) {
    if (pfm[v]->empty()) continue;
    std::vector<double> d;
    matcher.disparity(v == 0 ? kLeft : kRight, &d);
    written = WritePFM(*pfm[v], d.data(), left.cols, left.rows);
  }
  if (!written) {
    cout << "Error: can not write disparity maps\n";
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}

int run() {
  DevicePlaneCost::device = FLAGS_device;
  CCMethod *cost_fn = GetCCType(FLAGS_cc_name);  // NULL for unknown names, rejected by the cost constructors
  int rc = EXIT_SUCCESS;
  if (FLAGS_use_pp && !(FLAGS_l_disp_pfm.empty() && FLAGS_r_disp_pfm.empty()) && !FLAGS_quiet)
    cout << "Note: the PFM maps hold the plane disparities before post-processing\n";
  if (FLAGS_batch_list.empty()) {
    if (!FLAGS_quiet) cout << "Load Image: " << FLAGS_l_img_file << " " << FLAGS_r_img_file << "\n";
    rc = run_pair(PairFiles{FLAGS_l_img_file, FLAGS_r_img_file, FLAGS_l_dis_file, FLAGS_r_dis_file, FLAGS_l_disp_pfm, FLAGS_r_disp_pfm}, cost_fn);
  } else {
    std::ifstream list(FLAGS_batch_list.c_str());
    if (!list) {
      cout << "Error: can not open batch list " << FLAGS_batch_list << "\n";
      delete cost_fn;
      return EXIT_FAILURE;
    }
    DevicePlaneCost::keep_context = true;  // the next pair's PreSSPC / PreCSPC takes over the device buffers of the last
    const double t0 = static_cast<double>(getTickCount());
    string line;
    int pairs = 0, failed = 0, skipped = 0, line_no = 0;
    while (std::getline(list, line)) {
      ++line_no;
      std::istringstream is(line);
      PairFiles f;
      if (!(is >> f.l_img)) continue;  // blank line
      if (f.l_img[0] == '#') continue;
      if (!(is >> f.r_img >> f.l_dis >> f.r_dis)) {
        cout << "Error: batch list line " << line_no << " needs l_img r_img l_dis r_dis: " << line << "\n";
        ++failed;
        continue;
      }
      is >> f.l_pfm >> f.r_pfm;
      if (FLAGS_batch_skip_existing && std::ifstream(f.l_dis.c_str()).good() && std::ifstream(f.r_dis.c_str()).good()) {
        ++skipped;
        continue;
      }
      if (!FLAGS_quiet) cout << "Load Image: " << f.l_img << " " << f.r_img << "\n";
      int pair_rc = EXIT_FAILURE;
      try {
        pair_rc = run_pair(f, cost_fn);
      } catch (const std::exception &e) {  // a bad pair must not take the batch down
        cout << "Error: " << e.what() << "\n";
      }
      ++pairs;
      if (pair_rc != EXIT_SUCCESS) {
        ++failed;
        cout << "Pair FAILED (line " << line_no << "): " << f.l_img << " " << f.r_img << "\n";
      }
    }
    const double seconds = (static_cast<double>(getTickCount()) - t0) / getTickFrequency();
    cout << "Batch: " << pairs << " pairs in " << seconds << " s, " << failed << " failed, " << skipped << " skipped\n";
    if (failed) rc = EXIT_FAILURE;
    DevicePlaneCost::release_kept_context();
  }
  delete cost_fn;
  return rc;
}
}  // namespace

int main(int argc, char **argv) {
  gflags::ParseCommandLineFlags(&argc, &argv, true);
  if (!FLAGS_quiet) cout << "PatchMatch Stereo Matching (MI355X)" << endl;
  try {
    return run();
  } catch (const std::exception &e) {
    cout << "Error: " << e.what() << endl;
    return EXIT_FAILURE;
  }
}
