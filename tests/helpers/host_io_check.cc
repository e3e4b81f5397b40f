// Test helper (CPU only): exercises the host layer's image I/O and flag parser without touching the GPU library.
//   host_io_check <in> <out_colour> <out_gray> [flags...]   -> copies the image, writes its G channel, echoes flags
#include "commfunc.h"

DEFINE_string(l_img_file, "l_img.png", "");
DEFINE_int32(max_dis, 0, "");
DEFINE_bool(use_cs, false, "");
DEFINE_bool(use_pp, true, "");
DEFINE_double(reg_lambda, 0.0, "");

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  Mat img = imread(argv[1], CV_LOAD_IMAGE_COLOR);
  if (!img.data) return 3;
  Mat gray(img.rows, img.cols, CV_8UC1);
  for (int y = 0; y < img.rows; ++y)
    for (int x = 0; x < img.cols; ++x) gray.at<unsigned char>(y, x) = img.ptr<unsigned char>(y)[3 * x + 1];
  if (!imwrite(argv[2], img) || !imwrite(argv[3], gray)) return 4;
  int n = argc - 3;
  char **rest = argv + 3;
  gflags::ParseCommandLineFlags(&n, &rest, true);
  std::printf("%s|%d|%d|%d|%.17g\n", FLAGS_l_img_file.c_str(), FLAGS_max_dis, (int)FLAGS_use_cs, (int)FLAGS_use_pp, FLAGS_reg_lambda);
  return 0;
}
