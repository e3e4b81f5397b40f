// TEST-ONLY stand-in for <opencv2/opencv.hpp> (tests/test_reference_loops.py).  NOT OpenCV, not part of the product, and not a
// reference build: the image has no OpenCV 2.4, so nothing compiled against this header PINS anything.
// Purpose: compile the reference's UNMODIFIED loops (cs_patchmatch.cc, plane_cost/pre_{ss,cs}_pc.cc, cc/grd_cc.cpp, read in place
// from /root/reference) and run them next to the oracle -- a check that the oracle's TRANSCRIPTION of the reference-owned code (loop
// structure, traversal orders, accept rules, indexing, the order of the random draws) has no slip.  Everything OpenCV would do is
// delegated to the oracle's own restated contracts (pyrDown 8U, RGB2GRAY on 32F, Sobel ksize 1, Mat::inv, Matx::dot, Vec / double),
// and cv::RNG is the counter-based generator of DESIGN.md section 3.1 in its ROW_SHARED mode: the reference (USE_OMP,
// commfunc.h:170) constructs `RNG rng(time(NULL))` once per image row (cs_patchmatch.cc:129-131, 308-310), so the n-th
// construction after refcheck::begin() identifies (phase, iteration, halving step, view) and the n-th uniform() of a row its pixel.
#pragma once
#include <algorithm>
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" {  // oracle/cspm_oracle.h (the checker's contracts)
void csor_pyrdown_bgr8(const uint8_t *src, int w, int h, uint8_t *dst);
void csor_rgb2gray_f32(const double *rgb, int w, int h, float *gray);
void csor_sobel_x_ks1(const float *gray, int w, int h, double *grd);
double csor_rng_u01(uint64_t seed, uint32_t stream, uint64_t pix, uint32_t draw);
uint32_t csor_stream_id(int phase, int iter, int step, int view);
}

namespace cv {

typedef unsigned char uchar;
enum { CV_8U = 0, CV_32F = 5, CV_64F = 6 };
enum { CV_BGR2RGB = 4, CV_BGR2GRAY = 6, CV_RGB2GRAY = 7, CV_BGR2Lab = 44 };
enum { NORM_L2 = 4 };
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(cv::CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(cv::CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(cv::CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(cv::CV_32F, 3)
#define CV_64FC1 CV_MAKETYPE(cv::CV_64F, 1)
#define CV_64FC3 CV_MAKETYPE(cv::CV_64F, 3)
#define CV_Assert(expr)                                                                                        \
  do {                                                                                                         \
    if (!(expr)) throw std::runtime_error(std::string("CV_Assert failed: ") + #expr + " (" + __FILE__ + ")");  \
  } while (0)

class Mat {
 public:
  int rows = 0, cols = 0;
  uchar *data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type;
    step = (size_t)c * elemSize();
    buf_ = std::shared_ptr<uchar>(new uchar[std::max<size_t>(step * r, 1)], std::default_delete<uchar[]>());
    data = buf_.get();
  }
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    std::memset(m.data, 0, m.step * r);
    return m;
  }
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return (type_ >> 3) + 1; }
  size_t elemSize() const { return (size_t)channels() * (depth() == CV_8U ? 1 : depth() == CV_32F ? 4 : 8); }
  Mat clone() const {
    Mat m(rows, cols, type_);
    std::memcpy(m.data, data, step * rows);
    return m;
  }
  template <class T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
  template <class T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
  template <class T> T &at(int y, int x) { return ptr<T>(y)[x]; }
  template <class T> const T &at(int y, int x) const { return ptr<T>(y)[x]; }
  // Mat::convertTo(dst, rtype) without scaling: element-wise casts (8U -> 64F, 64F -> 32F: the two the path uses; pre_cs_pc.cc:62-63,
  // grd_cc.cpp:69-72).  dst may be *this.
  void convertTo(Mat &dst, int rdepth) const {
    const int n = rows * cols * channels();
    Mat out(rows, cols, CV_MAKETYPE(rdepth, channels()));
    for (int i = 0; i < n; ++i) {
      double v;
      if (depth() == CV_8U) v = data[i];
      else if (depth() == CV_32F) v = reinterpret_cast<const float *>(data)[i];
      else v = reinterpret_cast<const double *>(data)[i];
      if (rdepth == CV_64F) reinterpret_cast<double *>(out.data)[i] = v;
      else if (rdepth == CV_32F) reinterpret_cast<float *>(out.data)[i] = (float)v;
      else {  // saturate_cast<uchar>(double): round half to even, clamp (cen_cc.cc:13: the values are the integers of an 8-bit image)
        const double rv = std::nearbyint(v);
        out.data[i] = (uchar)(rv < 0 ? 0 : rv > 255 ? 255 : rv);
      }
    }
    dst = out;
  }
  // Mat::inv() (DECOMP_LU) of a small CV_64FC1 matrix, as the oracle restates cv::invert of OpenCV 2.4 (oracle/cspm_oracle.c
  // csor_scale_weights): closed forms for n <= 3, LU with partial pivoting and reciprocal pivots above
  Mat inv() const {
    const int n = rows;
    if (n != cols || type_ != CV_64FC1) throw std::runtime_error("refcheck stand-in: inv() of a square CV_64FC1 matrix only");
    std::vector<double> A((size_t)n * n), B((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) A[i * n + j] = at<double>(i, j);
    Mat R = Mat::zeros(n, n, CV_64FC1);
    if (n == 1) { R.at<double>(0, 0) = 1. / A[0]; return R; }
    if (n == 2) {
      double d = A[0] * A[3] - A[1] * A[2];
      d = 1. / d;
      R.at<double>(0, 0) = A[3] * d; R.at<double>(0, 1) = -A[1] * d;
      R.at<double>(1, 0) = -A[2] * d; R.at<double>(1, 1) = A[0] * d;
      return R;
    }
    if (n == 3) {
      double d = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
      d = 1. / d;
      R.at<double>(0, 0) = (A[4] * A[8] - A[5] * A[7]) * d;
      R.at<double>(0, 1) = (A[2] * A[7] - A[1] * A[8]) * d;
      R.at<double>(0, 2) = (A[1] * A[5] - A[2] * A[4]) * d;
      R.at<double>(1, 0) = (A[5] * A[6] - A[3] * A[8]) * d;
      R.at<double>(1, 1) = (A[0] * A[8] - A[2] * A[6]) * d;
      R.at<double>(1, 2) = (A[2] * A[3] - A[0] * A[5]) * d;
      R.at<double>(2, 0) = (A[3] * A[7] - A[4] * A[6]) * d;
      R.at<double>(2, 1) = (A[1] * A[6] - A[0] * A[7]) * d;
      R.at<double>(2, 2) = (A[0] * A[4] - A[1] * A[3]) * d;
      return R;
    }
    for (int i = 0; i < n; ++i) B[i * n + i] = 1.0;
    for (int i = 0; i < n; ++i) {
      int k = i;
      for (int j = i + 1; j < n; ++j)
        if (std::fabs(A[j * n + i]) > std::fabs(A[k * n + i])) k = j;
      if (k != i) {
        for (int j = i; j < n; ++j) std::swap(A[i * n + j], A[k * n + j]);
        for (int j = 0; j < n; ++j) std::swap(B[i * n + j], B[k * n + j]);
      }
      const double d = -1 / A[i * n + i];
      for (int j = i + 1; j < n; ++j) {
        const double alpha = A[j * n + i] * d;
        for (int q = i + 1; q < n; ++q) A[j * n + q] += alpha * A[i * n + q];
        for (int q = 0; q < n; ++q) B[j * n + q] += alpha * B[i * n + q];
      }
      A[i * n + i] = -d;
    }
    for (int i = n - 1; i >= 0; --i)
      for (int j = 0; j < n; ++j) {
        double s = B[i * n + j];
        for (int q = i + 1; q < n; ++q) s -= A[i * n + q] * B[q * n + j];
        B[i * n + j] = s * A[i * n + i];
      }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) R.at<double>(i, j) = B[i * n + j];
    return R;
  }

 private:
  int type_ = 0;
  std::shared_ptr<uchar> buf_;
};

struct Vec3d {
  double val[3];
  Vec3d(double a = 0, double b = 0, double c = 0) { val[0] = a; val[1] = b; val[2] = c; }
  double &operator[](int i) { return val[i]; }
  const double &operator[](int i) const { return val[i]; }
  double dot(const Vec3d &o) const {  // cv::Matx::dot: s = 0; s += a[i]*b[i]
    double s = 0;
    for (int i = 0; i < 3; ++i) s += val[i] * o.val[i];
    return s;
  }
};
inline Vec3d operator+(const Vec3d &a, const Vec3d &b) { return Vec3d(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3d operator/(const Vec3d &a, double alpha) {  // cv::Vec / double multiplies by the reciprocal
  const double inv = 1. / alpha;
  return Vec3d(a[0] * inv, a[1] * inv, a[2] * inv);
}
inline double norm(const Vec3d &v, int /*NORM_L2*/) {  // sqrt of the sequential sum of squares
  double s = v[0] * v[0];
  s += v[1] * v[1];
  s += v[2] * v[2];
  return std::sqrt(s);
}
struct Point3d {
  double x, y, z;
  Point3d(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {}
  Point3d(const Vec3d &v) : x(v[0]), y(v[1]), z(v[2]) {}
  operator Vec3d() const { return Vec3d(x, y, z); }
};

template <class T> inline T saturate_cast(int v);
template <> inline uchar saturate_cast<uchar>(int v) { return (uchar)(v < 0 ? 0 : v > 255 ? 255 : v); }

inline int64_t getTickCount() { return 0; }
inline double getTickFrequency() { return 1.0; }
inline void imshow(const std::string &, const Mat &) {}
inline int waitKey(int) { return 0; }

inline void pyrDown(const Mat &src, Mat &dst) {  // 8UC3, default size ((w+1)/2, (h+1)/2), BORDER_REFLECT_101: the oracle's contract
  CV_Assert(src.type() == CV_8UC3);
  Mat out((src.rows + 1) / 2, (src.cols + 1) / 2, CV_8UC3);
  csor_pyrdown_bgr8(src.data, src.cols, src.rows, out.data);
  dst = out;
}
inline void cvtColor(const Mat &src, Mat &dst, int code) {
  if (code == CV_BGR2RGB) {
    CV_Assert(src.type() == CV_8UC3);
    Mat out(src.rows, src.cols, CV_8UC3);
    for (int i = 0; i < src.rows * src.cols; ++i) { out.data[3 * i] = src.data[3 * i + 2]; out.data[3 * i + 1] = src.data[3 * i + 1]; out.data[3 * i + 2] = src.data[3 * i]; }
    dst = out;
  } else if (code == CV_RGB2GRAY && src.depth() == CV_32F) {  // on 32F: r*0.299f + g*0.587f + b*0.114f in float, the oracle's contract
    CV_Assert(src.type() == CV_32FC3);
    const int n = src.rows * src.cols;
    std::vector<double> rgb((size_t)n * 3);
    for (int i = 0; i < 3 * n; ++i) rgb[i] = reinterpret_cast<const float *>(src.data)[i];
    Mat out(src.rows, src.cols, CV_32FC1);
    csor_rgb2gray_f32(rgb.data(), src.cols, src.rows, reinterpret_cast<float *>(out.data));
    dst = out;
  } else if (code == CV_RGB2GRAY || code == CV_BGR2GRAY) {
    // on 8UC3: OpenCV 2.4's fixed-point weights R 4899, G 9617, B 1868, >> 14 with rounding (the contract the oracle restates for
    // CenCC, cen_cc.cc:14, and for GrdPC / CSPC, grd_pc.cc:37)
    CV_Assert(src.type() == CV_8UC3);
    Mat out(src.rows, src.cols, CV_8UC1);
    const int ri = code == CV_RGB2GRAY ? 0 : 2, bi = 2 - ri;
    for (int i = 0; i < src.rows * src.cols; ++i)
      out.data[i] = (uchar)((src.data[3 * i + ri] * 4899 + src.data[3 * i + 1] * 9617 + src.data[3 * i + bi] * 1868 + (1 << 13)) >> 14);
    dst = out;
  } else if (code == CV_BGR2Lab) {
    // GrdPC / CSPC convert to Lab in their constructors (grd_pc.cc:32, cspc.cc:49) and read it only under USE_LAB_WGT, which the
    // reference leaves undefined (grd_pc.h:25): an image of the right shape, never read
    dst = Mat::zeros(src.rows, src.cols, CV_8UC3);
  } else {
    throw std::runtime_error("refcheck stand-in: cvtColor code not used by the path");
  }
}
inline void Sobel(const Mat &src, Mat &dst, int ddepth, int dx, int dy, int ksize) {
  CV_Assert(ddepth == CV_64F && dx == 1 && dy == 0 && ksize == 1);
  Mat out(src.rows, src.cols, CV_64FC1);
  if (src.type() == CV_32FC1) {
    csor_sobel_x_ks1(reinterpret_cast<const float *>(src.data), src.cols, src.rows, reinterpret_cast<double *>(out.data));
  } else {  // 8U gray (grd_pc.cc:40, cspc.cc:57): [-1 0 1], BORDER_REFLECT_101, exact integers
    CV_Assert(src.type() == CV_8UC1);
    const int w = src.cols;
    for (int y = 0; y < src.rows; ++y)
      for (int x = 0; x < w; ++x) {
        const int xm = x - 1 < 0 ? (w > 1 ? 1 : 0) : x - 1, xp = x + 1 >= w ? (w > 1 ? w - 2 : 0) : x + 1;
        out.at<double>(y, x) = (double)((int)src.at<uchar>(y, xp) - (int)src.at<uchar>(y, xm));
      }
  }
  dst = out;
}
inline void minMaxLoc(const Mat &m, double *mn, double *mx) {
  CV_Assert(m.type() == CV_64FC1);
  double lo = m.at<double>(0, 0), hi = lo;
  for (int y = 0; y < m.rows; ++y)
    for (int x = 0; x < m.cols; ++x) { const double v = m.at<double>(y, x); lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  if (mn) *mn = lo;
  if (mx) *mx = hi;
}

namespace refcheck {
struct State {
  uint64_t seed = 0;
  int H = 0, steps = 0;
  long constructions = 0;  // RNG objects constructed since begin()
  bool active = false;
};
inline State &state() { static State s; return s; }
// call right before CSPatchMatch::PatchMatch: H = image rows, steps = halving steps of one PlaneRefinement (max_dis / 2, halved while >= 0.1)
inline void begin(uint64_t seed, int H, int steps) { State &s = state(); s.seed = seed; s.H = H; s.steps = steps; s.constructions = 0; s.active = true; }
}  // namespace refcheck

class RNG {
 public:
  enum { UNIFORM = 0, NORMAL = 1 };
  RNG() { attach(); }
  explicit RNG(uint64_t) { attach(); }
  // cv::RNG::uniform(double a, double b) = u * (b - a) + a; the first draw of a pixel in both phases (cs_patchmatch.cc:134, 322)
  double uniform(double a, double b) {
    ++x_;
    return u01(0) * (b - a) + a;
  }
  // fill(Vec3d, NORMAL, 0, 1) (cs_patchmatch.cc:138): the specified generator has no normal deviates -- the oracle and the device draw the
  // direction by rejection from the unit ball; the reference normalises what it gets (:139-140).
  // fill(Vec3d, UNIFORM, -n, n) (cs_patchmatch.cc:325): draws 1..3 of the pixel.
  void fill(Vec3d &v, int dist, double a, double b) {
    if (dist == UNIFORM) {
      for (int k = 0; k < 3; ++k) v[k] = u01(1 + k) * (b - a) + a;
      return;
    }
    for (int t = 0; t < 32; ++t) {
      for (int k = 0; k < 3; ++k) v[k] = u01(1 + 3 * t + k) * (1.0 - -1.0) + -1.0;
      double s = v[0] * v[0];
      s += v[1] * v[1];
      s += v[2] * v[2];
      if (s <= 1.0 && s > 1e-12) break;
    }
  }

 private:
  void attach() {
    refcheck::State &s = refcheck::state();
    x_ = -1;
    if (!s.active) { stream_ = 0; return; }
    const long c = s.constructions++;
    if (c < 2L * s.H) {  // InitRandomPlane: view 0 rows, then view 1 rows
      stream_ = csor_stream_id(0, 0, 0, (int)(c / s.H));
    } else {             // PlaneRefinement: per iteration and halving step, view 0 rows then view 1 rows
      const long r = c - 2L * s.H, g = r / (2L * s.H);
      stream_ = csor_stream_id(1, (int)(g / s.steps), (int)(g % s.steps), (int)((r / s.H) % 2));
    }
  }
  double u01(uint32_t draw) const { return csor_rng_u01(refcheck::state().seed, stream_, (uint64_t)x_, draw); }
  uint32_t stream_ = 0;
  long x_ = -1;  // column of the pixel being drawn for (ROW_SHARED: the stream is keyed by the column only)
};

}  // namespace cv
