// cv_compat.h -- the small part of the OpenCV 2.4 API the hot path's host side touches,
// so that the plugin surface keeps the reference's signatures (const Mat&, Mat*, Vec3d, Point3d).
// OpenCV is not installed in this image; when <opencv2/core/core.hpp> exists, build with -DCSPM_USE_OPENCV
// and the real types are used instead.  Nothing here computes on the hot path: all arithmetic is in
// libcspm_hip.so.
#pragma once
#ifdef CSPM_USE_OPENCV
#include <opencv2/opencv.hpp>
#else
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace cv {

typedef unsigned char uchar;
enum { CV_8U = 0, CV_32F = 5, CV_64F = 6 };
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(cv::CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(cv::CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(cv::CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(cv::CV_64F, 1)
#define CV_64FC3 CV_MAKETYPE(cv::CV_64F, 3)
#define CV_LOAD_IMAGE_COLOR 1
#define CV_Assert(expr)                                                                                          \
  do {                                                                                                           \
    if (!(expr)) throw std::runtime_error(std::string("CV_Assert failed: ") + #expr + " (" + __FILE__ + ")");    \
  } while (0)

struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};

class Mat {
 public:
  int rows = 0, cols = 0;
  uchar *data = nullptr;
  size_t step = 0;  // bytes per row
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
  Size size() const { return Size(cols, rows); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int y = 0; y < rows; ++y) std::memcpy(m.data + y * m.step, data + y * step, (size_t)cols * elemSize());
    return m;
  }
  template <class T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
  template <class T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
  template <class T> T &at(int y, int x) { return ptr<T>(y)[x]; }
  template <class T> const T &at(int y, int x) const { return ptr<T>(y)[x]; }

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
struct Point3d {
  double x, y, z;
  Point3d(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {}
  Point3d(const Vec3d &v) : x(v[0]), y(v[1]), z(v[2]) {}
  operator Vec3d() const { return Vec3d(x, y, z); }
};

// ---- image files: 8-bit PNG (via zlib) and binary PNM.  main.cc:68-69,133-134 ----
Mat imread(const std::string &path, int flags = CV_LOAD_IMAGE_COLOR);  // 8UC3 BGR; empty Mat on failure
bool imwrite(const std::string &path, const Mat &img);                 // 8UC1 or 8UC3 (BGR)

inline int64_t getTickCount();
inline double getTickFrequency() { return 1e9; }

}  // namespace cv
#include <chrono>
inline int64_t cv::getTickCount() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#endif
