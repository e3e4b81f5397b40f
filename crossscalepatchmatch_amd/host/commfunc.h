// commfunc.h -- shared constants and scalar helpers of the host layer; same names and meaning as the
// reference's CSPM/commfunc.h:24-29,117-145 so plugin code written against it compiles unchanged.
#pragma once
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>

#include "cv_compat.h"
#include "gflags_compat.h"

using namespace std;
using namespace cv;

const int kViewNum = 2;                                   // commfunc.h:24
const double kDoubleEps = 0.00000001;                     // commfunc.h:26
const double kDoubleMax = numeric_limits<double>::max();  // commfunc.h:27
enum RefView { kLeft = 0, kRight = 1 };                   // commfunc.h:29

// commfunc.h:117-121: round-half-to-even through the 2^52+2^51 magic constant
inline int Round2Int(double d) {
  d += 6755399441055744.0;
  int32_t lo;
  std::memcpy(&lo, &d, sizeof lo);
  return lo;
}
// commfunc.h:129-145: a single wrap-around
inline int HandleBorder(const int &loc, const int &size) { return loc < 0 ? loc + size : (loc >= size ? loc - size : loc); }
