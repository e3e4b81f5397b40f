// cc_method.h -- cost-computation plugin interface, as CSPM/cc_method.h:15-33.
#pragma once
#include "commfunc.h"

class CCMethod {
 public:
  CCMethod() {}
  virtual ~CCMethod() {}
  // lImg / rImg: CV_64FC3 RGB 0..255; costVol: caller-allocated array of maxDis zeroed CV_64FC1 Mats, slab d = disparity d
  virtual void buildCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *costVol) = 0;
  virtual void buildRightCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *rCostVol) = 0;
};
