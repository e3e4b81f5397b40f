// ca_method.h -- cost-aggregation plugin interface, as CSPM/ca_method.h:8-25.  PatchMatch never calls it (the
// reference's ca_filter/* is not compiled into CSPM.vcxproj); kept as a header-level surface only.
#pragma once
#include "commfunc.h"

class CAMethod {
 public:
  CAMethod() {}
  virtual ~CAMethod() {}
  virtual void aggreCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *costVol) = 0;
};
