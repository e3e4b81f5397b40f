// cspc.h -- CSPC: the cross-scale version of GrdPC -- scale_num pyramid levels of on-the-fly colour + gradient costs coupled
// by the first row of the inverse regularisation matrix (CSPM/plane_cost/cspc.h:19-61, cspc.cc).  Same constructor as the
// reference; cspm_main reaches the class with --pc_name=IMG --use_cs=true.
#pragma once
#include "device_plane_cost.h"

#define COST_ALPHA 0.1
#define TAU_CLR 10.0
#define TAU_GRD 2.0
#define WGT_GAMMA  10.0

class CSPC : public DevicePlaneCost {
 public:
  CSPC(const Mat &l_img, const Mat &r_img, const int &max_disp, const int &wnd_size, const int &scale_num, const double &reg_lambda)
      : DevicePlaneCost(l_img, r_img, max_disp, wnd_size, scale_num, reg_lambda) {
    if (scale_num < 1) throw std::runtime_error("CSPC: scale_num must be >= 1");
  }
};
