// pre_cs_pc.h -- PreCSPC: cross-scale plane cost, scale_num pyramid levels coupled by the first row of the
// inverse regularisation matrix (CSPM/plane_cost/pre_cs_pc.h:19-61).  Same constructor as the reference.
#pragma once
#include "../cc_method.h"
#include "device_plane_cost.h"

#define WGT_GAMMA 10.0

class PreCSPC : public DevicePlaneCost {
 public:
  PreCSPC(const Mat &l_img, const Mat &r_img, const int &max_disp, const int &wnd_size, const int &scale_num,
          CCMethod *cc_method, const double &reg_lambda)
      : DevicePlaneCost(l_img, r_img, max_disp, wnd_size, scale_num, cc_method, reg_lambda) {}
};
