// pre_ss_pc.h -- PreSSPC: single-scale plane cost over a precomputed cost volume
// (CSPM/plane_cost/pre_ss_pc.h:18-50).  Same constructor as the reference; the object owns a cspm_ctx.
#pragma once
#include "../cc_method.h"
#include "device_plane_cost.h"

class PreSSPC : public DevicePlaneCost {
 public:
  PreSSPC(const Mat &l_img, const Mat &r_img, const int &max_disp, const int &wnd_size, CCMethod *cc_method)
      : DevicePlaneCost(l_img, r_img, max_disp, wnd_size, /*scale_num=*/0, cc_method, 0.0) {}
};
