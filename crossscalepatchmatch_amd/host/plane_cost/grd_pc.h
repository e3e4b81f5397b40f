// grd_pc.h -- GrdPC: single-scale plane cost that interpolates the other view's colour and x-gradient at the real-valued
// column x -+ q_disp, no cost volumes, no CCMethod (CSPM/plane_cost/grd_pc.h:25-73, grd_pc.cc).  Same constructor as the
// reference.  Its main.cc includes this header but keeps the instantiation commented out (main.cc:106-107); cspm_main reaches
// the class with --pc_name=IMG.
#pragma once
#include "device_plane_cost.h"

#define COST_ALPHA 0.1
#define TAU_CLR 10.0
#define TAU_GRD 2.0
#define WGT_GAMMA  10.0

class GrdPC : public DevicePlaneCost {
 public:
  GrdPC(const Mat &l_img, const Mat &r_img, const int &max_disp, const int &wnd_size)
      : DevicePlaneCost(l_img, r_img, max_disp, wnd_size, 0, 0.0) {}
};
