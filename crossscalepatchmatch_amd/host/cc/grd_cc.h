// grd_cc.h -- GrdCC: truncated colour + x-gradient matching cost (CSPM/cc/grd_cc.h:6-27, grd_cc.cpp:60-154).
// buildCV / buildRightCV keep the host-buffer contract of CCMethod and run the GRD kernel of libcspm_hip.so
// (cspm_grd_build_cv_host); PreSSPC / PreCSPC recognise a GrdCC and build the cost on the device without the
// host round trip.
#pragma once
#include "../cc_method.h"

#define BORDER_THRES 3
#define TAU_CLR 10.0
#define TAU_GRD 2.0
#define ALPHA 0.1

class GrdCC : public CCMethod {
 public:
  // device < 0 (default): the GPU of the calling thread's DeviceSlot at the time of the call (plane_cost/device_plane_cost.h)
  explicit GrdCC(int device = -1) : device_(device) {}
  ~GrdCC() {}
  void buildCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *costVol);
  void buildRightCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *rCostVol);

 private:
  void build(const Mat &lImg, const Mat &rImg, int maxDis, Mat *vol, int right);
  int device_;
};
