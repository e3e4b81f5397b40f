// cen_cc.h -- CenCC: 9x9 census / Hamming matching cost (CSPM/cc/cen_cc.h:5-24, cen_cc.cc:4-137), second
// CCMethod behind the same plugin slot.  buildCV / buildRightCV keep the host-buffer contract and run the census
// kernels of libcspm_hip.so; PreSSPC / PreCSPC recognise a CenCC and build its volumes on the device.
#pragma once
#include "../cc_method.h"

#define CENCUS_WND 9
#define CENCUS_BIT 80

class CenCC : public CCMethod {
 public:
  // device < 0 (default): the GPU of the calling thread's DeviceSlot at the time of the call (plane_cost/device_plane_cost.h)
  explicit CenCC(int device = -1) : device_(device) {}
  ~CenCC() {}
  void buildCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *costVol);
  void buildRightCV(const Mat &lImg, const Mat &rImg, const int maxDis, Mat *rCostVol);

 private:
  void build(const Mat &lImg, const Mat &rImg, int maxDis, Mat *vol, int right);
  int device_;
};
