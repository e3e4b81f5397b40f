// device_plane_cost.h -- common implementation of PreSSPC / PreCSPC / GrdPC / CSPC above the C ABI (include/cspm.h).
#pragma once
#include <vector>
#include "../cc_method.h"
#include "i_plane_cost.h"

class DevicePlaneCost : public IPlaneCost, public IDevicePlaneCost {
 public:
  // scale_num == 0: PreSSPC (pre_ss_pc.cc:12-65); >= 1: PreCSPC (pre_cs_pc.cc:12-115).  cc_method is borrowed.
  DevicePlaneCost(const Mat &l_img, const Mat &r_img, int max_disp, int wnd_size, int scale_num, CCMethod *cc_method,
                  double reg_lambda);
  // the volume-free variants, no CCMethod: scale_num == 0: GrdPC (grd_pc.cc:11-66); >= 1: CSPC (cspc.cc:11-93)
  DevicePlaneCost(const Mat &l_img, const Mat &r_img, int max_disp, int wnd_size, int scale_num, double reg_lambda);
  ~DevicePlaneCost();
  virtual double GetPlaneCost(const int &ref_x, const int &ref_y, const Plane &plane, const RefView &view) const;
  virtual cspm_ctx *device_ctx() const { return ctx_; }
  static int device;  // GPU used by objects constructed from now on (the CLI's --device)
  // batch mode: a destroyed object parks its cspm_ctx (device buffers included) for the next object on the same GPU,
  // so a stream of equally sized pairs allocates once
  static bool keep_context;
  static void release_kept_context();
  // is this context still owned by a live (or parked) DevicePlaneCost?  CSPatchMatch borrows the context of the cost object it
  // ran on (planes(), disparity()) and must not touch it once that object is gone
  static bool is_live(const cspm_ctx *ctx);
  static void adopt(cspm_ctx *ctx);   // a context owned by someone else (CSPatchMatch's own, for a foreign IPlaneCost) joins / leaves
  static void disown(cspm_ctx *ctx);  // the registry

 private:
  DevicePlaneCost(const DevicePlaneCost &);
  void upload_foreign(CCMethod *cc, int view, int level);
  void open_context(const Mat &l_img, const Mat &r_img);
  cspm_ctx *ctx_;
  int ctx_device_;
  static cspm_ctx *kept_ctx_;
  static int kept_device_;
  static std::vector<cspm_ctx *> live_;
};
