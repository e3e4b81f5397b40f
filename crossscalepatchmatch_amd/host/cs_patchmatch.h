// cs_patchmatch.h -- class CSPatchMatch with the reference's public interface (CSPM/cs_patchmatch.h:17-68).
#pragma once
#include "commfunc.h"
#include "plane.h"
#include "plane_cost/i_plane_cost.h"

#define WMF_GAMMA 10.0

class CSPatchMatch {
 public:
  CSPatchMatch(const Mat &l_img, const Mat &r_img, const int &max_dis, const int &dis_scale);
  ~CSPatchMatch();
  // init, iter_num x (spatial, view, refinement), PlaneToDisp, optional PostProcessing (cs_patchmatch.cc:51-109).
  // A device cost (PreSSPC / PreCSPC / GrdPC / CSPC of this host layer) runs the whole loop on its GPU.  Any other IPlaneCost
  // (a plugin: i_plane_cost.h:28-33) is priced through its GetPlaneCost, candidate batch by candidate batch, while the plane
  // field, the random streams and the accept rules stay on the device (PatchMatchForeign).
  void PatchMatch(const int &iter_num, const IPlaneCost *plane_cost, const bool &use_pp);
  Mat &dis(const RefView &view) { return dis_[view]; }
  // PatchMatch in two halves for callers that overlap their own work (file decoding, the previous pair's encoding) with the GPU:
  // Begin enqueues the whole loop on the cost object's stream and returns (a device cost; a foreign IPlaneCost runs to completion here),
  // End waits for it and fills dis() -- PlaneToDisp, or PostProcessing when use_pp.  PatchMatch == Begin + End.
  void PatchMatchBegin(const int &iter_num, const IPlaneCost *plane_cost, const bool &use_pp);
  void PatchMatchEnd();

  // additions (the reference seeds from time(NULL) and has one schedule)
  void set_seed(uint64_t seed) { seed_ = seed; }
  void set_schedule(int schedule, int rb_rounds = 1) { schedule_ = schedule; rb_rounds_ = rb_rounds; }
  // final plane field of a view, for callers that want sub-pixel disparities.  Both read the device context of the plane cost
  // the last PatchMatch ran on: call them while that object is alive (they throw otherwise).  They return the PLANE field --
  // PostProcessing (use_pp) works on the 8-bit maps only (cs_patchmatch.cc:508-588) and does not change it.
  void planes(const RefView &view, std::vector<Plane> *out, std::vector<double> *min_cost) const;
  // unquantised disparity a*x+b*y+c of every pixel, row-major (what PlaneToDisp rounds, cs_patchmatch.cc:590-601)
  void disparity(const RefView &view, std::vector<double> *out) const;

 private:
  Mat img_[kViewNum], dis_[kViewNum];
  int wid_, hei_, max_dis_, dis_scale_;
  uint64_t seed_;
  int schedule_, rb_rounds_;
  cspm_ctx *last_ctx_;
  cspm_ctx *own_ctx_;  // foreign IPlaneCost: the context that holds the plane field
  cspm_ctx *pending_ctx_;  // PatchMatchBegin without its PatchMatchEnd yet
  bool pending_pp_;
  void PatchMatchForeign(int iter_num, const IPlaneCost *plane_cost, bool use_pp);
  CSPatchMatch(const CSPatchMatch &);
};
