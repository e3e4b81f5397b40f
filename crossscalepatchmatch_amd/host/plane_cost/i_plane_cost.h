// i_plane_cost.h -- the per-call plane-cost boundary of the reference (CSPM/plane_cost/i_plane_cost.h:13-34),
// plus the bulk interface a device-resident cost object offers to CSPatchMatch.
#pragma once
#include "../commfunc.h"
#include "../plane.h"

struct cspm_ctx;

class IPlaneCost {
 public:
  IPlaneCost() {}
  virtual ~IPlaneCost() {}
  // aggregated slanted-window cost of `plane` at pixel (ref_x, ref_y) of `view`
  virtual double GetPlaneCost(const int &ref_x, const int &ref_y, const Plane &plane, const RefView &view) const = 0;
};

// A plane cost that lives on the GPU.  CSPatchMatch::PatchMatch dynamic_casts for it and then runs the whole
// init / propagation / refinement loop on the device instead of calling GetPlaneCost ~40 times per pixel.
class IDevicePlaneCost {
 public:
  virtual ~IDevicePlaneCost() {}
  virtual cspm_ctx *device_ctx() const = 0;
};
