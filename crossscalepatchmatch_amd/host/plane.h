// plane.h -- disparity plane d(x, y) = a*x + b*y + c kept as (unit normal, anchor point, derived a/b/c).
// Interface-compatible with the reference's class Plane (CSPM/plane.h:10-49): same constructors, setters, getters.
#pragma once
#include "commfunc.h"

class Plane {
  Vec3d n_;      // unit normal
  Point3d at_;   // anchor: (x, y, disparity)
  Vec3d abc_;    // derived parameters

 public:
  Plane() {}
  Plane(const Vec3d &norm, const Point3d &point) : n_(norm), at_(point) { update_param(); }

  // getters (by value, like the reference)
  Vec3d norm() const { return n_; }
  Point3d point() const { return at_; }
  Vec3d param() const { return abc_; }

  // setters; callers re-derive the parameters with update_param() (cs_patchmatch.cc:141,263-265,329-331)
  void set_norm(const Point3d &norm) { n_ = norm; }
  void set_point(const Point3d &point) { at_ = point; }

  // (a, b, c) = (-nx, -ny, n.p) / nz with |nz| clamped away from zero, sign kept (plane.h:25-34)
  void update_param() {
    double nz = n_[2] < 0.0 ? -1.0 : 1.0;
    nz *= std::max(std::fabs(n_[2]), kDoubleEps);
    abc_[0] = -n_[0] / nz;
    abc_[1] = -n_[1] / nz;
    abc_[2] = n_.dot(at_) / nz;
  }

  // addition: adopt parameters that were computed on the device
  void set_param(const Vec3d &param) { abc_ = param; }
};
