// plane.h -- class Plane with the reference's interface (CSPM/plane.h:10-49): unit normal + anchor point,
// derived disparity-plane parameters d(x,y) = a*x + b*y + c.
#pragma once
#include "commfunc.h"

class Plane {
 public:
  Plane() {}
  Plane(const Vec3d &norm, const Point3d &point) : norm_(norm), point_(point) { update_param(); }
  void set_point(const Point3d &point) { point_ = point; }
  void set_norm(const Point3d &norm) { norm_ = norm; }
  // plane.h:25-34: the denominator keeps the sign of nz and never gets closer to zero than kDoubleEps
  void update_param() {
    const double mag = std::max(std::fabs(norm_[2]), kDoubleEps);
    const double denom = norm_[2] < 0.0 ? -mag : mag;
    param_ = Vec3d(-norm_[0] / denom, -norm_[1] / denom, norm_.dot(point_) / denom);
  }
  Vec3d norm() const { return norm_; }
  Point3d point() const { return point_; }
  Vec3d param() const { return param_; }
  // not in the reference: adopt parameters computed on the device
  void set_param(const Vec3d &param) { param_ = param; }

 private:
  Vec3d norm_, param_;
  Point3d point_;
};
