// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// The reference's Ceres cost functors, restated with the oracle's own tiny vector / quaternion
// templates (Eigen is not available).  Same operator() contract:
//   template<class T> bool operator()(const T* q /*x,y,z,w*/, const T* t, T* residual) const
//   lidarFactor.hpp:14-56   LidarEdgeFactor      (3 residuals)
//   lidarFactor.hpp:58-106  LidarPlaneFactor     (1 residual)
//   lidarFactor.hpp:108-139 LidarPlaneNormFactor (1 residual)
//   ceres_cost_function.h:54-96   CostFunctor32  (2 residuals, angle-axis + t)
//   ceres_cost_function.h:147-185 CostFunctor22  (1 residual,  angle-axis + t)
#pragma once
#include <limits>
#include "orc_ceres.h"
#include "orc_math.h"

namespace orc {

typedef V3<double> Vec3d;

struct LidarEdgeFactor {
  Vec3d curr_point, last_point_a, last_point_b;
  double s;
  LidarEdgeFactor(Vec3d c, Vec3d a, Vec3d b, double s_) : curr_point(c), last_point_a(a), last_point_b(b), s(s_) {}
  template <class T>
  bool operator()(const T* q, const T* t, T* residual) const {
    V3<T> cp(T(curr_point.x), T(curr_point.y), T(curr_point.z));
    V3<T> lpa(T(last_point_a.x), T(last_point_a.y), T(last_point_a.z));
    V3<T> lpb(T(last_point_b.x), T(last_point_b.y), T(last_point_b.z));
    Quat<T> q_last_curr(q[0], q[1], q[2], q[3]);
    q_last_curr = slerp_from_identity(T(s), q_last_curr, std::numeric_limits<double>::epsilon());
    V3<T> t_last_curr(T(s) * t[0], T(s) * t[1], T(s) * t[2]);
    V3<T> lp = rotate(q_last_curr, cp) + t_last_curr;
    V3<T> nu = cross(lp - lpa, lp - lpb);
    V3<T> de = lpa - lpb;
    residual[0] = nu.x / norm(de);
    residual[1] = nu.y / norm(de);
    residual[2] = nu.z / norm(de);
    return true;
  }
};

struct LidarPlaneFactor {
  Vec3d curr_point, last_point_j, last_point_l, last_point_m, ljm_norm;
  double s;
  LidarPlaneFactor(Vec3d c, Vec3d j, Vec3d l, Vec3d m, double s_)
      : curr_point(c), last_point_j(j), last_point_l(l), last_point_m(m), s(s_) {
    ljm_norm = cross(last_point_j - last_point_l, last_point_j - last_point_m);
    ljm_norm = ljm_norm / norm(ljm_norm);  // Eigen normalize(): *this /= norm()
  }
  template <class T>
  bool operator()(const T* q, const T* t, T* residual) const {
    V3<T> cp(T(curr_point.x), T(curr_point.y), T(curr_point.z));
    V3<T> lpj(T(last_point_j.x), T(last_point_j.y), T(last_point_j.z));
    V3<T> ljm(T(ljm_norm.x), T(ljm_norm.y), T(ljm_norm.z));
    Quat<T> q_last_curr(q[0], q[1], q[2], q[3]);
    q_last_curr = slerp_from_identity(T(s), q_last_curr, std::numeric_limits<double>::epsilon());
    V3<T> t_last_curr(T(s) * t[0], T(s) * t[1], T(s) * t[2]);
    V3<T> lp = rotate(q_last_curr, cp) + t_last_curr;
    residual[0] = dot(lp - lpj, ljm);
    return true;
  }
};

struct LidarPlaneNormFactor {
  Vec3d curr_point, plane_unit_norm;
  double negative_OA_dot_norm;
  LidarPlaneNormFactor(Vec3d c, Vec3d n, double d) : curr_point(c), plane_unit_norm(n), negative_OA_dot_norm(d) {}
  template <class T>
  bool operator()(const T* q, const T* t, T* residual) const {
    Quat<T> q_w_curr(q[0], q[1], q[2], q[3]);
    V3<T> t_w_curr(t[0], t[1], t[2]);
    V3<T> cp(T(curr_point.x), T(curr_point.y), T(curr_point.z));
    V3<T> point_w = rotate(q_w_curr, cp) + t_w_curr;
    V3<T> nrm(T(plane_unit_norm.x), T(plane_unit_norm.y), T(plane_unit_norm.z));
    residual[0] = dot(nrm, point_w) + T(negative_OA_dot_norm);
    return true;
  }
};

// VO — 3D-2D: observed_x0,y0,z0 = 3-D point in the previous camera frame; observed_x1_bar,y1_bar =
// normalised image coordinates in the current frame (ceres_cost_function.h:54-96).
struct CostFunctor32 {
  double x0, y0, z0, x1_bar, y1_bar;
  CostFunctor32(double a, double b, double c, double d, double e) : x0(a), y0(b), z0(c), x1_bar(d), y1_bar(e) {}
  template <class T>
  bool operator()(const T* angles, const T* t, T* residuals) const {
    T X0[3] = {T(x0), T(y0), T(z0)};
    T X1[3];
    AngleAxisRotatePoint(angles, X0, X1);
    X1[0] = X1[0] + t[0]; X1[1] = X1[1] + t[1]; X1[2] = X1[2] + t[2];
    residuals[0] = X1[0] - X1[2] * T(x1_bar);
    residuals[1] = X1[1] - X1[2] * T(y1_bar);
    return true;
  }
};

// VO — 2D-2D epipolar (ceres_cost_function.h:147-185).
struct CostFunctor22 {
  double x0_bar, y0_bar, x1_bar, y1_bar;
  CostFunctor22(double a, double b, double c, double d) : x0_bar(a), y0_bar(b), x1_bar(c), y1_bar(d) {}
  template <class T>
  bool operator()(const T* angles, const T* t, T* residuals) const {
    T X0[3] = {T(x0_bar), T(y0_bar), T(1.0)};
    T RX0[3];
    AngleAxisRotatePoint(angles, X0, RX0);
    T c[3] = {t[1] * RX0[2] - t[2] * RX0[1], t[2] * RX0[0] - t[0] * RX0[2], t[0] * RX0[1] - t[1] * RX0[0]};
    residuals[0] = T(x1_bar) * c[0] + T(y1_bar) * c[1] + c[2];
    return true;
  }
};

}  // namespace orc
