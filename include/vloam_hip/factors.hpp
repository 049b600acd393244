// vloam_hip/factors.hpp — the reference's Ceres cost-functor surface, without a Ceres include.
//
// Same structs, constructors and `template <typename T> bool operator()(const T* q, const T* t, T* residual) const`
// as /root/reference/src/lidar_odometry_mapping/include/lidar_odometry_mapping/lidarFactor.hpp:14-139 and
// src/visual_odometry/include/visual_odometry/ceres_cost_function.h:54-96,147-185, written against a tiny
// self-contained vector algebra (no Eigen), so a user who HAS Ceres can still do
//   new ceres::AutoDiffCostFunction<vloam::factors::LidarEdgeFactor, 3, 4, 3>(new vloam::factors::LidarEdgeFactor(...))
// while the GPU path evaluates the same residuals with closed-form Jacobians (csrc/lm_solve.hip).
// T must support + - * / < >, sqrt, sin, cos, acos, abs (found by ADL, as for ceres::Jet).  The interpolation ratio s is honoured
// exactly like the reference (lidarFactor.hpp:26-33: q_last_curr = Identity.slerp(s, q), t_last_curr = s * t); the GPU path runs with
// s == 1 (DISTORTION == false, laser_odometry.h:90), where the slerp returns q itself.
// The static Create(...) factories (lidarFactor.hpp:47-52, 95-101, 129-134; ceres_cost_function.h:87-92, 176-181) return
// ceres::CostFunction*, i.e. need Ceres: they are compiled with -DVLOAM_HIP_WITH_CERES (which includes <ceres/ceres.h>) and are UNTESTED
// here — Ceres is absent from this image.  Points are taken as const double[3] OR as any vector class with operator[] — an
// Eigen::Vector3d as the reference's call sites pass (laser_odometry.cpp:337-347, laser_mapping.cpp:507-510): `Vec3Like` overloads,
// no Eigen include needed (tests/test_cpp_headers.py drives them with a stand-in).
#pragma once
#include <cmath>
#include <limits>
#include <type_traits>
#ifdef VLOAM_HIP_WITH_CERES
#include <ceres/ceres.h>
#endif

namespace vloam {
namespace factors {

// "anything indexable that is a class": Eigen::Vector3d, std::array<double, 3>, ... (plain arrays / pointers take the const double[3] overloads)
template <class V> using Vec3Like = typename std::enable_if<std::is_class<V>::value && std::is_convertible<decltype(std::declval<const V&>()[0]), double>::value, int>::type;

template <class T> struct V3 { T x, y, z; };
template <class T> inline V3<T> sub(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// q = (x, y, z, w); v + w (2 u x v) + u x (2 u x v)   (Eigen's QuaternionBase::_transformVector)
template <class T, class Q> inline V3<T> rotate(const Q* q, const V3<T>& v) {
  V3<T> u{q[0], q[1], q[2]};
  V3<T> uv = cross(u, v);
  uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  V3<T> c = cross(u, uv);
  return {v.x + q[3] * uv.x + c.x, v.y + q[3] * uv.y + c.y, v.z + q[3] * uv.z + c.z};
}

// Eigen::Quaternion::slerp from the identity (Eigen/src/Geometry/Quaternion.h): d = Identity . q = q.w; near-parallel -> linear weights,
// otherwise sin((1 - s) theta) / sin(theta) and sin(s theta) / sin(theta), the second negated when d < 0.
template <class T> inline void slerp_from_identity(const T& s, const T* q, T* out) {
  using std::acos; using std::sin; using std::abs;
  const T one = T(1.0) - T(std::numeric_limits<double>::epsilon());
  const T d = q[3];
  const T absD = abs(d);
  T scale0, scale1;
  if (absD >= one) { scale0 = T(1.0) - s; scale1 = s; }
  else {
    const T theta = acos(absD), sinTheta = sin(theta);
    scale0 = sin((T(1.0) - s) * theta) / sinTheta;
    scale1 = sin(s * theta) / sinTheta;
  }
  if (d < T(0.0)) scale1 = -scale1;
  out[0] = scale1 * q[0]; out[1] = scale1 * q[1]; out[2] = scale1 * q[2]; out[3] = scale0 + scale1 * q[3];
}

struct LidarEdgeFactor {
  LidarEdgeFactor(const double c[3], const double a[3], const double b[3], double s_) : s(s_) { for (int i = 0; i < 3; i++) { cp[i] = c[i]; lpa[i] = a[i]; lpb[i] = b[i]; } }
  template <class V, Vec3Like<V> = 0>   // LidarEdgeFactor(Eigen::Vector3d curr_point_, Eigen::Vector3d last_point_a_, Eigen::Vector3d last_point_b_, double s_), lidarFactor.hpp:16-19
  LidarEdgeFactor(const V& c, const V& a, const V& b, double s_) : s(s_) { for (int i = 0; i < 3; i++) { cp[i] = c[i]; lpa[i] = a[i]; lpb[i] = b[i]; } }
  template <typename T> bool operator()(const T* q, const T* t, T* residual) const {
    using std::sqrt;
    V3<T> p{T(cp[0]), T(cp[1]), T(cp[2])}, a{T(lpa[0]), T(lpa[1]), T(lpa[2])}, b{T(lpb[0]), T(lpb[1]), T(lpb[2])};
    T qs[4];
    slerp_from_identity(T(s), q, qs);   // lidarFactor.hpp:29-33
    V3<T> r = rotate(qs, p);
    V3<T> lp{r.x + T(s) * t[0], r.y + T(s) * t[1], r.z + T(s) * t[2]};
    V3<T> nu = cross(sub(lp, a), sub(lp, b));
    V3<T> de = sub(a, b);
    T n = sqrt(dot(de, de));
    residual[0] = nu.x / n; residual[1] = nu.y / n; residual[2] = nu.z / n;
    return true;
  }
#ifdef VLOAM_HIP_WITH_CERES
  // lidarFactor.hpp:47-52
  template <class V, Vec3Like<V> = 0>
  static ceres::CostFunction* Create(const V& curr_point_, const V& last_point_a_, const V& last_point_b_, const double s_) {   // lidarFactor.hpp:47-52
    return (new ceres::AutoDiffCostFunction<LidarEdgeFactor, 3, 4, 3>(new LidarEdgeFactor(curr_point_, last_point_a_, last_point_b_, s_)));
  }
  static ceres::CostFunction* Create(const double* curr_point_, const double* last_point_a_, const double* last_point_b_, const double s_) {
    return (new ceres::AutoDiffCostFunction<LidarEdgeFactor, 3, 4, 3>(new LidarEdgeFactor(curr_point_, last_point_a_, last_point_b_, s_)));
  }
#endif
  double cp[3], lpa[3], lpb[3], s;
};

struct LidarPlaneFactor {
  LidarPlaneFactor(const double c[3], const double j[3], const double l[3], const double m[3], double s_) : s(s_) {
    for (int i = 0; i < 3; i++) { cp[i] = c[i]; lpj[i] = j[i]; }
    V3<double> a{j[0] - l[0], j[1] - l[1], j[2] - l[2]}, b{j[0] - m[0], j[1] - m[1], j[2] - m[2]};
    V3<double> n = cross(a, b);
    double nn = std::sqrt(dot(n, n));
    ljm[0] = n.x / nn; ljm[1] = n.y / nn; ljm[2] = n.z / nn;
  }
  template <class V, Vec3Like<V> = 0>   // LidarPlaneFactor(Eigen::Vector3d curr_point_, last_point_j_, last_point_l_, last_point_m_, double s_), lidarFactor.hpp:60-70
  LidarPlaneFactor(const V& c, const V& j, const V& l, const V& m, double s_) : s(s_) {
    const double cc[3] = {c[0], c[1], c[2]}, jj[3] = {j[0], j[1], j[2]}, ll[3] = {l[0], l[1], l[2]}, mm[3] = {m[0], m[1], m[2]};
    *this = LidarPlaneFactor(cc, jj, ll, mm, s_);
  }
  template <typename T> bool operator()(const T* q, const T* t, T* residual) const {
    V3<T> p{T(cp[0]), T(cp[1]), T(cp[2])}, j{T(lpj[0]), T(lpj[1]), T(lpj[2])}, n{T(ljm[0]), T(ljm[1]), T(ljm[2])};
    T qs[4];
    slerp_from_identity(T(s), q, qs);   // lidarFactor.hpp:84-88
    V3<T> r = rotate(qs, p);
    V3<T> lp{r.x + T(s) * t[0], r.y + T(s) * t[1], r.z + T(s) * t[2]};
    residual[0] = dot(sub(lp, j), n);
    return true;
  }
#ifdef VLOAM_HIP_WITH_CERES
  // lidarFactor.hpp:95-101
  template <class V, Vec3Like<V> = 0>
  static ceres::CostFunction* Create(const V& curr_point_, const V& last_point_j_, const V& last_point_l_, const V& last_point_m_, const double s_) {
    return (new ceres::AutoDiffCostFunction<LidarPlaneFactor, 1, 4, 3>(new LidarPlaneFactor(curr_point_, last_point_j_, last_point_l_, last_point_m_, s_)));
  }
  static ceres::CostFunction* Create(const double* curr_point_, const double* last_point_j_, const double* last_point_l_, const double* last_point_m_, const double s_) {
    return (new ceres::AutoDiffCostFunction<LidarPlaneFactor, 1, 4, 3>(new LidarPlaneFactor(curr_point_, last_point_j_, last_point_l_, last_point_m_, s_)));
  }
#endif
  double cp[3], lpj[3], ljm[3], s;
};

struct LidarPlaneNormFactor {
  LidarPlaneNormFactor(const double c[3], const double n[3], double d_) : d(d_) { for (int i = 0; i < 3; i++) { cp[i] = c[i]; nrm[i] = n[i]; } }
  template <class V, Vec3Like<V> = 0>   // LidarPlaneNormFactor(Eigen::Vector3d curr_point_, Eigen::Vector3d plane_unit_norm_, double negative_OA_dot_norm_), lidarFactor.hpp:110-113
  LidarPlaneNormFactor(const V& c, const V& n, double d_) : d(d_) { for (int i = 0; i < 3; i++) { cp[i] = c[i]; nrm[i] = n[i]; } }
  template <typename T> bool operator()(const T* q, const T* t, T* residual) const {
    V3<T> p{T(cp[0]), T(cp[1]), T(cp[2])}, n{T(nrm[0]), T(nrm[1]), T(nrm[2])};
    V3<T> r = rotate(q, p);
    V3<T> w{r.x + t[0], r.y + t[1], r.z + t[2]};
    residual[0] = dot(n, w) + T(d);
    return true;
  }
#ifdef VLOAM_HIP_WITH_CERES
  // lidarFactor.hpp:129-134
  template <class V, Vec3Like<V> = 0>
  static ceres::CostFunction* Create(const V& curr_point_, const V& plane_unit_norm_, const double negative_OA_dot_norm_) {
    return (new ceres::AutoDiffCostFunction<LidarPlaneNormFactor, 1, 4, 3>(new LidarPlaneNormFactor(curr_point_, plane_unit_norm_, negative_OA_dot_norm_)));
  }
  static ceres::CostFunction* Create(const double* curr_point_, const double* plane_unit_norm_, const double negative_OA_dot_norm_) {
    return (new ceres::AutoDiffCostFunction<LidarPlaneNormFactor, 1, 4, 3>(new LidarPlaneNormFactor(curr_point_, plane_unit_norm_, negative_OA_dot_norm_)));
  }
#endif
  double cp[3], nrm[3], d;
};

// ceres::AngleAxisRotatePoint
template <class T> inline void angle_axis_rotate(const T* w_, const T* pt, T* out) {
  using std::sqrt; using std::sin; using std::cos;
  const T th2 = w_[0] * w_[0] + w_[1] * w_[1] + w_[2] * w_[2];
  if (th2 > T(std::numeric_limits<double>::epsilon())) {
    const T th = sqrt(th2), c = cos(th), s = sin(th), ti = T(1.0) / th;
    const T w[3] = {w_[0] * ti, w_[1] * ti, w_[2] * ti};
    const T x[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - c);
    for (int k = 0; k < 3; k++) out[k] = pt[k] * c + x[k] * s + w[k] * tmp;
  } else {
    const T x[3] = {w_[1] * pt[2] - w_[2] * pt[1], w_[2] * pt[0] - w_[0] * pt[2], w_[0] * pt[1] - w_[1] * pt[0]};
    for (int k = 0; k < 3; k++) out[k] = pt[k] + x[k];
  }
}

struct CostFunctor32 {  // 3D - 2D
  CostFunctor32(double x0_, double y0_, double z0_, double x1_bar_, double y1_bar_) : x0(x0_), y0(y0_), z0(z0_), x1_bar(x1_bar_), y1_bar(y1_bar_) {}
  template <typename T> bool operator()(const T* const angles, const T* const t, T* residuals) const {
    T X0[3] = {T(x0), T(y0), T(z0)}, X1[3];
    angle_axis_rotate(angles, X0, X1);
    X1[0] = X1[0] + t[0]; X1[1] = X1[1] + t[1]; X1[2] = X1[2] + t[2];
    residuals[0] = X1[0] - X1[2] * T(x1_bar);
    residuals[1] = X1[1] - X1[2] * T(y1_bar);
    return true;
  }
#ifdef VLOAM_HIP_WITH_CERES
  // ceres_cost_function.h:87-92
  static ceres::CostFunction* Create(const double observed_x0, const double observed_y0, const double observed_z0, const double observed_x1_bar, const double observed_y1_bar) {
    return (new ceres::AutoDiffCostFunction<CostFunctor32, 2, 3, 3>(new CostFunctor32(observed_x0, observed_y0, observed_z0, observed_x1_bar, observed_y1_bar)));
  }
#endif
  double x0, y0, z0, x1_bar, y1_bar;
};

struct CostFunctor22 {  // 2D - 2D epipolar
  CostFunctor22(double x0_bar_, double y0_bar_, double x1_bar_, double y1_bar_) : x0_bar(x0_bar_), y0_bar(y0_bar_), x1_bar(x1_bar_), y1_bar(y1_bar_) {}
  template <typename T> bool operator()(const T* const angles, const T* const t, T* residuals) const {
    T X0[3] = {T(x0_bar), T(y0_bar), T(1.0)}, R[3];
    angle_axis_rotate(angles, X0, R);
    const T c[3] = {t[1] * R[2] - t[2] * R[1], t[2] * R[0] - t[0] * R[2], t[0] * R[1] - t[1] * R[0]};
    residuals[0] = T(x1_bar) * c[0] + T(y1_bar) * c[1] + c[2];
    return true;
  }
#ifdef VLOAM_HIP_WITH_CERES
  // ceres_cost_function.h:176-181
  static ceres::CostFunction* Create(const double observed_x0_bar, const double observed_y0_bar, const double observed_x1_bar, const double observed_y1_bar) {
    return (new ceres::AutoDiffCostFunction<CostFunctor22, 1, 3, 3>(new CostFunctor22(observed_x0_bar, observed_y0_bar, observed_x1_bar, observed_y1_bar)));
  }
#endif
  double x0_bar, y0_bar, x1_bar, y1_bar;
};

}  // namespace factors
}  // namespace vloam
