// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not linked into libvloam_hip.so.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PARITY UNPINNED: the reference (YukunXia/VLOAM-CMU-16833) has no tests and its
// arithmetic lives in Eigen 3.3 / Ceres 2.0 / PCL 1.10 / FLANN 1.9, none of which
// exist in this image.  This header restates the handful of Eigen operations the
// hot path uses, following the *published* Eigen 3.3 algorithms named per function.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <algorithm>

namespace orc {

// ---------------------------------------------------------------- vectors
template <class T>
struct V3 {
  T x, y, z;
  V3() : x(T(0)), y(T(0)), z(T(0)) {}
  V3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> inline V3<T> operator*(const T& s, const V3<T>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> inline V3<T> operator/(const V3<T>& a, const T& s) { return {a.x / s, a.y / s, a.z / s}; }
// Eigen 3.3 MatrixBase::cross (Geometry/OrthoMethods.h): component formula, no FMA.
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> inline T norm(const V3<T>& a) { using std::sqrt; return sqrt(dot(a, a)); }

// ---------------------------------------------------------------- quaternion
// Coefficient order x,y,z,w == Eigen::Quaternion::coeffs() == the reference's
// para_q[4] / parameters[0..3] layout (laser_odometry.h:126-130, laser_mapping.h:141).
template <class T>
struct Quat {
  T x, y, z, w;
  Quat() : x(T(0)), y(T(0)), z(T(0)), w(T(1)) {}
  Quat(T x_, T y_, T z_, T w_) : x(x_), y(y_), z(z_), w(w_) {}
};

// Eigen 3.3 QuaternionBase::_transformVector (Geometry/Quaternion.h):
//   uv = vec().cross(v); uv += uv; return v + w()*uv + vec().cross(uv);
// This is what `q * point` evaluates to in TransformToStart (laser_odometry.cpp:161),
// pointAssociateToMap (laser_mapping.cpp:149) and inside every lidarFactor functor.
template <class T> inline V3<T> rotate(const Quat<T>& q, const V3<T>& v) {
  V3<T> u(q.x, q.y, q.z);
  V3<T> uv = cross(u, v);
  uv = uv + uv;
  V3<T> wuv = q.w * uv;
  V3<T> c = cross(u, uv);
  return V3<T>((v.x + wuv.x) + c.x, (v.y + wuv.y) + c.y, (v.z + wuv.z) + c.z);
}

// Eigen 3.3 generic quat_product (Geometry/Quaternion.h internal::quat_product).
template <class T> inline Quat<T> qmul(const Quat<T>& a, const Quat<T>& b) {
  return Quat<T>(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                 a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
                 a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
// Eigen 3.3 QuaternionBase::inverse(): conjugate / squaredNorm.
inline Quat<double> qinverse(const Quat<double>& q) {
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (n2 > 0.0) return Quat<double>(-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2);
  return Quat<double>(0, 0, 0, 0);
}

// Eigen 3.3 QuaternionBase::slerp (Geometry/Quaternion.h). Called as
// Identity.slerp(s, q) in TransformToStart (laser_odometry.cpp:158) and in
// LidarEdgeFactor / LidarPlaneFactor (lidarFactor.hpp:29-31, 79-81).  Templated so
// the Jet instantiation differentiates through it exactly as Ceres autodiff does.
template <class T> inline Quat<T> slerp_from_identity(const T& t, const Quat<T>& other, double eps) {
  using std::acos; using std::sin; using std::abs;
  const T one = T(1) - T(eps);
  T d = other.w;  // Identity.dot(other)
  T absD = abs(d);
  T scale0, scale1;
  if (absD >= one) {
    scale0 = T(1) - t;
    scale1 = t;
  } else {
    T theta = acos(absD);
    T sinTheta = sin(theta);
    scale0 = sin((T(1) - t) * theta) / sinTheta;
    scale1 = sin((t * theta)) / sinTheta;
  }
  if (d < T(0)) scale1 = -scale1;
  // scale0 * Identity.coeffs() + scale1 * other.coeffs()
  return Quat<T>(scale0 * T(0) + scale1 * other.x, scale0 * T(0) + scale1 * other.y,
                 scale0 * T(0) + scale1 * other.z, scale0 * T(1) + scale1 * other.w);
}

// ---------------------------------------------------------------- 3x3 symmetric eigen
// Stand-in for Eigen::SelfAdjointEigenSolver<Matrix3d> (laser_mapping.cpp:500):
// eigenvalues ascending, unit eigenvectors in columns; eigenvector sign is arbitrary
// (immaterial: swapping a<->b only flips the edge residual's sign).  Cyclic Jacobi,
// converges to ~1e-16 relative; Eigen's QL iteration agrees to rounding.
inline void sym_eig3(const double A_[3][3], double evals[3], double evecs[3][3]) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = A_[i][j];
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (A[p][q] == 0.0) continue;
      double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) {  // A <- A * G
        double akp = A[k][p], akq = A[k][q];
        A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; k++) {  // A <- G^T * A
        double apk = A[p][k], aqk = A[q][k];
        A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; k++) {
        double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
      }
    }
  }
  int idx[3] = {0, 1, 2};
  double d[3] = {A[0][0], A[1][1], A[2][2]};
  std::sort(idx, idx + 3, [&](int a, int b) { return d[a] < d[b]; });
  for (int k = 0; k < 3; k++) {
    evals[k] = d[idx[k]];
    for (int r = 0; r < 3; r++) evecs[r][k] = V[r][idx[k]];
  }
}

// ---------------------------------------------------------------- small dense LS
// Householder QR least squares  min ||A x - b||, A is m x n row-major (m >= n).
// Stand-in for Eigen's householderQr().solve (Ceres DENSE_QR) and
// colPivHouseholderQr().solve (laser_mapping.cpp:557; with full column rank the
// pivoted and unpivoted factorisations give the same LS solution to rounding).
// A and b are overwritten.  Returns false if R has a zero pivot.
inline bool householder_ls(double* A, double* b, int m, int n, double* x) {
  for (int k = 0; k < n; k++) {
    double nrm = 0;
    for (int i = k; i < m; i++) nrm += A[i * n + k] * A[i * n + k];
    nrm = std::sqrt(nrm);
    if (nrm == 0.0) return false;
    double alpha = (A[k * n + k] > 0) ? -nrm : nrm;
    double v0 = A[k * n + k] - alpha;
    // v = (v0, A[k+1..m-1][k]); H = I - 2 v v^T / (v^T v)
    double vtv = v0 * v0;
    for (int i = k + 1; i < m; i++) vtv += A[i * n + k] * A[i * n + k];
    if (vtv != 0.0) {
      for (int j = k + 1; j < n; j++) {
        double s = v0 * A[k * n + j];
        for (int i = k + 1; i < m; i++) s += A[i * n + k] * A[i * n + j];
        s = 2.0 * s / vtv;
        A[k * n + j] -= s * v0;
        for (int i = k + 1; i < m; i++) A[i * n + j] -= s * A[i * n + k];
      }
      double s = v0 * b[k];
      for (int i = k + 1; i < m; i++) s += A[i * n + k] * b[i];
      s = 2.0 * s / vtv;
      b[k] -= s * v0;
      for (int i = k + 1; i < m; i++) b[i] -= s * A[i * n + k];
    }
    A[k * n + k] = alpha;
    for (int i = k + 1; i < m; i++) A[i * n + k] = 0.0;
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
    if (A[k * n + k] == 0.0) return false;
    x[k] = s / A[k * n + k];
  }
  return true;
}

}  // namespace orc
