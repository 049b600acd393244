// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// Restatement of the slice of Ceres Solver 2.0.0 the reference's hot path drives:
//   ceres::Problem with two parameter blocks {q[4] + EigenQuaternionParameterization, t[3]}
//   (laser_odometry.cpp:217-258, laser_mapping.cpp:461-467) or {angle_axis[3], t[3]}
//   (visual_odometry.cpp:258-259), AutoDiffCostFunction residual blocks sharing one
//   HuberLoss(0.1), Solve() with DENSE_QR, max_num_iterations = 4 (LO / mapping) or 100 (VO).
// Ceres is an un-vendored dependency (README.md:24 "Ceres 2.0"); this file follows its published
// algorithm: internal/ceres/{trust_region_minimizer,levenberg_marquardt_strategy,
// trust_region_step_evaluator,corrector,loss_function,local_parameterization,residual_block,
// dense_qr_solver}.cc and include/ceres/jet.h of the 2.0.0 release.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>
#include "orc_math.h"

namespace orc {

// ------------------------------------------------------------------ Jet (ceres/jet.h)
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; i++) v[i] = 0.0; }
  Jet(double a_) : a(a_) { for (int i = 0; i < N; i++) v[i] = 0.0; }  // NOLINT implicit
  Jet(double a_, int k) : a(a_) { for (int i = 0; i < N; i++) v[i] = 0.0; v[k] = 1.0; }
};
#define ORC_JET_BIN(expr_a, expr_v)                          \
  Jet<N> h; h.a = (expr_a); for (int i = 0; i < N; i++) h.v[i] = (expr_v); return h;
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { ORC_JET_BIN(f.a + g.a, f.v[i] + g.v[i]) }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { ORC_JET_BIN(f.a - g.a, f.v[i] - g.v[i]) }
template <int N> inline Jet<N> operator-(const Jet<N>& f) { ORC_JET_BIN(-f.a, -f.v[i]) }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { ORC_JET_BIN(f.a * g.a, f.a * g.v[i] + f.v[i] * g.a) }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  // jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
  const double ginv = 1.0 / g.a;
  const double fbg = f.a * ginv;
  ORC_JET_BIN(fbg, (f.v[i] - fbg * g.v[i]) * ginv)
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { ORC_JET_BIN(s * f.a, s * f.v[i]) }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator>=(const Jet<N>& f, const Jet<N>& g) { return f.a >= g.a; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline Jet<N> sqrt(const Jet<N>& f) { const double t = std::sqrt(f.a); const double s = 1.0 / (2.0 * t); ORC_JET_BIN(t, s * f.v[i]) }
template <int N> inline Jet<N> sin(const Jet<N>& f) { const double c = std::cos(f.a); ORC_JET_BIN(std::sin(f.a), c * f.v[i]) }
template <int N> inline Jet<N> cos(const Jet<N>& f) { const double s = -std::sin(f.a); ORC_JET_BIN(std::cos(f.a), s * f.v[i]) }
template <int N> inline Jet<N> acos(const Jet<N>& f) { const double t = -1.0 / std::sqrt(1.0 - f.a * f.a); ORC_JET_BIN(std::acos(f.a), t * f.v[i]) }
template <int N> inline Jet<N> abs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
#undef ORC_JET_BIN

// ------------------------------------------------------------------ cost functions
// Two parameter blocks, sizes (size0, 3).  Jacobians row-major [nres x size].
struct CostFunction {
  int nres = 0;
  virtual ~CostFunction() {}
  virtual void Evaluate(const double* p0, const double* p1, double* residuals, double* jac0, double* jac1) const = 0;
};

// ceres::AutoDiffCostFunction<F, NRES, N0, 3>: evaluates F::operator()<Jet<N0+3>>.
template <class F, int NRES, int N0>
struct AutoDiffCost : CostFunction {
  F f;
  explicit AutoDiffCost(const F& f_) : f(f_) { nres = NRES; }
  void Evaluate(const double* p0, const double* p1, double* residuals, double* jac0, double* jac1) const override {
    if (!jac0 && !jac1) {
      f(p0, p1, residuals);
      return;
    }
    typedef Jet<N0 + 3> J;
    J x0[N0], x1[3], r[NRES];
    for (int i = 0; i < N0; i++) x0[i] = J(p0[i], i);
    for (int i = 0; i < 3; i++) x1[i] = J(p1[i], N0 + i);
    f(x0, x1, r);
    for (int k = 0; k < NRES; k++) {
      residuals[k] = r[k].a;
      if (jac0) for (int i = 0; i < N0; i++) jac0[k * N0 + i] = r[k].v[i];
      if (jac1) for (int i = 0; i < 3; i++) jac1[k * 3 + i] = r[k].v[N0 + i];
    }
  }
};

// ------------------------------------------------------------------ solver
struct IterationSummary {
  int iteration;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  bool step_is_valid, step_is_successful;
};
struct SolveSummary {
  std::vector<IterationSummary> iterations;
  double initial_cost = 0, final_cost = 0;
  int termination = 0;  // 0 NO_CONVERGENCE (iteration cap), 1 CONVERGENCE, 2 FAILURE
  int num_residual_blocks = 0, num_residuals = 0;
  // debug capture at iteration 0 (unscaled tangent-space Jacobian, after the loss corrector)
  double H0[36], g0[6];
  std::vector<double> residuals0;  // corrected residuals at the initial point
  std::vector<double> raw_residuals0;  // un-corrected residuals at the initial point
};
struct SolveOptions {
  int max_num_iterations = 50;
  double huber_a = 0.1;       // <= 0: no loss (trivial)
  bool quaternion_block0 = true;  // block0 = q[4] (x,y,z,w) with EigenQuaternionParameterization; else plain R^3
  // Ceres defaults (solver.h)
  double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  int max_num_consecutive_invalid_steps = 5;
  bool jacobi_scaling = true;
};

class Problem {
 public:
  std::vector<std::unique_ptr<CostFunction>> blocks;
  void Add(CostFunction* c) { blocks.emplace_back(c); }
  // p0: size 4 (quaternion) or 3; p1: size 3.  Updated in place like ceres::Solve.
  void Solve(const SolveOptions& opt, double* p0, double* p1, SolveSummary* summary);
};

// ceres::AngleAxisRotatePoint (ceres/rotation.h) — used by the VO functors.
template <class T>
inline void AngleAxisRotatePoint(const T angle_axis[3], const T pt[3], T result[3]) {
  using std::sqrt; using std::sin; using std::cos;
  const T theta2 = angle_axis[0] * angle_axis[0] + angle_axis[1] * angle_axis[1] + angle_axis[2] * angle_axis[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T costheta = cos(theta);
    const T sintheta = sin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {angle_axis[0] * theta_inverse, angle_axis[1] * theta_inverse, angle_axis[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    const T w_cross_pt[3] = {angle_axis[1] * pt[2] - angle_axis[2] * pt[1], angle_axis[2] * pt[0] - angle_axis[0] * pt[2],
                             angle_axis[0] * pt[1] - angle_axis[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

}  // namespace orc
