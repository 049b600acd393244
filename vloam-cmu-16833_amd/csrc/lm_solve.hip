// Levenberg–Marquardt on one workgroup: the whole ceres::Solve() of the reference
// (laser_odometry.cpp:457-463, laser_mapping.cpp:609-617, visual_odometry.cpp:423) runs inside ONE
// kernel launch.  16 wavefronts evaluate the residual blocks (closed-form Jacobians in the tangent
// space of EigenQuaternionParameterization, Huber corrector), reduce the 6x6 J^T J / J^T r / cost
// with wavefront shuffles + a fixed-order LDS pass (bit-reproducible), and lane 0 runs the
// trust-region bookkeeping of Ceres 2.0 (Jacobi scaling, LM diagonal clamp, step acceptance,
// radius schedule, tolerances) on the 6x6 normal equations.  No host round trips, no atomics.
//
// Ceres is not vendored by the reference; the algorithm restated here is spelled out in
// oracle/orc_ceres.cpp (CPU oracle, DENSE_QR on the stacked Jacobian) and SURVEY.md Appendix A.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include "lm_solve.h"

namespace vloam {

constexpr int kLmThreads = 512;
constexpr int kAcc = 28;  // cost, g[6], H upper triangle[21]

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return d3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Eigen QuaternionBase::_transformVector: v + w * (2 u x v) + u x (2 u x v)
__device__ __forceinline__ D3 quat_rotate(const double* q, D3 v) {
  D3 u = d3(q[0], q[1], q[2]);
  D3 uv = cross(u, v);
  uv = uv + uv;
  D3 wuv = q[3] * uv;
  D3 c = cross(u, uv);
  return d3((v.x + wuv.x) + c.x, (v.y + wuv.y) + c.y, (v.z + wuv.z) + c.z);
}
__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* r) {  // Hamilton, (x,y,z,w)
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// x_plus = Plus(x, delta): EigenQuaternionParameterization on the first block, identity on the rest.
__device__ void lm_plus(const double* x, const double* delta, double* out, bool quat) {
  if (quat) {
    const double n = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (n > 0.0) {
      const double s = sin(n) / n;
      double dq[4] = {s * delta[0], s * delta[1], s * delta[2], cos(n)};
      quat_mul(dq, x, out);
    } else {
      for (int i = 0; i < 4; i++) out[i] = x[i];
    }
    for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + delta[3 + i];
  } else {
    for (int i = 0; i < 6; i++) out[i] = x[i] + delta[i];
  }
}

// Forward-mode dual number with 6 partials — used for the angle-axis (VO) functors, where it
// reproduces what ceres::AutoDiffCostFunction computes.
struct Dual6 {
  double a, v[6];
};
__device__ __forceinline__ Dual6 dconst(double a) { Dual6 r; r.a = a; for (int i = 0; i < 6; i++) r.v[i] = 0; return r; }
__device__ __forceinline__ Dual6 dvar(double a, int k) { Dual6 r = dconst(a); r.v[k] = 1.0; return r; }
__device__ __forceinline__ Dual6 operator+(Dual6 f, Dual6 g) { Dual6 r; r.a = f.a + g.a; for (int i = 0; i < 6; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
__device__ __forceinline__ Dual6 operator-(Dual6 f, Dual6 g) { Dual6 r; r.a = f.a - g.a; for (int i = 0; i < 6; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
__device__ __forceinline__ Dual6 operator*(Dual6 f, Dual6 g) { Dual6 r; r.a = f.a * g.a; for (int i = 0; i < 6; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
__device__ __forceinline__ Dual6 operator/(Dual6 f, Dual6 g) {
  Dual6 r; const double gi = 1.0 / g.a, fg = f.a * gi; r.a = fg; for (int i = 0; i < 6; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r;
}
__device__ __forceinline__ Dual6 dsqrt(Dual6 f) { Dual6 r; r.a = sqrt(f.a); const double s = 1.0 / (2.0 * r.a); for (int i = 0; i < 6; i++) r.v[i] = s * f.v[i]; return r; }
__device__ __forceinline__ Dual6 dsin(Dual6 f) { Dual6 r; r.a = sin(f.a); const double c = cos(f.a); for (int i = 0; i < 6; i++) r.v[i] = c * f.v[i]; return r; }
__device__ __forceinline__ Dual6 dcos(Dual6 f) { Dual6 r; r.a = cos(f.a); const double s = -sin(f.a); for (int i = 0; i < 6; i++) r.v[i] = s * f.v[i]; return r; }

// ceres::AngleAxisRotatePoint (ceres/rotation.h)
__device__ void angle_axis_rotate(const Dual6* w_, const Dual6* pt, Dual6* out) {
  const Dual6 theta2 = w_[0] * w_[0] + w_[1] * w_[1] + w_[2] * w_[2];
  if (theta2.a > DBL_EPSILON) {
    const Dual6 theta = dsqrt(theta2);
    const Dual6 costheta = dcos(theta), sintheta = dsin(theta);
    const Dual6 ti = dconst(1.0) / theta;
    const Dual6 w[3] = {w_[0] * ti, w_[1] * ti, w_[2] * ti};
    const Dual6 wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const Dual6 tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (dconst(1.0) - costheta);
    for (int k = 0; k < 3; k++) out[k] = pt[k] * costheta + wxp[k] * sintheta + w[k] * tmp;
  } else {
    const Dual6 wxp[3] = {w_[1] * pt[2] - w_[2] * pt[1], w_[2] * pt[0] - w_[0] * pt[2], w_[0] * pt[1] - w_[1] * pt[0]};
    for (int k = 0; k < 3; k++) out[k] = pt[k] + wxp[k];
  }
}

// One residual block: residuals r[nres] and tangent-space Jacobian J[nres][6].  Returns nres.
__device__ int lm_factor(int type, D3 p, D3 A, D3 B, const double* x, double r[3], double J[3][6]) {
  if (type >= 1 && type <= 3) {
    D3 rp = quat_rotate(x, p);                       // R(q) p
    D3 lp = rp + d3(x[4], x[5], x[6]);
    // d lp / d delta = -2 [R p]x  (EigenQuaternionParameterization: q+ = exp(delta) * q), d lp / d t = I
    if (type == 1) {                                 // LidarEdgeFactor, lidarFactor.hpp:21-45
      D3 nu = cross(lp - A, lp - B);
      D3 de = A - B;
      double dn = sqrt(dot(de, de));
      r[0] = nu.x / dn; r[1] = nu.y / dn; r[2] = nu.z / dn;
      D3 v = d3((B.x - A.x) / dn, (B.y - A.y) / dn, (B.z - A.z) / dn);   // d r / d lp = [v]x
      const double M[3][3] = {{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}};
      const double R2[3][3] = {{0, 2 * rp.z, -2 * rp.y}, {-2 * rp.z, 0, 2 * rp.x}, {2 * rp.y, -2 * rp.x, 0}};  // -2 [rp]x
      for (int k = 0; k < 3; k++) {
        for (int a = 0; a < 3; a++) {
          J[k][a] = M[k][0] * R2[0][a] + M[k][1] * R2[1][a] + M[k][2] * R2[2][a];
          J[k][3 + a] = M[k][a];
        }
      }
      return 3;
    }
    D3 n = (type == 2) ? B : A;
    if (type == 2) r[0] = dot(lp - A, B);            // LidarPlaneFactor, lidarFactor.hpp:72-93
    else r[0] = dot(A, lp) + B.x;                    // LidarPlaneNormFactor, lidarFactor.hpp:115-127
    D3 nr = cross(n, rp);                            // n^T (-2 [rp]x) = -2 (n x rp)^T
    J[0][0] = -2 * nr.x; J[0][1] = -2 * nr.y; J[0][2] = -2 * nr.z;
    J[0][3] = n.x; J[0][4] = n.y; J[0][5] = n.z;
    return 1;
  }
  // VO functors on (angle_axis[3], t[3]) — dual numbers == Ceres autodiff
  Dual6 w[3] = {dvar(x[0], 0), dvar(x[1], 1), dvar(x[2], 2)};
  Dual6 t[3] = {dvar(x[3], 3), dvar(x[4], 4), dvar(x[5], 5)};
  if (type == 4) {                                   // CostFunctor32, ceres_cost_function.h:68-85
    Dual6 X0[3] = {dconst(p.x), dconst(p.y), dconst(p.z)}, X1[3];
    angle_axis_rotate(w, X0, X1);
    for (int k = 0; k < 3; k++) X1[k] = X1[k] + t[k];
    Dual6 r0 = X1[0] - X1[2] * dconst(A.x);
    Dual6 r1 = X1[1] - X1[2] * dconst(A.y);
    r[0] = r0.a; r[1] = r1.a;
    for (int a = 0; a < 6; a++) { J[0][a] = r0.v[a]; J[1][a] = r1.v[a]; }
    return 2;
  }
  {                                                  // CostFunctor22, ceres_cost_function.h:159-174
    Dual6 X0[3] = {dconst(p.x), dconst(p.y), dconst(1.0)}, RX[3];
    angle_axis_rotate(w, X0, RX);
    Dual6 c[3] = {t[1] * RX[2] - t[2] * RX[1], t[2] * RX[0] - t[0] * RX[2], t[0] * RX[1] - t[1] * RX[0]};
    Dual6 r0 = dconst(A.x) * c[0] + dconst(A.y) * c[1] + c[2];
    r[0] = r0.a;
    for (int a = 0; a < 6; a++) J[0][a] = r0.v[a];
    return 1;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// Evaluate every factor at x: cost, g = J^T r, H = J^T J (upper triangle), after the Huber corrector.
// Result lands in s_out[kAcc] (LDS).  All kLmThreads threads must call.
__device__ void lm_evaluate(const FactorTable& F, int n_slots, const double* x, double huber_a, double* s_part /* [16][kAcc] */,
                            double* s_out, bool store_resid, int* s_count) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc[kAcc];
  for (int i = 0; i < kAcc; i++) acc[i] = 0.0;
  int cnt = 0;
  const int cap = F.cap;
  for (int k = tid; k < n_slots; k += kLmThreads) {
    const int type = F.type[k];
    if (type == 0) continue;
    cnt++;
    D3 p = d3(F.p[k], F.p[cap + k], F.p[2 * cap + k]);
    D3 A = d3(F.A[k], F.A[cap + k], F.A[2 * cap + k]);
    D3 B = d3(F.B[k], F.B[cap + k], F.B[2 * cap + k]);
    double r[3], J[3][6];
    const int nr = lm_factor(type, p, A, B, x, r, J);
    if (store_resid) for (int q = 0; q < 3; q++) F.resid[q * cap + k] = q < nr ? r[q] : 0.0;
    double sq = 0.0;
    for (int q = 0; q < nr; q++) sq += r[q] * r[q];
    // ceres::HuberLoss::Evaluate + Corrector (rho'' <= 0 -> plain sqrt(rho') scaling)
    double rho0 = sq, sc = 1.0;
    if (huber_a > 0.0 && sq > huber_a * huber_a) {
      const double rr = sqrt(sq);
      rho0 = 2.0 * huber_a * rr - huber_a * huber_a;
      sc = sqrt(fmax(DBL_MIN, huber_a / rr));
    }
    acc[0] += 0.5 * rho0;
    for (int q = 0; q < nr; q++) {
      const double rq = r[q] * sc;
      double Jq[6];
      for (int a = 0; a < 6; a++) Jq[a] = J[q][a] * sc;
      int h = 7;
      for (int a = 0; a < 6; a++) {
        acc[1 + a] += Jq[a] * rq;
        for (int b = a; b < 6; b++) acc[h++] += Jq[a] * Jq[b];
      }
    }
  }
  for (int i = 0; i < kAcc; i++) {
    double v = wave_sum(acc[i]);
    if (lane == 0) s_part[wave * kAcc + i] = v;
  }
  cnt += __shfl_xor(cnt, 32); cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 8);
  cnt += __shfl_xor(cnt, 4); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 1);
  if (lane == 0) ((int*)(s_part + 16 * kAcc))[wave] = cnt;
  __syncthreads();
  if (tid < kAcc) {
    double s = 0.0;
    for (int w = 0; w < kLmThreads / 64; w++) s += s_part[w * kAcc + tid];  // fixed order: reproducible
    s_out[tid] = s;
  }
  if (tid == 0) {
    int c = 0;
    for (int w = 0; w < kLmThreads / 64; w++) c += ((int*)(s_part + 16 * kAcc))[w];
    *s_count = c;
  }
  __syncthreads();
}

__device__ __forceinline__ double Hget(const double* acc, int a, int b) {  // symmetric accessor into the packed triangle
  if (a > b) { int t = a; a = b; b = t; }
  // offset of row a in the packed upper triangle: sum_{i<a} (6 - i)
  const int off = a * 6 - (a * (a - 1)) / 2;
  return acc[7 + off + (b - a)];
}

// 6x6 SPD solve by Cholesky; returns false on a non-positive pivot.
__device__ bool chol6_solve(double M[6][6], const double* rhs, double* y) {
  double L[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = M[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double z[6];
  for (int i = 0; i < 6; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i][k] * z[k]; z[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; i--) { double s = z[i]; for (int k = i + 1; k < 6; k++) s -= L[k][i] * y[k]; y[i] = s / L[i][i]; }
  return true;
}

struct LmShared {
  double part[16 * kAcc + 16];
  double cur[kAcc];    // accumulators at x
  double cand[kAcc];   // accumulators at the candidate
  double x[7], xc[7];
  int go;              // 1: evaluate candidate next, 0: finished
  int count;
};

__global__ __launch_bounds__(kLmThreads) void k_lm_solve(FactorTable F, const int* n_slots_ptr, int n_slots_fixed, double* x_io,
                                                         LMRecord* rec, int max_iters, double huber_a, int quat, const int* enable_flag) {
  __shared__ LmShared sh;
  const int tid = threadIdx.x;
  if (enable_flag && *enable_flag == 0) return;
  const int n_slots = n_slots_ptr ? min(*n_slots_ptr, F.cap) : n_slots_fixed;
  const int na = quat ? 7 : 6;
  if (tid < na) sh.x[tid] = x_io[tid];
  __syncthreads();

  lm_evaluate(F, n_slots, sh.x, huber_a, sh.part, sh.cur, true, &sh.count);

  // ---- trust-region state (registers of thread 0; other threads only follow sh.go)
  double scale[6], diagonal[6];
  double radius = 1e4, decrease_factor = 2.0, minimum_cost = DBL_MAX, current_cost = 0, x_cost = 0, x_norm = 0, gmax = 0;
  double best[7];
  bool reuse_diagonal = false;
  int num_invalid = 0, iteration = 0, n_rec = 0, termination = 0, n_evals = 1;
  double it_cost = 0, it_cost_change = 0, it_step_norm = 0, it_rho = 0;
  bool it_valid = true, it_success = true;

  auto grad_max = [&](const double* xx, const double* acc) {
    double ng[6], pg[7];
    for (int a = 0; a < 6; a++) ng[a] = -acc[1 + a];
    lm_plus(xx, ng, pg, quat);
    double m = 0;
    for (int i = 0; i < na; i++) m = fmax(m, fabs(xx[i] - pg[i]));
    return m;
  };

  if (tid == 0) {
    x_cost = sh.cur[0];
    for (int a = 0; a < 6; a++) scale[a] = 1.0 / (1.0 + sqrt(Hget(sh.cur, a, a)));  // jacobi_scaling, fixed at iteration 0
    gmax = grad_max(sh.x, sh.cur);
    { double s = 0; for (int i = 0; i < na; i++) s += sh.x[i] * sh.x[i]; x_norm = sqrt(s); }
    current_cost = x_cost;
    it_cost = x_cost;
    for (int i = 0; i < 7; i++) { best[i] = i < na ? sh.x[i] : 0; rec->x_in[i] = best[i]; }
    for (int a = 0; a < 6; a++) { rec->g0[a] = sh.cur[1 + a]; for (int b = 0; b < 6; b++) rec->H0[a * 6 + b] = Hget(sh.cur, a, b); }
    rec->initial_cost = x_cost;
    rec->n_factors = sh.count;
  }

  for (;;) {
    if (tid == 0) {
      int go = -1;  // -1: keep looping inside thread 0 (invalid step), 0: stop, 1: evaluate candidate
      while (go < 0) {
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
        if (it_success && x_cost < minimum_cost) { minimum_cost = x_cost; for (int i = 0; i < na; i++) best[i] = sh.x[i]; }
        if (n_rec < kLmMaxTrace) {
          double* row = rec->trace[n_rec];
          row[0] = it_cost; row[1] = it_cost_change; row[2] = gmax; row[3] = it_step_norm; row[4] = it_rho; row[5] = radius;
          row[6] = it_valid; row[7] = it_success;
        }
        n_rec++;
        if (iteration >= max_iters) { termination = 0; go = 0; break; }
        if (gmax <= 1e-10) { termination = 1; go = 0; break; }
        if (radius <= 1e-32) { termination = 1; go = 0; break; }
        iteration++;
        // ---- LevenbergMarquardtStrategy::ComputeStep on the normal equations of the scaled Jacobian
        double Hs[6][6], gs[6];
        for (int a = 0; a < 6; a++) { gs[a] = sh.cur[1 + a] * scale[a]; for (int b = 0; b < 6; b++) Hs[a][b] = Hget(sh.cur, a, b) * scale[a] * scale[b]; }
        if (!reuse_diagonal) for (int a = 0; a < 6; a++) diagonal[a] = fmin(fmax(Hs[a][a], 1e-6), 1e32);
        double M[6][6];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) M[a][b] = Hs[a][b] + (a == b ? diagonal[a] / radius : 0.0);
        double y[6], step[6];
        bool ok = chol6_solve(M, gs, y);
        for (int a = 0; a < 6; a++) if (!isfinite(y[a])) ok = false;
        reuse_diagonal = true;
        double model_cost_change = 0;
        it_valid = false;
        if (ok) {
          double sg = 0, sHs = 0;
          for (int a = 0; a < 6; a++) step[a] = -y[a];
          for (int a = 0; a < 6; a++) { sg += step[a] * gs[a]; double t = 0; for (int b = 0; b < 6; b++) t += Hs[a][b] * step[b]; sHs += step[a] * t; }
          model_cost_change = -sg - 0.5 * sHs;  // == -(Js s)^T (r + Js s / 2)
          it_valid = model_cost_change > 0.0;
        }
        if (!it_valid) {
          // ---- HandleInvalidStep
          if (++num_invalid >= 5) { termination = 2; go = 0; break; }
          radius *= 0.5;
          it_cost = x_cost; it_cost_change = 0; it_step_norm = 0; it_rho = 0; it_success = false;
          continue;
        }
        num_invalid = 0;
        double delta[6];
        for (int a = 0; a < 6; a++) delta[a] = step[a] * scale[a];
        lm_plus(sh.x, delta, sh.xc, quat);
        sh.cand[0] = model_cost_change;  // parked; overwritten by the evaluation after being read back below
        go = 1;
      }
      sh.go = go;
    }
    __syncthreads();
    if (sh.go == 0) break;
    double model_cost_change = sh.cand[0];  // (only thread 0 uses it)
    __syncthreads();
    lm_evaluate(F, n_slots, sh.xc, huber_a, sh.part, sh.cand, false, &sh.count);
    if (tid == 0) {
      n_evals++;
      double candidate_cost = sh.cand[0];
      if (!isfinite(candidate_cost)) candidate_cost = DBL_MAX;
      bool stop = false;
      { double s = 0; for (int i = 0; i < na; i++) s += (sh.x[i] - sh.xc[i]) * (sh.x[i] - sh.xc[i]); it_step_norm = sqrt(s); }
      if (it_step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; stop = true; }  // ParameterToleranceReached
      if (!stop) {
        it_cost_change = x_cost - candidate_cost;
        if (fabs(it_cost_change) <= 1e-6 * x_cost) { termination = 1; stop = true; }  // FunctionToleranceReached
      }
      if (!stop) {
        it_rho = candidate_cost >= DBL_MAX ? -DBL_MAX : (current_cost - candidate_cost) / model_cost_change;
        if (it_rho > 1e-3) {  // HandleSuccessfulStep
          for (int i = 0; i < na; i++) sh.x[i] = sh.xc[i];
          for (int i = 0; i < kAcc; i++) sh.cur[i] = sh.cand[i];
          { double s = 0; for (int i = 0; i < na; i++) s += sh.x[i] * sh.x[i]; x_norm = sqrt(s); }
          x_cost = candidate_cost;
          gmax = grad_max(sh.x, sh.cur);
          it_cost = x_cost; it_success = true;
          radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * it_rho - 1.0, 3.0));
          radius = fmin(1e16, radius);
          decrease_factor = 2.0;
          reuse_diagonal = false;
          current_cost = candidate_cost;
        } else {              // HandleUnsuccessfulStep
          it_success = false;
          radius = radius / decrease_factor;
          decrease_factor *= 2.0;
          reuse_diagonal = true;
          it_cost = candidate_cost;
        }
      }
      sh.go = stop ? 0 : 1;
    }
    __syncthreads();
    if (sh.go == 0) break;
    __syncthreads();
  }

  if (tid == 0) {
    for (int i = 0; i < na; i++) x_io[i] = best[i];
    for (int i = 0; i < 7; i++) rec->x_out[i] = i < na ? best[i] : 0;
    rec->final_cost = minimum_cost;
    rec->n_iterations = n_rec < kLmMaxTrace ? n_rec : kLmMaxTrace;
    rec->termination = termination;
    rec->n_evals = n_evals;
  }
}

void lm_launch(hipStream_t st, const FactorTable& F, const int* d_n_slots, int n_slots_fixed, double* d_x, LMRecord* d_rec, int max_iters,
               double huber_a, bool quat, const int* d_enable, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLmSolve, st, k_lm_solve, dim3(1), dim3(kLmThreads), 0, st, F, d_n_slots, n_slots_fixed, d_x, d_rec, max_iters, huber_a, quat ? 1 : 0,
                     d_enable);
}

}  // namespace vloam
