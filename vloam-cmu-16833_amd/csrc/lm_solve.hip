// Levenberg–Marquardt in one launch: the whole ceres::Solve() of the reference
// (laser_odometry.cpp:457-463, laser_mapping.cpp:609-617, visual_odometry.cpp:423) runs inside ONE kernel launch.
//   factors    the kernels that emit a LiDAR factor (k_lo_assoc*, k_map_fit) leave, next to the raw points, the form the evaluation consumes
//              (FactorTable::dg: edge -> orthonormal pair across the line + offsets, plane -> normal + offset).  Scan-to-scan (kLmDirect): lane t
//              owns table slots t + 256 m themselves; scan-to-map (kLmRowMask): the solve compacts on its own from the 64-bit accepted-slot masks
//              k_map_fit leaves per 64-slot row; VO (kLmPacked): k_lm_compact packs the accepted factors in slot order first.
//   evaluation factors stay in the registers of 256 lanes per workgroup across the evaluations of a solve; residual blocks with closed-form Jacobians
//              in the tangent space of EigenQuaternionParameterization, Huber as the weight rho' (one rsqrt per block); the 6x6 J^T J / J^T r / cost
//              reduced through a fixed-order LDS transpose (bit-reproducible).
//   workgroups a single sequence runs both LiDAR problems as EIGHT cooperating workgroups on ONE XCD (lm_coop_block), a batch as 4 / 6 spread ones;
//              there is NO grid barrier: the partial sums travel as tagged 8-byte granules {32 payload bits, (launch generation, evaluation)} that
//              every workgroup polls, each adds them in workgroup order and runs the (cheap) trust-region bookkeeping redundantly.  A workgroup
//              whose partners never show up gives up after spin_limit polls and the lead workgroup redoes the solve alone (degrade, not fail).
//   step       lane 0 runs Ceres 2.0's trust-region bookkeeping (Jacobi scaling, LM diagonal clamp, step acceptance, radius schedule,
//              tolerances) on the 6x6 normal equations; helper lanes on other wavefronts prepare what can be prepared off its dependency chain.
// No host round trips, no float atomics.  Ceres is not vendored by the reference; the algorithm restated here is spelled out in
// oracle/orc_ceres.cpp (CPU oracle, DENSE_QR on the stacked Jacobian), tests/ceres_transcription.py and SURVEY.md Appendix A.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <float.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include "lm_solve.h"

namespace vloam {

constexpr int kLmThreads = 256;

constexpr int kAcc = 28;  // cost, g[6], H upper triangle[21]
typedef unsigned long long u64;

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


// Cooperative solves of a single sequence are launched 8 x wide and every eighth workgroup works: the dispatcher deals workgroups to the
// eight XCDs round-robin, so the NB working ones share ONE XCD and an exchange of partial sums costs 1 350 / 1 600 cycles (NB = 4 / 6)
// instead of 2 200 / 3 100 across XCDs (tools/microbench/solver_limits.hip, profiles/r04_solver_limits.txt) — same arithmetic, same
// results.  The scan-to-scan and the scan-to-map solves, which overlap in time, sit on different XCDs (2 and 6).  Returns the index of this
// workgroup among the working ones, -1: nothing to do.  (Round 3 tried this for batches too: there it crowds all solves of a session
// into one XCD's 32 compute units and loses; a batch keeps the plain launch, where session s already owns XCD s % 8.)
template <int NB, int MODE>
__device__ __forceinline__ int lm_coop_block() {
  const int b = (int)blockIdx.x;
  if (NB > 1 && (int)gridDim.x == NB * 8) { const int xcd = MODE == 1 ? 2 : 6; return (b & 7) != xcd ? -1 : (b >> 3); }
  return b;
}
// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier).  __syncthreads() also waits for every global store
// the wavefront has in flight — thread 0's trace rows, the residual hook of the first evaluation, the tagged granules — a memory round
// trip in front of barriers that only hand LDS data over.  Used where nothing but LDS crosses the barrier.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// acc += (cost, J^T r, upper triangle of J^T J) of one residual row; everything statically indexed (registers)
__device__ __forceinline__ void accumulate_row(double (&acc)[kAcc], const double (&J)[6], double r) {
#pragma unroll
  for (int a = 0; a < 6; a++) acc[1 + a] += J[a] * r;
  int h = 7;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = a; b < 6; b++) acc[h++] += J[a] * J[b];
}

// ceres::HuberLoss::Evaluate + Corrector: rho'' <= 0 for Huber, so residual and Jacobian rows are scaled by sqrt(rho').
// Branching form (VO path).
__device__ __forceinline__ double huber(double sq, double a, double* cost) {
  if (a > 0.0 && sq > a * a) {
    const double rr = sqrt(sq);
    *cost += 0.5 * (2.0 * a * rr - a * a);
    return sqrt(fmax(DBL_MIN, a / rr));
  }
  *cost += 0.5 * sq;
  return 1.0;
}
// Branch-free form given |r| (straight-line code lets the compiler interleave several factors per lane, which is what
// hides the ~12-cycle dependent f64 latency when only one wavefront sits on each SIMD).
__device__ __forceinline__ double huber_sel(double sq, double rr, double a, double sqrt_a, double* cost) {
  const bool out = sq > a * a;
  const double sc_out = sqrt_a * rsqrt(fmax(rr, DBL_MIN));  // sqrt(a / |r|)
  *cost += out ? 0.5 * (2.0 * a * rr - a * a) : 0.5 * sq;
  return out ? sc_out : 1.0;
}

// LidarEdgeFactor (lidarFactor.hpp:21-45) with lp = R p + t:  r = ((lp - a) x (lp - b)) / |a - b| == v x (lp - a),
// v = (b - a) / |a - b|.  r lies in the plane normal to v: with an orthonormal pair (e1, e2), e1 x e2 = v (k_lm_compact),
// r = c1 e2 - c2 e1 where c_i = e_i . lp + d_i, d_i = -(e_i . a).  (c1, c2) is r in rotated coordinates, and J^T J, J^T r,
// |r|^2 (hence the Huber weight) are invariant under an orthonormal change of residual coordinates — so the factor is
// evaluated as TWO point-to-plane rows sharing one weight instead of three generic rows:  d c_i / d lp = e_i^T,
// d lp / d delta = -2 [R p]x (EigenQuaternionParameterization, q+ = exp(delta) * q),  e^T (-2 [rp]x) = -2 (e x rp)^T.
__device__ __forceinline__ void eval_edge(D3 p, D3 e1, D3 e2, double d1, double d2, const double (&Rm)[9], D3 t, double huber_a, double sqrt_a,
                                          double (&acc)[kAcc], double* r3, bool want_r3) {
  const D3 rp = d3(Rm[0] * p.x + Rm[1] * p.y + Rm[2] * p.z, Rm[3] * p.x + Rm[4] * p.y + Rm[5] * p.z, Rm[6] * p.x + Rm[7] * p.y + Rm[8] * p.z);
  const D3 lp = rp + t;
  const double c1 = dot(e1, lp) + d1, c2 = dot(e2, lp) + d2;
  if (want_r3) { r3[0] = c1 * e2.x - c2 * e1.x; r3[1] = c1 * e2.y - c2 * e1.y; r3[2] = c1 * e2.z - c2 * e1.z; }
  const double sq = c1 * c1 + c2 * c2;
  const double sc = huber_sel(sq, sqrt(sq), huber_a, sqrt_a, &acc[0]);
  const double s2 = -2.0 * sc;
  {
    const D3 nr = cross(e1, rp);
    const double J[6] = {s2 * nr.x, s2 * nr.y, s2 * nr.z, e1.x * sc, e1.y * sc, e1.z * sc};
    accumulate_row(acc, J, c1 * sc);
  }
  {
    const D3 nr = cross(e2, rp);
    const double J[6] = {s2 * nr.x, s2 * nr.y, s2 * nr.z, e2.x * sc, e2.y * sc, e2.z * sc};
    accumulate_row(acc, J, c2 * sc);
  }
}

// LidarPlaneFactor / LidarPlaneNormFactor (lidarFactor.hpp:72-93, 115-127) in the common form r = n . lp + d
// (k_lm_compact rewrites the plane factor's (lp - j) . n as n . lp - n . j);  d r / d lp = n^T,  n^T (-2 [rp]x) = -2 (n x rp)^T.
__device__ __forceinline__ void eval_plane(D3 p, D3 n, double d, const double (&Rm)[9], D3 t, double huber_a, double sqrt_a, double (&acc)[kAcc],
                                           double* r3) {
  const D3 rp = d3(Rm[0] * p.x + Rm[1] * p.y + Rm[2] * p.z, Rm[3] * p.x + Rm[4] * p.y + Rm[5] * p.z, Rm[6] * p.x + Rm[7] * p.y + Rm[8] * p.z);
  const D3 lp = rp + t;
  const double r0 = dot(n, lp) + d;
  r3[0] = r0; r3[1] = 0.0; r3[2] = 0.0;
  const double sc = huber_sel(r0 * r0, fabs(r0), huber_a, sqrt_a, &acc[0]);
  const D3 nr = cross(n, rp);
  const double s2 = -2.0 * sc;
  const double J[6] = {s2 * nr.x, s2 * nr.y, s2 * nr.z, n.x * sc, n.y * sc, n.z * sc};
  accumulate_row(acc, J, r0 * sc);
}

// ---- packets: NW factors evaluated side by side.  With one wavefront per SIMD nothing hides the ~8-cycle dependent f64
// latency except independent instructions next to each other, and the compiler keeps source order inside a basic block: every
// step below is written for all NW factors at once, so the per-factor dependency chains (rotate, residual, Huber weight,
// Jacobian row) interleave NW-fold.  The accumulation is 27 independent FMAs per row already.
template <int NW>
__device__ __forceinline__ void rotate_pk(const double (&p)[NW][3], const double (&Rm)[9], double (&rp)[NW][3]) {
#pragma unroll
  for (int u = 0; u < NW; u++) rp[u][0] = Rm[0] * p[u][0] + Rm[1] * p[u][1] + Rm[2] * p[u][2];
#pragma unroll
  for (int u = 0; u < NW; u++) rp[u][1] = Rm[3] * p[u][0] + Rm[4] * p[u][1] + Rm[5] * p[u][2];
#pragma unroll
  for (int u = 0; u < NW; u++) rp[u][2] = Rm[6] * p[u][0] + Rm[7] * p[u][1] + Rm[8] * p[u][2];
}
// One point-to-plane row per factor: c = n . (rp + t) + d, J = [-2 (n x rp), n], weighted by w = rho' of the row's residual block:
// acc += (w J^T c, w J^T J).  Ceres' Corrector (rho'' <= 0 for Huber) scales residual and Jacobian by sqrt(rho') and the normal equations
// then hold rho' J^T J: the same numbers with ONE dependent rsqrt behind the squared norm instead of sqrt -> rsqrt -> two scalings (the
// factor loops are a latency chain at one wavefront per SIMD: ~32 cycles per dependent f64 operation, profiles/r04_solver_limits.txt).
template <int NW>
__device__ __forceinline__ void row_pk(const double (&rp)[NW][3], const double (&n)[NW][3], const double (&c)[NW], const double (&w)[NW],
                                       double (&acc)[kAcc]) {
  double J[NW][6], Jw[NW][6];
#pragma unroll
  for (int u = 0; u < NW; u++) {
    J[u][0] = -2.0 * (n[u][1] * rp[u][2] - n[u][2] * rp[u][1]);
    J[u][1] = -2.0 * (n[u][2] * rp[u][0] - n[u][0] * rp[u][2]);
    J[u][2] = -2.0 * (n[u][0] * rp[u][1] - n[u][1] * rp[u][0]);
    J[u][3] = n[u][0]; J[u][4] = n[u][1]; J[u][5] = n[u][2];
  }
#pragma unroll
  for (int u = 0; u < NW; u++)
#pragma unroll
    for (int a = 0; a < 6; a++) Jw[u][a] = w[u] * J[u][a];
#pragma unroll
  for (int u = 0; u < NW; u++) {
#pragma unroll
    for (int a = 0; a < 6; a++) acc[1 + a] += Jw[u][a] * c[u];
    int h = 7;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) acc[h++] += Jw[u][a] * J[u][b];
  }
}
// ceres::HuberLoss(a) on the squared norm sq of a residual block -> w = rho' (1 inside, a / |r| outside) and the block's cost
// (rho / 2: sq / 2 inside, a |r| - a^2 / 2 outside), with |r| = sq * rsqrt(sq): one transcendental per block
template <int NW>
__device__ __forceinline__ void huber_pk(const double (&sq)[NW], double a, double (&w)[NW], double* cost) {
  double ir[NW];
#pragma unroll
  for (int u = 0; u < NW; u++) ir[u] = rsqrt(fmax(sq[u], DBL_MIN));  // 1 / |r|
#pragma unroll
  for (int u = 0; u < NW; u++) {
    const bool out = sq[u] > a * a;
    const double rr = sq[u] * ir[u];
    *cost += out ? 0.5 * (2.0 * a * rr - a * a) : 0.5 * sq[u];
    w[u] = out ? a * ir[u] : 1.0;
  }
}
template <int NW>
__device__ __forceinline__ void eval_plane_pk(const double (&p)[NW][3], const double (&n)[NW][3], const double (&d)[NW], const double (&Rm)[9], D3 t,
                                              double huber_a, double (&acc)[kAcc], double (&r0)[NW]) {
  double rp[NW][3], sq[NW], w[NW];
  rotate_pk<NW>(p, Rm, rp);
#pragma unroll
  for (int u = 0; u < NW; u++) r0[u] = n[u][0] * (rp[u][0] + t.x) + n[u][1] * (rp[u][1] + t.y) + n[u][2] * (rp[u][2] + t.z) + d[u];
#pragma unroll
  for (int u = 0; u < NW; u++) sq[u] = r0[u] * r0[u];
  huber_pk<NW>(sq, huber_a, w, &acc[0]);
  row_pk<NW>(rp, n, r0, w, acc);
}
// edge factors as two rows sharing one weight (see eval_edge)
template <int NW>
__device__ __forceinline__ void eval_edge_pk(const double (&p)[NW][3], const double (&e1)[NW][3], const double (&e2)[NW][3], const double (&d1)[NW],
                                             const double (&d2)[NW], const double (&Rm)[9], D3 t, double huber_a,
                                             double (&acc)[kAcc], double (&c1)[NW], double (&c2)[NW]) {
  double rp[NW][3], sq[NW], w[NW];
  rotate_pk<NW>(p, Rm, rp);
#pragma unroll
  for (int u = 0; u < NW; u++) c1[u] = e1[u][0] * (rp[u][0] + t.x) + e1[u][1] * (rp[u][1] + t.y) + e1[u][2] * (rp[u][2] + t.z) + d1[u];
#pragma unroll
  for (int u = 0; u < NW; u++) c2[u] = e2[u][0] * (rp[u][0] + t.x) + e2[u][1] * (rp[u][1] + t.y) + e2[u][2] * (rp[u][2] + t.z) + d2[u];
#pragma unroll
  for (int u = 0; u < NW; u++) sq[u] = c1[u] * c1[u] + c2[u] * c2[u];
  huber_pk<NW>(sq, huber_a, w, &acc[0]);
  row_pk<NW>(rp, e1, c1, w, acc);
  row_pk<NW>(rp, e2, c2, w, acc);
}
// NE edge factors and NP plane factors SIDE BY SIDE, step by step (the compiler keeps source order inside a basic block): a lane of a
// cooperative solve typically owns one edge and one or two plane factors, and evaluated one packet after the other their dependency chains
// (rotate -> residual -> rsqrt -> weight -> rows) simply add up.
template <int NE, int NP>
__device__ __forceinline__ void eval_mixed_pk(const double (&pe)[NE][3], const double (&e1)[NE][3], const double (&e2)[NE][3], const double (&d1)[NE],
                                              const double (&d2)[NE], const double (&pp)[NP][3], const double (&n)[NP][3], const double (&d)[NP],
                                              const double (&Rm)[9], D3 t, double huber_a, double (&acc)[kAcc], double (&c1)[NE], double (&c2)[NE],
                                              double (&r0)[NP]) {
  double rpe[NE][3], rpp[NP][3], sq[NE + NP], w[NE + NP];
  rotate_pk<NE>(pe, Rm, rpe);
  rotate_pk<NP>(pp, Rm, rpp);
#pragma unroll
  for (int u = 0; u < NE; u++) c1[u] = e1[u][0] * (rpe[u][0] + t.x) + e1[u][1] * (rpe[u][1] + t.y) + e1[u][2] * (rpe[u][2] + t.z) + d1[u];
#pragma unroll
  for (int u = 0; u < NE; u++) c2[u] = e2[u][0] * (rpe[u][0] + t.x) + e2[u][1] * (rpe[u][1] + t.y) + e2[u][2] * (rpe[u][2] + t.z) + d2[u];
#pragma unroll
  for (int u = 0; u < NP; u++) r0[u] = n[u][0] * (rpp[u][0] + t.x) + n[u][1] * (rpp[u][1] + t.y) + n[u][2] * (rpp[u][2] + t.z) + d[u];
#pragma unroll
  for (int u = 0; u < NE; u++) sq[u] = c1[u] * c1[u] + c2[u] * c2[u];
#pragma unroll
  for (int u = 0; u < NP; u++) sq[NE + u] = r0[u] * r0[u];
  huber_pk<NE + NP>(sq, huber_a, w, &acc[0]);
  double we[NE], wp[NP];
#pragma unroll
  for (int u = 0; u < NE; u++) we[u] = w[u];
#pragma unroll
  for (int u = 0; u < NP; u++) wp[u] = w[NE + u];
  row_pk<NE>(rpe, e1, c1, we, acc);
  row_pk<NE>(rpe, e2, c2, we, acc);
  row_pk<NP>(rpp, n, r0, wp, acc);
}

// CostFunctor32 / CostFunctor22 on (angle_axis[3], t[3]) — dual numbers == Ceres autodiff
__device__ __forceinline__ void eval_vo(int type, D3 p, D3 A, const double* x, double huber_a, double (&acc)[kAcc], double* r3) {
  Dual6 w[3] = {dvar(x[0], 0), dvar(x[1], 1), dvar(x[2], 2)};
  Dual6 t[3] = {dvar(x[3], 3), dvar(x[4], 4), dvar(x[5], 5)};
  if (type == 4) {  // ceres_cost_function.h:68-85
    Dual6 X0[3] = {dconst(p.x), dconst(p.y), dconst(p.z)}, X1[3];
    angle_axis_rotate(w, X0, X1);
#pragma unroll
    for (int k = 0; k < 3; k++) X1[k] = X1[k] + t[k];
    const Dual6 q0 = X1[0] - X1[2] * dconst(A.x);
    const Dual6 q1 = X1[1] - X1[2] * dconst(A.y);
    r3[0] = q0.a; r3[1] = q1.a; r3[2] = 0.0;
    const double sc = huber(q0.a * q0.a + q1.a * q1.a, huber_a, &acc[0]);
    double J[6];
#pragma unroll
    for (int a = 0; a < 6; a++) J[a] = q0.v[a] * sc;
    accumulate_row(acc, J, q0.a * sc);
#pragma unroll
    for (int a = 0; a < 6; a++) J[a] = q1.v[a] * sc;
    accumulate_row(acc, J, q1.a * sc);
  } else {          // ceres_cost_function.h:159-174
    Dual6 X0[3] = {dconst(p.x), dconst(p.y), dconst(1.0)}, RX[3];
    angle_axis_rotate(w, X0, RX);
    const Dual6 c0 = t[1] * RX[2] - t[2] * RX[1], c1 = t[2] * RX[0] - t[0] * RX[2], c2 = t[0] * RX[1] - t[1] * RX[0];
    const Dual6 q0 = dconst(A.x) * c0 + dconst(A.y) * c1 + c2;
    r3[0] = q0.a; r3[1] = 0.0; r3[2] = 0.0;
    const double sc = huber(q0.a * q0.a, huber_a, &acc[0]);
    double J[6];
#pragma unroll
    for (int a = 0; a < 6; a++) J[a] = q0.v[a] * sc;
    accumulate_row(acc, J, q0.a * sc);
  }
}

// row stride of the transpose buffer: == 8 (mod 32) doubles, so that the 8 rows a wavefront reads in the column pass land in
// different 64-bit banks (a 256-double stride would put all of them in the same 8 banks)
constexpr int kRedStride = kLmThreads + 8;
struct LmShared {
  double red[kAcc * kRedStride];  // [value][thread] transpose buffer of the block reduction (rows padded against bank conflicts)
  double part[8 * kAcc];          // [sub-sum][value]
  double acc2[2][kAcc];  // accumulators at x (acc2[curidx]) and at the candidate (acc2[curidx ^ 1]): accepting a step flips the index
  double Hs[2][21], gs[2][6], diag[2][6];  // Jacobi-scaled normal equations + clamped LM diagonal, one set per accumulator buffer (acc2): the candidate's set is prepared while thread 0 decides whether to accept it
  int curidx;
  double x[8], xc[8];
  double mcc;          // model_cost_change of the pending candidate
  double inv_mcc, step_norm_c;   // 1 / mcc and |x - xc| of the pending candidate, by a helper lane while the candidate is evaluated
  double gmax_c, xnorm_c;  // gradient max-norm / |x| at the point just evaluated (computed by helper lanes in parallel)
  double scale[6], best[8];  // trust-region state that must survive the evaluations (kept out of registers)
  int scan[kLmThreads], scan2[kLmThreads];
  int n_edge;
  int go, go2;         // 1: evaluate candidate next / go on to the next step, 0: finished (one word per hand-over: the loop has no barrier between the acceptance and the next step)
  int n_valid;
  int failed;          // cooperative solve: a workgroup gave up waiting at the grid barrier -> every workgroup abandons the solve
  double x0[8];        // the parameters the solve started from (what an abandoned solve hands back)
#ifdef VLOAM_LM_STAMPS
  long long step_cyc[8];  // debug build: thread 0's cycles in the sections of the trust-region step, summed over the iterations
  long long ev_cyc[8];  // debug build: thread 0's cycles in the phases of lm_evaluate, summed over the evaluations after the first
#endif
};

// Factors owned by a lane stay in its registers across the evaluations of a solve (one workgroup = 256 lanes x 512 VGPRs):
// edge factor e = lane + 256 m (m < kCacheE) and plane factor q = lane + 256 m (m < kCacheP) are fetched once — the first
// evaluation issues all of those loads together — and later evaluations touch no memory at all.  Unused slots hold zeros,
// which evaluate to an exactly-zero contribution, so the evaluation itself is straight-line code: with a single wavefront
// per SIMD the only way to hide the dependent f64 latency is to let independent factors interleave in one basic block.
// p is an f32 point by construction (cloud coordinates) and is kept as f32.
constexpr int kCacheE = 3, kCacheP = 6;
struct LmCache {
  float pe[kCacheE][3];
  double de[kCacheE][8];  // e1, e2, d1, d2
  float pp[kCacheP][3];
  double dp[kCacheP][4];  // n, d
  unsigned live_e, live_p;  // direct mode: which of this lane's slots hold a factor
  int slot_e[kCacheE], slot_p[kCacheP];  // row-mask mode: the table slot of every cached factor (residual hook of the first evaluation)
};

// ---- factor sources of a solve
//   kLmPacked  the factors were compacted into cpack by k_lm_compact (the visual-odometry problem)
//   kLmDirect  scan-to-scan odometry: lane t owns the table slots t + 256 m themselves (see lm_evaluate)
//   kLmRowMask scan-to-map: the fit kernel left one 64-bit mask of accepted slots per 64-slot row; the solve compacts ON ITS OWN — a
//              workgroup-wide scan of the row counts gives every row its offset, compact index k -> (row by binary search over the
//              offsets, bit = k-th accepted slot of the row) -> slot, and every lane fetches ITS factors straight from the raw table
//              and digests them (edge -> orthonormal pair, plane -> normal + offset) into its register cache.  Same lane ownership and
//              the same summation order as the packed form, without the k_lm_compact launch in front of every solve
//              (5.9 us + a launch boundary, twice per sweep on the stream that bounds the sweep period).
constexpr int kLmPacked = 0, kLmDirect = 1, kLmRowMask = 2;

__device__ __forceinline__ int nth_set_bit(u64 m, int j) {   // position of the j-th (0-based) set bit of m; j < popcount(m)
  unsigned w = (unsigned)m;
  int pos = 0;
  int c = __popc(w);
  if (j >= c) { j -= c; pos = 32; w = (unsigned)(m >> 32); }
#pragma unroll
  for (int width = 16; width > 0; width >>= 1) {
    c = __popc(w & ((1u << width) - 1u));
    if (j >= c) { j -= c; pos += width; w >>= width; }
  }
  return pos;
}
// compact index k (< total) -> table slot, from the row offsets / row masks staged in LDS
__device__ __forceinline__ int lm_slot_of(const int* rowoff, const u64* rowmask, int nrows, int k) {
  int r = 0;   // the largest row index with rowoff[r] <= k (empty rows share their successor's offset and are skipped by "largest")
#pragma unroll
  for (int step = 256; step > 0; step >>= 1) {
    const int t = r + step;
    if (t <= nrows && rowoff[t] <= k) r = t;
  }
  return (r << 6) + nth_set_bit(rowmask[r], k - rowoff[r]);
}
// raw table slot -> what the evaluation consumes (the digest k_lm_compact writes for the packed form)
__device__ __forceinline__ void lm_digest_edge(const FactorTable& F, int slot, double (&p)[3], double (&fr)[8]) {
  const int cap = F.cap;
  double a[3], b[3];
#pragma unroll
  for (int q = 0; q < 3; q++) { p[q] = F.p[q * cap + slot]; a[q] = F.A[q * cap + slot]; b[q] = F.B[q * cap + slot]; }
  edge_frame(a[0], a[1], a[2], b[0], b[1], b[2], fr);
}
__device__ __forceinline__ void lm_digest_plane(const FactorTable& F, int slot, double (&p)[3], double (&nd)[4]) {
  const int cap = F.cap;
  const int ty = F.type[slot];
  double a[3], b[3];
#pragma unroll
  for (int q = 0; q < 3; q++) { p[q] = F.p[q * cap + slot]; a[q] = F.A[q * cap + slot]; b[q] = F.B[q * cap + slot]; }
  if (ty == 2) {  // LidarPlaneFactor (lp - j) . n  ->  n . lp + d with d = -(n . j)
    nd[0] = b[0]; nd[1] = b[1]; nd[2] = b[2]; nd[3] = -(b[0] * a[0] + b[1] * a[1] + b[2] * a[2]);
  } else {        // LidarPlaneNormFactor: n . lp + negative_OA_dot_norm
    nd[0] = a[0]; nd[1] = a[1]; nd[2] = a[2]; nd[3] = b[0];
  }
}

// Evaluate the compacted factors at x: cost, g = J^T r, H = J^T J (upper triangle) -> s_out[kAcc] (LDS).
// DIRECT (scan-to-scan odometry, table of exactly 256 x (kCacheE + kCacheP) slots): lane t owns the table slots t + 256 m
// themselves — corner slots feed its edge cache, plane slots its plane cache — and builds the solver's form of each factor on
// the fly, so no compaction pass runs at all; empty slots hold zeros and whole-wavefront-empty packets are skipped.
//
// NB > 1 (scan-to-map problems, thousands of factors): NB workgroups on NB compute units share the factors — workgroup b, lane t
// is lane b * 256 + t of an NB * 256 wide virtual workgroup — reduce their own share, publish the kAcc partial sums, meet at a
// grid barrier and ALL add the partials in workgroup order, so every workgroup holds bit-identical accumulators and runs the
// (cheap) trust-region bookkeeping redundantly: one barrier per evaluation, nothing to broadcast.  `eval_idx` counts the
// evaluations of this solve (barrier target and double-buffer parity of the partial sums).
template <bool QUAT, int MODE, int NB>
__device__ __forceinline__ void lm_evaluate(const FactorTable& F, int n_edge, int n_valid, const double* x, double huber_a, LmShared& sh,
                                            double* s_out, bool first, LmCache& C, long long* cyc_factors, int eval_idx, unsigned tag_base) {
  static_assert(NB == 1 || QUAT, "the cooperative form exists for the quaternion problems");
  constexpr bool DIRECT = MODE == kLmDirect;
  const int tid = threadIdx.x;
  const int blk = NB > 1 ? lm_coop_block<NB, MODE>() : 0;
  const int vt = blk * kLmThreads + tid;     // lane of the virtual workgroup
  constexpr int VT = NB * kLmThreads;
  const long long tf0 = clock64();
  double acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; i++) acc[i] = 0.0;
  const int cap = F.cap;
  double xl[7];
#pragma unroll
  for (int i = 0; i < 7; i++) xl[i] = x[i];
  if (QUAT) {
    // rotation matrix of q = (x, y, z, w) as Eigen's toRotationMatrix() builds it (q is unit up to rounding)
    double Rm[9];
    {
      const double tx = 2 * xl[0], ty = 2 * xl[1], tz = 2 * xl[2];
      const double twx = tx * xl[3], twy = ty * xl[3], twz = tz * xl[3];
      const double txx = tx * xl[0], txy = ty * xl[0], txz = tz * xl[0], tyy = ty * xl[1], tyz = tz * xl[1], tzz = tz * xl[2];
      Rm[0] = 1 - (tyy + tzz); Rm[1] = txy - twz; Rm[2] = txz + twy;
      Rm[3] = txy + twz; Rm[4] = 1 - (txx + tzz); Rm[5] = tyz - twx;
      Rm[6] = txz - twy; Rm[7] = tyz + twx; Rm[8] = 1 - (txx + tyy);
    }
    const D3 tt = d3(xl[4], xl[5], xl[6]);
    const int n_plane = n_valid - n_edge;
    const double* cp = F.cpack;
    if (first && DIRECT) {
      // the producer (k_lo_assoc*) left the evaluation's form of every factor next to the raw one (FactorTable::dg, by slot): type, point and
      // digest are requested together — one memory trip — and dead slots are zeroed by selects (zeros evaluate to an exactly-zero contribution)
      C.live_e = 0u; C.live_p = 0u;
      constexpr int kEdgeSlots = kCacheE * kLmThreads;  // the table's corner part (the plane part follows)
      const double* dg = F.dg;
      int ty_e[kCacheE], ty_p[kCacheP];
      double pe[kCacheE][3], de[kCacheE][8], pp[kCacheP][3], dp[kCacheP][4];
#pragma unroll
      for (int m = 0; m < kCacheE; m++) {
        ty_e[m] = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) pe[m][q] = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) de[m][q] = 0.0;
        if (m * VT < kEdgeSlots) {   // compile-time: with NB workgroups only the first cache slots can hold a table slot at all (no loads for the others)
          const int slot_raw = vt + m * VT;
          const int slot = slot_raw < kEdgeSlots ? slot_raw : 0;
          ty_e[m] = slot_raw < kEdgeSlots ? F.type[slot] : 0;
#pragma unroll
          for (int q = 0; q < 3; q++) pe[m][q] = F.p[q * cap + slot];
#pragma unroll
          for (int q = 0; q < 8; q++) de[m][q] = dg[q * cap + slot];
        }
      }
#pragma unroll
      for (int m = 0; m < kCacheP; m++) {
        ty_p[m] = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) pp[m][q] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) dp[m][q] = 0.0;
        if (m * VT < kCacheP * kLmThreads) {
          const int slot_raw = kEdgeSlots + vt + m * VT;
          const int slot = slot_raw < cap ? slot_raw : 0;
          ty_p[m] = slot_raw < cap ? F.type[slot] : 0;
#pragma unroll
          for (int q = 0; q < 3; q++) pp[m][q] = F.p[q * cap + slot];
#pragma unroll
          for (int q = 0; q < 4; q++) dp[m][q] = dg[q * cap + slot];
        }
      }
#pragma unroll
      for (int m = 0; m < kCacheE; m++) {
        const bool live = ty_e[m] == 1;
#pragma unroll
        for (int q = 0; q < 3; q++) C.pe[m][q] = live ? (float)pe[m][q] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) C.de[m][q] = live ? de[m][q] : 0.0;
        if (live) C.live_e |= 1u << m;
      }
#pragma unroll
      for (int m = 0; m < kCacheP; m++) {
        const bool live = ty_p[m] == 2;   // LidarPlaneFactor
#pragma unroll
        for (int q = 0; q < 3; q++) C.pp[m][q] = live ? (float)pp[m][q] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) C.dp[m][q] = live ? dp[m][q] : 0.0;
        if (live) C.live_p |= 1u << m;
      }
    }
    if (first && MODE == kLmPacked) {
#pragma unroll
      for (int m = 0; m < kCacheE; m++) {
        const int k = vt + m * VT;
        const bool live = k < n_edge;
#pragma unroll
        for (int a = 0; a < 3; a++) C.pe[m][a] = live ? (float)cp[a * cap + k] : 0.f;
#pragma unroll
        for (int a = 0; a < 8; a++) C.de[m][a] = live ? cp[(3 + a) * cap + k] : 0.0;
      }
#pragma unroll
      for (int m = 0; m < kCacheP; m++) {
        const int q = vt + m * VT;
        const bool live = q < n_plane;
#pragma unroll
        for (int a = 0; a < 3; a++) C.pp[m][a] = live ? (float)cp[a * cap + n_edge + q] : 0.f;
#pragma unroll
        for (int a = 0; a < 4; a++) C.dp[m][a] = live ? cp[(3 + a) * cap + n_edge + q] : 0.0;
      }
    }
    if (first && MODE == kLmRowMask) {
      // self-compaction (see kLmRowMask): row offsets / masks were staged by the solve's prologue in the reduction buffer, which is free
      // until the first reduction.  Cached factors go to registers, factors beyond the cache are digested into cpack once (their later
      // evaluations stream them like the packed form), every factor leaves its slot in cslot for the residual hook.
      const int* rowoff = reinterpret_cast<const int*>(sh.red);
      const u64* rowmask = reinterpret_cast<const u64*>(sh.red + 512);
      const int nrows = cap >> 6;
      double* cpw = F.cpack;
      // cached factors: compact index -> slot (LDS tables), then point + producer-side digest (FactorTable::dg, k_map_fit) of all of them in
      // ONE memory trip; the slots stay in registers for the residual hook
      const double* dg = F.dg;
      double pe[kCacheE][3], de[kCacheE][8], pp[kCacheP][3], dp[kCacheP][4];
#pragma unroll
      for (int m = 0; m < kCacheE; m++) {
        C.slot_e[m] = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) pe[m][a] = 0.0;
#pragma unroll
        for (int a = 0; a < 8; a++) de[m][a] = 0.0;
        if (m * VT < n_edge) {   // uniform: cache slots nobody in the solve fills cost no loads (an ordinary sweep fills one edge and two plane slots per lane)
          const int k = vt + m * VT;
          const bool live = k < n_edge;
          const int slot = live ? lm_slot_of(rowoff, rowmask, nrows, k) : 0;
          C.slot_e[m] = slot;
          if (live) F.cslot[k] = slot;
#pragma unroll
          for (int a = 0; a < 3; a++) pe[m][a] = F.p[a * cap + slot];
#pragma unroll
          for (int a = 0; a < 8; a++) de[m][a] = dg[a * cap + slot];
        }
      }
#pragma unroll
      for (int m = 0; m < kCacheP; m++) {
        C.slot_p[m] = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) pp[m][a] = 0.0;
#pragma unroll
        for (int a = 0; a < 4; a++) dp[m][a] = 0.0;
        if (m * VT < n_plane) {
          const int q = vt + m * VT;
          const bool live = q < n_plane;
          const int slot = live ? lm_slot_of(rowoff, rowmask, nrows, n_edge + q) : 0;
          C.slot_p[m] = slot;
          if (live) F.cslot[n_edge + q] = slot;
#pragma unroll
          for (int a = 0; a < 3; a++) pp[m][a] = F.p[a * cap + slot];
#pragma unroll
          for (int a = 0; a < 4; a++) dp[m][a] = dg[a * cap + slot];
        }
      }
#pragma unroll
      for (int m = 0; m < kCacheE; m++) {
        const bool live = vt + m * VT < n_edge;
#pragma unroll
        for (int a = 0; a < 3; a++) C.pe[m][a] = live ? (float)pe[m][a] : 0.f;
#pragma unroll
        for (int a = 0; a < 8; a++) C.de[m][a] = live ? de[m][a] : 0.0;
      }
#pragma unroll
      for (int m = 0; m < kCacheP; m++) {
        const bool live = vt + m * VT < n_plane;
#pragma unroll
        for (int a = 0; a < 3; a++) C.pp[m][a] = live ? (float)pp[m][a] : 0.f;
#pragma unroll
        for (int a = 0; a < 4; a++) C.dp[m][a] = live ? dp[m][a] : 0.0;
      }
      for (int k = kCacheE * VT + vt; k < n_edge; k += VT) {
        double p[3], fr[8];
        const int slot = lm_slot_of(rowoff, rowmask, nrows, k);
        F.cslot[k] = slot;
        lm_digest_edge(F, slot, p, fr);
#pragma unroll
        for (int a = 0; a < 3; a++) cpw[a * cap + k] = p[a];
#pragma unroll
        for (int a = 0; a < 8; a++) cpw[(3 + a) * cap + k] = fr[a];
      }
      for (int q = kCacheP * VT + vt; q < n_plane; q += VT) {
        double p[3], nd[4];
        const int slot = lm_slot_of(rowoff, rowmask, nrows, n_edge + q);
        F.cslot[n_edge + q] = slot;
        lm_digest_plane(F, slot, p, nd);
#pragma unroll
        for (int a = 0; a < 3; a++) cpw[a * cap + n_edge + q] = p[a];
#pragma unroll
        for (int a = 0; a < 4; a++) cpw[(3 + a) * cap + n_edge + q] = nd[a];
      }
      // the staged tables make way for the reduction.  Only factors BEYOND the register cache are read back from cpack / cslot (and only then do
      // the stores above have to have landed); on an ordinary sweep there are none and an LDS-only barrier does (a full one waits out a store trip)
      if (n_edge > kCacheE * VT || n_plane > kCacheP * VT) __syncthreads(); else lds_barrier();
    }
#ifdef VLOAM_LM_STAMPS
    if (tid == 0 && first) sh.step_cyc[6] = clock64() - tf0;   // first evaluation: the cache fill (slot search, loads, barrier)
#endif
    auto put_resid = [&](int k, const double* r3, int slot_known) {   // slot_known >= 0: the factor's table slot is in a register (row-mask cache)
      const int slot = DIRECT ? k : (slot_known >= 0 ? slot_known : F.cslot[k]);
      F.resid[slot] = r3[0]; F.resid[cap + slot] = r3[1]; F.resid[2 * cap + slot] = r3[2];
    };
    // ---- edge factors (compact order == slot order: they come first).  Cached slots in packets of kPkE, streamed ones too.
    // (with NB workgroups a lane rarely owns more than one edge factor: packets of one instead of evaluating padding)
    constexpr int kPkE = NB > 1 ? 1 : 3, kPkP = (DIRECT && NB * kLmThreads >= kMaxFlat) ? 1 : 2;   // (scan-to-scan with >= 6 workgroups: a lane owns at most ONE plane slot — a packet of two would evaluate padding; zeros add exactly nothing, so the sums are the same bits)
    // g: first cache slot of the packet (compile-time at every call site), -1: streamed factors beyond the cache
    auto edge_resid = [&](const double (&c1)[kPkE], const double (&c2)[kPkE], const double (&e1)[kPkE][3], const double (&e2)[kPkE][3], int k0, int kstride, int g) {
#pragma unroll
      for (int u = 0; u < kPkE; u++) {
        const int k = k0 + u * kstride;
        const int known = (MODE == kLmRowMask && g >= 0 && g + u < kCacheE) ? C.slot_e[g + u < kCacheE ? g + u : 0] : -1;
        if (DIRECT ? ((C.live_e >> ((k0 - vt) / VT + u)) & 1u) != 0u : k < n_edge) {  // r = c1 e2 - c2 e1
          const double r3[3] = {c1[u] * e2[u][0] - c2[u] * e1[u][0], c1[u] * e2[u][1] - c2[u] * e1[u][1], c1[u] * e2[u][2] - c2[u] * e1[u][2]};
          put_resid(k, r3, known);
        }
      }
    };
    auto plane_resid = [&](const double (&r0)[kPkP], int q0, int qstride, int g) {
#pragma unroll
      for (int u = 0; u < kPkP; u++) {
        const int q = q0 + u * qstride;
        const int known = (MODE == kLmRowMask && g >= 0 && g + u < kCacheP) ? C.slot_p[g + u < kCacheP ? g + u : 0] : -1;
        if (DIRECT) { if ((C.live_p >> (q0 / VT + u)) & 1u) { const double r3[3] = {r0[u], 0.0, 0.0}; put_resid(kCacheE * kLmThreads + q, r3, -1); } }
        else if (q < n_plane) { const double r3[3] = {r0[u], 0.0, 0.0}; put_resid(n_edge + q, r3, known); }
      }
    };
    auto edge_packet = [&](const double (&p)[kPkE][3], const double (&e1)[kPkE][3], const double (&e2)[kPkE][3], const double (&d1)[kPkE],
                           const double (&d2)[kPkE], int k0 /* compact index of lane's first factor */, int kstride, int g) {
      double c1[kPkE], c2[kPkE];
      eval_edge_pk<kPkE>(p, e1, e2, d1, d2, Rm, tt, huber_a, acc, c1, c2);
      if (first) edge_resid(c1, c2, e1, e2, k0, kstride, g);
    };
    auto load_edge_slots = [&](int g, double (&p)[kPkE][3], double (&e1)[kPkE][3], double (&e2)[kPkE][3], double (&d1)[kPkE], double (&d2)[kPkE]) {
#pragma unroll
      for (int u = 0; u < kPkE; u++) {
        const int m = g + u < kCacheE ? g + u : kCacheE - 1;
        const bool have = g + u < kCacheE;  // compile-time: slots past the cache evaluate zeros
#pragma unroll
        for (int a = 0; a < 3; a++) { p[u][a] = have ? (double)C.pe[m][a] : 0.0; e1[u][a] = have ? C.de[m][a] : 0.0; e2[u][a] = have ? C.de[m][3 + a] : 0.0; }
        d1[u] = have ? C.de[m][6] : 0.0; d2[u] = have ? C.de[m][7] : 0.0;
      }
    };
    auto load_plane_slots = [&](int g, double (&p)[kPkP][3], double (&n)[kPkP][3], double (&d)[kPkP]) {
#pragma unroll
      for (int u = 0; u < kPkP; u++) {
        const int m = g + u < kCacheP ? g + u : kCacheP - 1;
        const bool have = g + u < kCacheP;
#pragma unroll
        for (int a = 0; a < 3; a++) { p[u][a] = have ? (double)C.pp[m][a] : 0.0; n[u][a] = have ? C.dp[m][a] : 0.0; }
        d[u] = have ? C.dp[m][3] : 0.0;
      }
    };
    auto edge_live = [&](int g) { return DIRECT ? __ballot(((C.live_e >> g) & ((1u << kPkE) - 1u)) != 0u) != 0ull : g * VT < n_edge; };
    auto plane_live = [&](int g) { return DIRECT ? __ballot(((C.live_p >> g) & ((1u << kPkP) - 1u)) != 0u) != 0ull : g * VT < n_plane; };
    // The FIRST edge packet and the FIRST plane packet of the cache side by side in one basic block (eval_mixed_pk): with NB workgroups that
    // is all a lane owns on an ordinary sweep, and the two dependency chains overlap instead of following each other.
    int ge0 = 0, gp0 = 0;
    if (edge_live(0) && plane_live(0)) {
      double pe[kPkE][3], e1[kPkE][3], e2[kPkE][3], d1[kPkE], d2[kPkE], pp[kPkP][3], n[kPkP][3], d[kPkP], c1[kPkE], c2[kPkE], r0[kPkP];
      load_edge_slots(0, pe, e1, e2, d1, d2);
      load_plane_slots(0, pp, n, d);
      eval_mixed_pk<kPkE, kPkP>(pe, e1, e2, d1, d2, pp, n, d, Rm, tt, huber_a, acc, c1, c2, r0);
      if (first) { edge_resid(c1, c2, e1, e2, vt, VT, 0); plane_resid(r0, vt, VT, 0); }
      ge0 = kPkE; gp0 = kPkP;
    }
#pragma unroll
    for (int g = 0; g < kCacheE; g += kPkE)
      if (g >= ge0 && edge_live(g)) {
        double p[kPkE][3], e1[kPkE][3], e2[kPkE][3], d1[kPkE], d2[kPkE];
        load_edge_slots(g, p, e1, e2, d1, d2);
        edge_packet(p, e1, e2, d1, d2, vt + g * VT, VT, g);
      }
    for (int base = ((kCacheE + kPkE - 1) / kPkE) * kPkE * VT; !DIRECT && base < n_edge; base += kPkE * VT) {  // beyond the cache: streamed
      double p[kPkE][3], e1[kPkE][3], e2[kPkE][3], d1[kPkE], d2[kPkE];
#pragma unroll
      for (int u = 0; u < kPkE; u++) {
        const int k = base + u * VT + vt;
        const bool live = k < n_edge;
#pragma unroll
        for (int a = 0; a < 3; a++) { p[u][a] = live ? cp[a * cap + k] : 0.0; e1[u][a] = live ? cp[(3 + a) * cap + k] : 0.0; e2[u][a] = live ? cp[(6 + a) * cap + k] : 0.0; }
        d1[u] = live ? cp[9 * cap + k] : 0.0; d2[u] = live ? cp[10 * cap + k] : 0.0;
      }
      edge_packet(p, e1, e2, d1, d2, base + vt, VT, -1);
    }
    // ---- plane factors
    auto plane_packet = [&](const double (&p)[kPkP][3], const double (&n)[kPkP][3], const double (&d)[kPkP], int q0, int qstride, int g) {
      double r0[kPkP];
      eval_plane_pk<kPkP>(p, n, d, Rm, tt, huber_a, acc, r0);
      if (first) plane_resid(r0, q0, qstride, g);
    };
#pragma unroll
    for (int g = 0; g < kCacheP; g += kPkP)
      if (g >= gp0 && plane_live(g)) {
        double p[kPkP][3], n[kPkP][3], d[kPkP];
        load_plane_slots(g, p, n, d);
        plane_packet(p, n, d, vt + g * VT, VT, g);
      }
    for (int base = ((kCacheP + kPkP - 1) / kPkP) * kPkP * VT; !DIRECT && base < n_plane; base += kPkP * VT) {
      double p[kPkP][3], n[kPkP][3], d[kPkP];
#pragma unroll
      for (int u = 0; u < kPkP; u++) {
        const int q = base + u * VT + vt;
        const bool live = q < n_plane;
#pragma unroll
        for (int a = 0; a < 3; a++) { p[u][a] = live ? cp[a * cap + n_edge + q] : 0.0; n[u][a] = live ? cp[(3 + a) * cap + n_edge + q] : 0.0; }
        d[u] = live ? cp[6 * cap + n_edge + q] : 0.0;
      }
      plane_packet(p, n, d, base + vt, VT, -1);
    }
  } else {
    for (int k = tid; k < n_valid; k += kLmThreads) {
      double r3[3];
      eval_vo(F.ctype[k], d3(F.cpack[k], F.cpack[cap + k], F.cpack[2 * cap + k]), d3(F.cpack[3 * cap + k], F.cpack[4 * cap + k], F.cpack[5 * cap + k]),
              xl, huber_a, acc, r3);
      if (first) {
        const int slot = F.cslot[k];
        F.resid[slot] = r3[0]; F.resid[cap + slot] = r3[1]; F.resid[2 * cap + slot] = r3[2];
      }
    }
  }
  *cyc_factors += clock64() - tf0;
#ifdef VLOAM_LM_STAMPS
  long long ev_t = clock64();
#define EV_STAMP(k) do { if (tid == 0 && !first) { const long long now_ = clock64(); sh.ev_cyc[k] += now_ - ev_t; ev_t = now_; } else if (tid == 0) ev_t = clock64(); } while (0)
  if (tid == 0 && !first) sh.ev_cyc[0] += ev_t - tf0;
#else
#define EV_STAMP(k) do { } while (0)
#endif
  // block reduction without cross-lane shuffles (a chain of ds_bpermute round trips is what dominated this kernel):
  // transpose through LDS, 8 strided sub-sums per value, then 8 -> 1.  Fixed order: bit-reproducible.
#pragma unroll
  for (int i = 0; i < kAcc; i++) sh.red[i * kRedStride + tid] = acc[i];
  EV_STAMP(1);
  lds_barrier();   // LDS only: sh.red (every lane's partial sums) -> the 8 * kAcc summing lanes
  EV_STAMP(2);
  if (tid < 8 * kAcc) {
    const int i = tid >> 3, sub = tid & 7;
    const double* col = sh.red + i * kRedStride + sub;
    // 8 independent partial sums (a dependent f64 add costs a lone wavefront ~32 cycles: 32 in a row were a third of the reduction) from
    // 16-byte LDS reads — lane (value i, sub) takes the lanes {2 sub, 2 sub + 1} + 16 j of the workgroup —, then a fixed tree: the order is
    // part of the result and the same in every workgroup
    const double* colp = sh.red + i * kRedStride + 2 * sub;
    double sx[4] = {0.0, 0.0, 0.0, 0.0}, sy[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < kLmThreads / 16; j += 4) {
#pragma unroll
      for (int c = 0; c < 4; c++) { const double2 v = *reinterpret_cast<const double2*>(colp + 16 * (j + c)); sx[c] += v.x; sy[c] += v.y; }
    }
    (void)col;
    sh.part[sub * kAcc + i] = ((sx[0] + sx[1]) + (sx[2] + sx[3])) + ((sy[0] + sy[1]) + (sy[2] + sy[3]));
  }
  EV_STAMP(3);
  lds_barrier();   // LDS only: sh.part (the strided sub-sums) -> the lanes that fold them / publish the granules
  EV_STAMP(4);
  if (tid < kAcc) {
    double pw[8];
#pragma unroll
    for (int w = 0; w < 8; w++) pw[w] = sh.part[w * kAcc + tid];
    const double s = ((pw[0] + pw[1]) + (pw[2] + pw[3])) + ((pw[4] + pw[5]) + (pw[6] + pw[7]));
    if constexpr (NB == 1) s_out[tid] = s;
    else {
      // Exchange of the partial sums between the NB workgroups WITHOUT a barrier: every f64 travels as two naturally aligned 8-byte
      // granules {32 payload bits, 32-bit tag}, tag = (solve generation, evaluation) — an 8-byte store is single-copy atomic, so a
      // reader that sees the tag it waits for holds the payload that was written with it, and there is nothing to order.  Each lane
      // publishes its value and then polls the NB x 2 granules of its own column: one store trip + the slowest workgroup's arrival,
      // instead of store -> release fetch-add -> poll -> load (two and a half trips).  Double-buffered by evaluation parity: nobody can
      // publish evaluation e + 2 before everybody has READ evaluation e (it needs everybody's e + 1 for that).  Every workgroup adds
      // the partials in workgroup order, so all of them hold bit-identical accumulators and run the (cheap) trust-region bookkeeping
      // redundantly.  Never hangs: a lane that gives up poisons the solve (every poller reads the poison word each round), the host is
      // told through the sticky error word (vloam_sync), and the solve is abandoned with x unchanged.
      u64* gw = reinterpret_cast<u64*>(F.gsync);
      u64* gran = gw + 8 + (size_t)(eval_idx & 1) * kLmMaxBlocks * 64;
      const u64 tag = (u64)(tag_base + (unsigned)eval_idx + 1u) << 32;
      const u64 poison = (u64)(tag_base >> 8) + 1ull;
      const u64 bits = (u64)__double_as_longlong(s);
      __hip_atomic_store(&gran[blk * 64 + tid], (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&gran[blk * 64 + 32 + tid], (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      EV_STAMP(5);
      u64 lo[NB], hi[NB];
      bool bad = false;
      for (int spins = 0;; spins++) {
#pragma unroll
        for (int q = 0; q < NB; q++) {
          lo[q] = __hip_atomic_load(&gran[q * 64 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hi[q] = __hip_atomic_load(&gran[q * 64 + 32 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool all = true;
#pragma unroll
        for (int q = 0; q < NB; q++) all = all && (lo[q] >> 32 << 32) == tag && (hi[q] >> 32 << 32) == tag;
        if (all) break;
        if (__hip_atomic_load(&gw[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == poison) { bad = true; break; }
        if (spins > F.spin_limit) {   // (~0.5 s by default; k_lm_solve then degrades to a one-workgroup solve)
          __hip_atomic_store(&gw[1], poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      EV_STAMP(6);
      if (bad) sh.failed = 1;  // nobody continues with partial sums that may be incomplete
      double pv[8];   // workgroup order, folded as a fixed tree (every workgroup adds the same values the same way: bit-identical accumulators)
#pragma unroll
      for (int q = 0; q < 8; q++) pv[q] = q < NB ? __longlong_as_double((long long)((hi[q < NB ? q : 0] << 32) | (lo[q < NB ? q : 0] & 0xffffffffull))) : 0.0;
      s_out[tid] = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
    }
  }
  lds_barrier();   // LDS only: s_out (this workgroup's or the exchanged totals) -> every lane of the caller
  EV_STAMP(7);
}

// packed upper triangle accessor; a, b are compile-time constants at every call site after unrolling
__device__ __forceinline__ double Hget(const double* acc, int a, int b) {
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  return acc[7 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
}

// 6x6 SPD solve by Cholesky on the packed lower triangle, in place (fully unrolled: lives in registers).  Column by column
// with one reciprocal square root per pivot and no divisions — f64 div / sqrt are ~150-cycle dependent sequences here and
// this runs on a single lane.
#define LIDX(i, j) ((i) * ((i) + 1) / 2 + (j))
__device__ __forceinline__ bool chol6_solve(double (&L)[21], const double (&rhs)[6], double (&y)[6]) {
  bool ok = true;
  double inv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double s = L[LIDX(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= L[LIDX(j, k)] * L[LIDX(j, k)];
    ok = ok && (s > 0.0);
    inv[j] = rsqrt(s);
    L[LIDX(j, j)] = s * inv[j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = L[LIDX(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= L[LIDX(i, k)] * L[LIDX(j, k)];
      L[LIDX(i, j)] = t * inv[j];
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { double s = rhs[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[LIDX(i, k)] * z[k];
    z[i] = s * inv[i]; }
#pragma unroll
  for (int i = 5; i >= 0; i--) { double s = z[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= L[LIDX(k, i)] * y[k];
    y[i] = s * inv[i]; }
  return ok;
}

// sin(n) / n and cos(n) from n^2.  The trust-region step is a small rotation: below |n| = 0.5 both are short alternating
// series in n^2 (next term < 1e-25 relative), with no square root, division or argument reduction.
__device__ __forceinline__ void sinc_cos(double n2, double* sinc, double* c) {
  if (n2 < 0.25) {
    double s = 1.0 / 121645100408832000.0, q = 1.0 / 6402373705728000.0;  // 1/19!, 1/18!
    s = 1.0 / 355687428096000.0 - n2 * s;   q = 1.0 / 20922789888000.0 - n2 * q;   // 1/17!, 1/16!
    s = 1.0 / 1307674368000.0 - n2 * s;     q = 1.0 / 87178291200.0 - n2 * q;       // 1/15!, 1/14!
    s = 1.0 / 6227020800.0 - n2 * s;        q = 1.0 / 479001600.0 - n2 * q;         // 1/13!, 1/12!
    s = 1.0 / 39916800.0 - n2 * s;          q = 1.0 / 3628800.0 - n2 * q;           // 1/11!, 1/10!
    s = 1.0 / 362880.0 - n2 * s;            q = 1.0 / 40320.0 - n2 * q;             // 1/9!, 1/8!
    s = 1.0 / 5040.0 - n2 * s;              q = 1.0 / 720.0 - n2 * q;               // 1/7!, 1/6!
    s = 1.0 / 120.0 - n2 * s;               q = 1.0 / 24.0 - n2 * q;                // 1/5!, 1/4!
    s = 1.0 / 6.0 - n2 * s;                 q = 0.5 - n2 * q;                       // 1/3!, 1/2!
    *sinc = 1.0 - n2 * s;                   *c = 1.0 - n2 * q;
  } else {
    const double n = sqrt(n2);
    double sn, cs;
    sincos(n, &sn, &cs);
    *sinc = sn / n; *c = cs;
  }
}

template <bool QUAT>
__device__ __forceinline__ void lm_plus_t(const double* x, const double (&delta)[6], double* out) {
  if (QUAT) {  // EigenQuaternionParameterization::Plus: x_plus = (sin|d|/|d| d, cos|d|) * x   (|d| == 0 gives x back exactly)
    double sc, cs;
    sinc_cos(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2], &sc, &cs);
    const double dq[4] = {sc * delta[0], sc * delta[1], sc * delta[2], cs};
    quat_mul(dq, x, out);
#pragma unroll
    for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + delta[3 + i];
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) out[i] = x[i] + delta[i];
  }
}

// t_w_curr += q_w_curr * t_last_curr;  q_w_curr = q_w_curr * q_last_curr  (laser_odometry.cpp:530-531), then the trajectory row.
// x = (q_last_curr, t_last_curr) as just solved.  Same expressions as k_lo_finish.
__device__ void lo_integrate(LOState* lo, const double* x, double* traj_row14) {
  const double* q = lo->q_w_curr;
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  const double vx = x[4], vy = x[5], vz = x[6];
  double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
  cx = cx + cx; cy = cy + cy; cz = cz + cz;
  const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
  const double t0 = lo->t_w_curr[0] + ((vx + w * cx) + dx), t1 = lo->t_w_curr[1] + ((vy + w * cy) + dy), t2 = lo->t_w_curr[2] + ((vz + w * cz) + dz);
  double r[4];
  r[0] = q[3] * x[0] + q[0] * x[3] + q[1] * x[2] - q[2] * x[1];
  r[1] = q[3] * x[1] + q[1] * x[3] + q[2] * x[0] - q[0] * x[2];
  r[2] = q[3] * x[2] + q[2] * x[3] + q[0] * x[1] - q[1] * x[0];
  r[3] = q[3] * x[3] - q[0] * x[0] - q[1] * x[1] - q[2] * x[2];
  lo->t_w_curr[0] = t0; lo->t_w_curr[1] = t1; lo->t_w_curr[2] = t2;
  for (int k = 0; k < 4; k++) lo->q_w_curr[k] = r[k];
  tf_lo_publish(lo, x);  // LaserOdometry::publish's LO -> VO prior (laser_odometry.cpp:563-567), coupled mode only
  if (traj_row14) {
    for (int k = 0; k < 4; k++) traj_row14[k] = r[k];
    traj_row14[4] = t0; traj_row14[5] = t1; traj_row14[6] = t2;
    for (int k = 0; k < 7; k++) traj_row14[7 + k] = traj_row14[k];  // until mapping overwrites it, the map pose equals the odometry pose
  }
}

// What one attempt at a solve leaves behind for the kernel's epilogue.
struct LmRun {
  double minimum_cost;
  int termination, n_rec, n_evals;
  bool failed;          // cooperative form only: a workgroup gave up waiting for its partners
  long long cyc_fac, cyc_eval, cyc_serial;
};

// One attempt: prologue (factor counts / self-compaction tables), first evaluation, the trust-region loop.  NB workgroups cooperate
// (NB == 1: this workgroup alone).  sh.x / sh.x0 hold the start point; on return sh.best holds the answer unless the attempt failed.
template <bool QUAT, int MODE, int NB>
__device__ __forceinline__ LmRun lm_solve_run(FactorTable& F, int edge_rows, LMRecord* rec, int max_iters, double huber_a, LmShared& sh, bool lead,
                                              int row_first, ulonglong2 mask_first, unsigned tag_base) {
  const int tid = threadIdx.x;
  constexpr int na = QUAT ? 7 : 6;
  // ---- prologue: count the factors.  Packed / direct: per-row counters (released for the next solve); row-mask form: scan the row
  // masks into row offsets and stage both in LDS for the self-compaction of the first evaluation (lm_evaluate)
  if constexpr (MODE == kLmRowMask) {
    static_assert(kLmThreads == 256, "two rows per lane cover tables of up to 512 rows");
    int* rowoff = reinterpret_cast<int*>(sh.red);            // [513]
    u64* rowmask_s = reinterpret_cast<u64*>(sh.red + 512);   // [512]
    const int nrows = F.cap >> 6;
    const int c0 = __popcll(mask_first.x), c = c0 + __popcll(mask_first.y);
    int inc = c;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if ((tid & 63) >= d) inc += o; }
    if ((tid & 63) == 63) sh.scan[tid >> 6] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += sh.scan[w];
    const int excl = base + inc - c;
    rowoff[2 * tid] = excl; rowoff[2 * tid + 1] = excl + c0;
    rowmask_s[2 * tid] = mask_first.x; rowmask_s[2 * tid + 1] = mask_first.y;
    if (tid == kLmThreads - 1) rowoff[2 * kLmThreads] = excl + c;
    __syncthreads();
    if (tid == 0) { sh.n_valid = rowoff[nrows]; sh.n_edge = rowoff[edge_rows]; }
    __syncthreads();
  } else {
    const int nrows = F.cap >> 6;
    int c = 0, ce = 0;
    for (int r = tid; r < nrows; r += kLmThreads) { const int q = r == tid ? row_first : F.rowcnt[r]; c += q; if (r < edge_rows) ce += q; }
    for (int d = 32; d > 0; d >>= 1) { c += __shfl_xor(c, d); ce += __shfl_xor(ce, d); }
    if ((tid & 63) == 0) { sh.scan[tid >> 6] = c; sh.scan2[tid >> 6] = ce; }
    __syncthreads();
    if (tid == 0) {
      int t = 0, te = 0;
      for (int w = 0; w < kLmThreads / 64; w++) { t += sh.scan[w]; te += sh.scan2[w]; }
      sh.n_valid = t; sh.n_edge = te;
    }
    if (NB == 1) for (int r = tid; r < nrows; r += kLmThreads) F.rowcnt[r] = 0;  // NB > 1: once every workgroup has read them (below)
    __syncthreads();
  }
  const long long t_pro = clock64();
  long long cyc_eval = 0, cyc_serial = 0, cyc_fac = 0, t_mark;
  const int n_valid = sh.n_valid, n_edge = sh.n_edge;

  LmCache cache;
  t_mark = clock64();
  int eval_idx = 0;
  lm_evaluate<QUAT, MODE, NB>(F, n_edge, n_valid, sh.x, huber_a, sh, sh.acc2[0], true, cache, &cyc_fac, eval_idx++, tag_base);
  cyc_eval += clock64() - t_mark;

#ifdef VLOAM_LM_STAMPS   // debug build only: where thread 0's serial sections spend their cycles (sums over the iterations -> trace rows 100, 101)
  long long lm_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lm_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LM_STAMP(k) do { lm_t[k] = clock64(); if ((k) > 0 && (k) != 6) lm_sum[k] += lm_t[k] - lm_t[(k) - 1]; } while (0)
#else
#define LM_STAMP(k) do { } while (0)
#endif
  // ---- trust-region state: registers of thread 0 (statically indexed); other threads only follow sh.go
  double radius = 1e4, inv_radius = 1e-4, decrease_factor = 2.0, minimum_cost = DBL_MAX, current_cost = 0, x_cost = 0, x_norm = 0, gmax = 0;   // inv_radius: what the step needs, kept by MULTIPLICATION (the division for `radius` itself — trace, termination test — is off the step's dependency chain)
  int num_invalid = 0, iteration = 0, n_rec = 0, termination = 0, n_evals = 1;
  double it_cost = 0, it_cost_change = 0, it_step_norm = 0, it_rho = 0;
  bool it_valid = true, it_success = true;

  auto grad_max = [&](const double* xx, const double* acc) {
    double ng[6], pg[7];
#pragma unroll
    for (int a = 0; a < 6; a++) ng[a] = -acc[1 + a];
    lm_plus_t<QUAT>(xx, ng, pg);
    double m = 0;
#pragma unroll
    for (int i = 0; i < na; i++) m = fmax(m, fabs(xx[i] - pg[i]));
    return m;
  };

  // helper lanes on other wavefronts take the slow scalar pieces (f64 sqrt / div / sincos) off thread 0's critical path
  auto x_norm_of = [&](const double* xx) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < na; i++) s += xx[i] * xx[i];
    return sqrt(s);
  };
  // What LevenbergMarquardtStrategy::ComputeStep needs of a point — the Jacobi-scaled normal equations Hs = S H S (packed lower
  // triangle), gs = S g, and the clamped LM diagonal — by 27 lanes of wavefront 3 at once, whenever the point changes (the start and
  // every accepted step; rejected steps keep all three, which is Ceres' reuse_diagonal).  The trust-region step of thread 0 then
  // starts from 33 LDS reads instead of rebuilding ~150 products on one lane.  jacobi_scaling is fixed at iteration 0.
  auto refresh_normal_equations = [&](const double* acc, bool first, int buf) {
    const int j = tid - 192;
    if (j < 0 || j >= 27) return;
    if (j < 21) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= j) a++;
      const int b = j - a * (a + 1) / 2;   // j == LIDX(a, b), a >= b
      const double sa = first ? 1.0 / (1.0 + sqrt(acc[7 + a * 6 - (a * (a - 1)) / 2])) : sh.scale[a];
      const double sb = first ? 1.0 / (1.0 + sqrt(acc[7 + b * 6 - (b * (b - 1)) / 2])) : sh.scale[b];
      sh.Hs[buf][j] = acc[7 + b * 6 - (b * (b - 1)) / 2 + (a - b)] * sa * sb;
    } else {
      const int a = j - 21;
      const double haa = acc[7 + a * 6 - (a * (a - 1)) / 2];
      const double sa = first ? 1.0 / (1.0 + sqrt(haa)) : sh.scale[a];
      if (first) sh.scale[a] = sa;
      sh.gs[buf][a] = acc[1 + a] * sa;
      sh.diag[buf][a] = fmin(fmax(haa * sa * sa, 1e-6), 1e32);
    }
  };
  refresh_normal_equations(sh.acc2[0], true, 0);
  if (tid == 64) sh.gmax_c = grad_max(sh.x, sh.acc2[0]);
  if (tid == 128) sh.xnorm_c = x_norm_of(sh.x);
  lds_barrier();   // LDS only: sh.acc2[0], sh.gmax_c, sh.xnorm_c, sh.failed -> every lane
  const bool failed_at_start = NB > 1 && sh.failed;  // (uniform: written before the barrier that ends lm_evaluate)
  if (tid == 0) {
    const double* cur = sh.acc2[0];
    x_cost = cur[0];
    gmax = sh.gmax_c;
    x_norm = sh.xnorm_c;
    current_cost = x_cost;
    it_cost = x_cost;
#pragma unroll
    for (int i = 0; i < 7; i++) sh.best[i] = i < na ? sh.x[i] : 0.0;
    if (lead) {
#pragma unroll
      for (int i = 0; i < 7; i++) rec->x_in[i] = sh.best[i];
#pragma unroll
      for (int a = 0; a < 6; a++) { rec->g0[a] = cur[1 + a];
#pragma unroll
        for (int b = 0; b < 6; b++) rec->H0[a * 6 + b] = Hget(cur, a, b); }
      rec->initial_cost = x_cost;
      rec->n_factors = n_valid;
    }
  }

  int curidx = 0;  // uniform copy of sh.curidx
  for (; !failed_at_start;) {
    t_mark = clock64();
    if (tid == 0) {
      LM_STAMP(0);
      int go = -1;  // -1: invalid step, loop again inside thread 0; 0: stop; 1: evaluate the candidate
      while (go < 0) {
        // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
        if (it_success && x_cost < minimum_cost) { minimum_cost = x_cost;
#pragma unroll
          for (int i = 0; i < na; i++) sh.best[i] = sh.x[i]; }
        if (lead && n_rec < kLmMaxTrace - 1) {   // (row kLmMaxTrace - 1 carries the running in-kernel time below, never an iteration)
          double* row = rec->trace[n_rec];
          row[0] = it_cost; row[1] = it_cost_change; row[2] = gmax; row[3] = it_step_norm; row[4] = it_rho; row[5] = radius;
          row[6] = it_valid; row[7] = it_success;
        }
        n_rec++;
        if (iteration >= max_iters) { termination = 0; go = 0; break; }
        if (gmax <= 1e-10) { termination = 1; go = 0; break; }
        if (radius <= 1e-32) { termination = 1; go = 0; break; }
        iteration++;
        // ---- LevenbergMarquardtStrategy::ComputeStep on the normal equations of the column-scaled Jacobian:
        // (Hs + D / radius) y = gs by Cholesky, step = -y
        double gs[6], dg[6], L[21], y[6], sc6[6];
#pragma unroll
        for (int i = 0; i < 21; i++) L[i] = sh.Hs[curidx][i];  // one batch of independent LDS reads
#pragma unroll
        for (int a = 0; a < 6; a++) { gs[a] = sh.gs[curidx][a]; dg[a] = sh.diag[curidx][a]; sc6[a] = sh.scale[a]; }
        LM_STAMP(1);   // bookkeeping + trace row + LDS reads issued
#pragma unroll
        for (int a = 0; a < 6; a++) { dg[a] = dg[a] * inv_radius; L[LIDX(a, a)] += dg[a]; }
        LM_STAMP(2);   // reciprocal radius, damped diagonal
        bool ok = chol6_solve(L, gs, y);
        LM_STAMP(3);   // Cholesky solve
#pragma unroll
        for (int a = 0; a < 6; a++) ok = ok && isfinite(y[a]);
        double model_cost_change = 0;
        it_valid = false;
        if (ok) {
          // model_cost_change = -(Js s)^T (r + Js s / 2) = -s.gs - s^T Hs s / 2 with s = -y; Hs y = gs - (D / radius) y gives
          // s^T Hs s = y.gs - sum (D_a / radius) y_a^2 — twelve products instead of the 6 x 6 quadratic form
          double yg = 0, yDy = 0;
#pragma unroll
          for (int a = 0; a < 6; a++) { yg += y[a] * gs[a]; yDy += dg[a] * y[a] * y[a]; }
          model_cost_change = 0.5 * (yg + yDy);
          it_valid = model_cost_change > 0.0;
        }
        if (!it_valid) {
          // ---- HandleInvalidStep
          if (++num_invalid >= 5) { termination = 2; go = 0; break; }
          radius = radius * (1.0 / decrease_factor);   // LevenbergMarquardtStrategy::StepIsInvalid == StepRejected(0.0): divide by decrease_factor, which doubles (oracle/orc_ceres.cpp)
          inv_radius = inv_radius * decrease_factor;
          decrease_factor *= 2.0;
          it_cost = x_cost; it_cost_change = 0; it_step_norm = 0; it_rho = 0; it_success = false;
          continue;
        }
        num_invalid = 0;
        double delta[6];
#pragma unroll
        for (int a = 0; a < 6; a++) delta[a] = -y[a] * sc6[a];
        LM_STAMP(4);   // model cost change, step
        lm_plus_t<QUAT>(sh.x, delta, sh.xc);
        sh.mcc = model_cost_change;
        go = 1;
        LM_STAMP(5);   // plus
      }
      sh.go = go;
    }
    lds_barrier();   // LDS only: sh.go, sh.dx / the trial point -> every lane
    cyc_serial += clock64() - t_mark;
    if (sh.go == 0) break;
    t_mark = clock64();
    if (tid == 160) {   // known before the candidate is evaluated, needed right after: off thread 0's chain (a reciprocal and a square root are ~25 dependent operations)
      double sn = 0;
#pragma unroll
      for (int i = 0; i < na; i++) sn += (sh.x[i] - sh.xc[i]) * (sh.x[i] - sh.xc[i]);
      sh.step_norm_c = sqrt(sn);
      sh.inv_mcc = 1.0 / sh.mcc;
    }
    double* cand = sh.acc2[curidx ^ 1];
    lm_evaluate<QUAT, MODE, NB>(F, n_edge, n_valid, sh.xc, huber_a, sh, cand, false, cache, &cyc_fac, eval_idx++, tag_base);
    cyc_eval += clock64() - t_mark;
    if (NB > 1 && sh.failed) break;  // uniform across the workgroup; every workgroup that still waits sees the poison word
    t_mark = clock64();
    // speculative (used only if the step is accepted), concurrent with thread 0's acceptance test: gradient norm, |x|, and — on
    // wavefront 3 — the candidate's scaled normal equations into the set that goes with its accumulators
    if (tid == 64) sh.gmax_c = grad_max(sh.xc, cand);
    if (tid == 128) sh.xnorm_c = x_norm_of(sh.xc);
    refresh_normal_equations(cand, false, curidx ^ 1);
    bool accepted = false;
    if (tid == 0) {
      LM_STAMP(6);
      n_evals++;
      double candidate_cost = cand[0];
      if (!isfinite(candidate_cost)) candidate_cost = DBL_MAX;
      bool stop = false;
      it_step_norm = sh.step_norm_c;
      if (it_step_norm <= 1e-8 * (x_norm + 1e-8)) { termination = 1; stop = true; }  // ParameterToleranceReached
      if (!stop) {
        it_cost_change = x_cost - candidate_cost;
        if (fabs(it_cost_change) <= 1e-6 * x_cost) { termination = 1; stop = true; }  // FunctionToleranceReached
      }
      if (!stop) {
        it_rho = candidate_cost >= DBL_MAX ? -DBL_MAX : (current_cost - candidate_cost) * sh.inv_mcc;   // (model_cost_change > 0: a valid step)
        if (it_rho > 1e-3) {  // HandleSuccessfulStep
          accepted = true;
#pragma unroll
          for (int i = 0; i < na; i++) sh.x[i] = sh.xc[i];
          sh.curidx = curidx ^ 1;  // the candidate's accumulators become the current point's
          x_cost = candidate_cost;
          it_cost = x_cost; it_success = true;
          { const double c = 2.0 * it_rho - 1.0, f = fmax(1.0 / 3.0, 1.0 - c * c * c); radius = radius / f; inv_radius = inv_radius * f; }
          if (radius >= 1e16) { radius = 1e16; inv_radius = 1e-16; }
          decrease_factor = 2.0;
          current_cost = candidate_cost;
        } else {              // HandleUnsuccessfulStep
          it_success = false;
          radius = radius * (1.0 / decrease_factor);  // decrease_factor is a power of two: exact, and folded to a multiply
          inv_radius = inv_radius * decrease_factor;
          decrease_factor *= 2.0;
          it_cost = candidate_cost;
        }
      }
      sh.go2 = stop ? 0 : 1;
      LM_STAMP(7);   // acceptance test
    }
    lds_barrier();   // LDS only: sh.go2, sh.gmax_c, sh.xnorm_c, the accepted point -> every lane
    if (accepted) { gmax = sh.gmax_c; x_norm = sh.xnorm_c; }
    cyc_serial += clock64() - t_mark;
    if (sh.go2 == 0) break;
    curidx = sh.curidx;   // accepted (uniform): the candidate's accumulators and its prepared normal equations become the current point's
  }

#ifdef VLOAM_LM_STAMPS
  if (tid == 0) for (int k = 0; k < 8; k++) if (k != 6) sh.step_cyc[k] = lm_sum[k];   // ([6]: the first evaluation's cache fill, lm_evaluate)
#endif
  LmRun R;
  R.minimum_cost = minimum_cost; R.termination = termination; R.n_rec = n_rec; R.n_evals = n_evals;
  R.failed = NB > 1 && sh.failed != 0;
  R.cyc_fac = cyc_fac; R.cyc_eval = cyc_eval; R.cyc_serial = cyc_serial;
  (void)t_pro;
  return R;
}

#ifndef VLOAM_LM_WPE
#define VLOAM_LM_WPE 1   // wavefronts per SIMD the solve is compiled for (1: the whole register file; A/B builds: 2)
#endif
template <bool QUAT, int MODE, int NB>
__global__ __launch_bounds__(kLmThreads) __attribute__((amdgpu_waves_per_eu(VLOAM_LM_WPE, VLOAM_LM_WPE))) void k_lm_solve(FactorTable F, int edge_rows, double* x_io, LMRecord* rec, int max_iters,
                                                         double huber_a, const int* enable_flag, LOState* fin_lo, double* fin_traj, size_t ss) {
  VL_SESSION(ss); F.rebase(so_); RB(x_io); RB(rec); RB(enable_flag); RB(fin_lo); RB(fin_traj);
  if (NB > 1 && lm_coop_block<NB, MODE>() < 0) return;   // single sequence: only the workgroups of the solve's XCD work (lm_coop_block)
  __shared__ LmShared sh;
  const int tid = threadIdx.x;
  const bool lead = NB == 1 || lm_coop_block<NB, MODE>() == 0;  // the workgroup that owns every global side effect other than its factors' residuals
  constexpr int na = QUAT ? 7 : 6;
  // the gate word, the parameters and this lane's first row counter are fetched in ONE round trip (a branch on the gate first would put a
  // dependent ~1.5 us memory trip in front of everything else the solve reads)
  const int enabled = enable_flag ? *enable_flag : 1;
  const double x_first = tid < na ? x_io[tid] : 0.0;
  int row_first = 0;
  ulonglong2 mask_first = make_ulonglong2(0ull, 0ull);   // kLmRowMask: the accepted-slot masks of rows 2 tid, 2 tid + 1
  if constexpr (MODE == kLmRowMask) { if (2 * tid < (F.cap >> 6)) mask_first = *reinterpret_cast<const ulonglong2*>(F.rowmask + 2 * tid); }
  else row_first = tid < (F.cap >> 6) ? F.rowcnt[tid] : 0;
  // generation of this launch (a host-side counter, lm_launch): tags the partial sums the workgroups of a cooperative solve publish
  // (lm_evaluate), so that nothing an earlier solve left in the exchange buffer — or a straggler of an abandoned attempt — can be taken
  // for this solve's data
  const unsigned tag_base = (F.gen & 0xffffffu) << 8;
  if (enabled == 0) {
    if (MODE != kLmRowMask && lead) for (int r = tid; r < (F.cap >> 6); r += kLmThreads) F.rowcnt[r] = 0;
    return;
  }
  if (tid < na) { sh.x[tid] = x_first; sh.x0[tid] = x_first; }
  if (tid == 7) { sh.x[7] = 0.0; sh.failed = 0; sh.curidx = 0; }
#ifdef VLOAM_LM_STAMPS
  if (tid < 8) sh.ev_cyc[tid] = 0;
#endif
  const long long t_start = clock64();
  LmRun R = lm_solve_run<QUAT, MODE, NB>(F, edge_rows, rec, max_iters, huber_a, sh, lead, row_first, mask_first, tag_base);
  if constexpr (NB > 1) {
    if (R.failed) {
      // A partner never showed up (not co-resident: a CU-masked queue, a partitioned device, a chip held by another process' spinning
      // solves) or gave up itself.  The solve DEGRADES instead of failing: the lead workgroup starts over on its own — same factors, same
      // trust-region loop, the partial sums added in one workgroup's order (poses agree with the cooperative form to round-off) — and
      // counts it (vloam_get_health); the host then stops launching cooperative solves on this handle (vloam_sync).  Every other
      // workgroup just leaves; a straggler that arrives later finds the poison word of ITS generation and leaves as well.
      if (!lead) return;
      __syncthreads();
      if (tid < na) sh.x[tid] = sh.x0[tid];
      if (tid == 7) { sh.x[7] = 0.0; sh.failed = 0; sh.curidx = 0; }
      R = lm_solve_run<QUAT, MODE, 1>(F, edge_rows, rec, max_iters, huber_a, sh, true, row_first, mask_first, tag_base);
      if (tid == 0 && F.fallbacks) atomicAdd(F.fallbacks, 1);
      // the host learns without a synchronisation (it polls this host-mapped word before every enqueue): no further cooperative launches
      if (tid == 0 && F.host_degraded) __hip_atomic_store(F.host_degraded, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (MODE != kLmRowMask && lead) {
      for (int r = tid; r < (F.cap >> 6); r += kLmThreads) F.rowcnt[r] = 0;  // every workgroup published its partial sums, i.e. is past its prologue's reads
    }
  }
  const double minimum_cost = R.minimum_cost;
  const int termination = R.termination, n_rec = R.n_rec, n_evals = R.n_evals;
  const long long cyc_fac = R.cyc_fac, cyc_eval = R.cyc_eval, cyc_serial = R.cyc_serial;
  if (tid == 0 && lead) {
#pragma unroll
    for (int i = 0; i < na; i++) x_io[i] = sh.best[i];
#pragma unroll
    for (int i = 0; i < 7; i++) rec->x_out[i] = i < na ? sh.best[i] : 0.0;
    if (fin_lo) lo_integrate(fin_lo, sh.best, fin_traj);  // LaserOdometry's pose integration rides on its last solve
    rec->final_cost = minimum_cost;
    rec->n_iterations = n_rec < kLmMaxTrace - 1 ? n_rec : kLmMaxTrace - 1;
    rec->termination = termination;
    rec->n_evals = n_evals;
    rec->cyc[0] = (double)cyc_fac;  // factor loops only (evaluations minus the block reductions)
    rec->cyc[1] = (double)cyc_eval; rec->cyc[2] = (double)cyc_serial;
    rec->cyc[3] = (double)(clock64() - t_start);
    rec->trace[kLmMaxTrace - 1][0] += rec->cyc[3]; rec->trace[kLmMaxTrace - 1][1] += 1.0;   // running sum / count over the handle's life (row 103 is never a real iteration: tools read the average in-kernel time of a solve under load from it)
#ifdef VLOAM_LM_STAMPS
    for (int k = 0; k < 8; k++) { rec->trace[100][k] = (double)sh.step_cyc[k]; rec->trace[101][k] = (double)sh.ev_cyc[k]; }
#endif
  }
}

// Deterministic stream compaction of the accepted factors (slot order): one wavefront per 64-slot row, many rows in flight
// across the chip (a single workgroup would expose the full HBM latency of every row).  Edge factors get
// v = (b - a) / |a - b| precomputed in place of b (lidarFactor.hpp:37-42 divides by de.norm()).
__global__ __launch_bounds__(64) void k_lm_compact(FactorTable F, int quat, const int* enable_flag, size_t ss) {
  VL_SESSION(ss); F.rebase(so_); RB(enable_flag);
  if (enable_flag && *enable_flag == 0) return;
  const int r = blockIdx.x, lane = threadIdx.x;
  if (F.rowcnt[r] == 0) return;
  int off = 0;
  for (int q = lane; q < r; q += 64) off += F.rowcnt[q];
  for (int d = 32; d > 0; d >>= 1) off += __shfl_xor(off, d);
  const int cap = F.cap;
  const int k = (r << 6) + lane;
  const int ty = F.type[k];
  double v[11];
#pragma unroll
  for (int a = 0; a < 3; a++) { v[a] = F.p[a * cap + k]; v[3 + a] = F.A[a * cap + k]; v[6 + a] = F.B[a * cap + k]; }
  v[9] = 0.0; v[10] = 0.0;
  const unsigned long long m = __ballot(ty != 0);
  if (ty) {
    const int o = off + __popcll(m & ((1ull << lane) - 1ull));
    F.ctype[o] = ty; F.cslot[o] = k;
    if (quat && ty == 1) {
      double fr[8];
      edge_frame(v[3], v[4], v[5], v[6], v[7], v[8], fr);
#pragma unroll
      for (int q = 0; q < 8; q++) v[3 + q] = fr[q];
    } else if (quat && ty == 2) {  // LidarPlaneFactor (lp - j) . n  ->  n . lp + d with d = -(n . j); A := n, B.x := d
      const double d = -(v[6] * v[3] + v[7] * v[4] + v[8] * v[5]);
      v[3] = v[6]; v[4] = v[7]; v[5] = v[8]; v[6] = d;
    }
#pragma unroll
    for (int a = 0; a < 11; a++) F.cpack[a * cap + o] = v[a];
  }
}

// ---- placement of the cooperative solve's sync words.  The workgroups exchange their partial sums as tagged 8-byte granules on a
// handful of cache lines, and on MI355X that round trip depends on which memory channel the lines map to (1.5 us at most addresses,
// 2.3 us at some; five exchanges per solve, four solves per sweep: worth choosing).  The probe replays the solver's exchange as it is
// today (lm_evaluate): kAcc lanes per workgroup publish {32 payload bits, tag} pairs at gran[blk * 64 (+ 32) + lane] of the parity buffer
// and poll the 2 x NB granules of their column — same layout, same loads, same polling loop.
constexpr int kCoop = 4;     // workgroups of the cooperative form (see lm_evaluate): the odometry table (<= 2 304 factors) ...
constexpr int kCoopMap = 6;  // ... and the scan-to-map problems (4-5 000 factors; 4: 4 227, 6: 4 300, 8: 4 298 scans/s on one box)
static_assert(kCoop <= kLmMaxBlocks && kCoopMap <= kLmMaxBlocks, "partial-sum buffer");
__global__ __launch_bounds__(kLmThreads) void k_lm_sync_probe(double* gsync, int iters, double* sink) {
  const int tid = threadIdx.x, blk = blockIdx.x;
  u64* gw = reinterpret_cast<u64*>(gsync);
  double acc = 0.0;
  for (int it = 0; it < iters; it++) {
    __syncthreads();
    if (tid < kAcc) {
      u64* gran = gw + 8 + (size_t)(it & 1) * kLmMaxBlocks * 64;
      const u64 tag = (u64)(unsigned)(it + 1) << 32;
      const u64 bits = (u64)__double_as_longlong((double)(it + blk + tid));
      __hip_atomic_store(&gran[blk * 64 + tid], (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&gran[blk * 64 + 32 + tid], (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      u64 lo[kCoop], hi[kCoop];
      for (int spins = 0; spins < (1 << 16); spins++) {
#pragma unroll
        for (int q = 0; q < kCoop; q++) {
          lo[q] = __hip_atomic_load(&gran[q * 64 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hi[q] = __hip_atomic_load(&gran[q * 64 + 32 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool all = true;
#pragma unroll
        for (int q = 0; q < kCoop; q++) all = all && (lo[q] >> 32 << 32) == tag && (hi[q] >> 32 << 32) == tag;
        if (all) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < kCoop; q++) acc += __longlong_as_double((long long)((hi[q] << 32) | (lo[q] & 0xffffffffull)));
    }
  }
  if (tid < kAcc && blk == 0) sink[tid] = acc;
}

// Times every candidate slot (n_cand slots, stride_bytes apart, each kLmSyncDoubles doubles) and returns their indices fastest
// first.  ~0.5 ms per candidate, once per handle.
int lm_sync_calibrate(hipStream_t st, double* pool, int n_cand, size_t stride_bytes, int* order_out) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
  double* sink = reinterpret_cast<double*>(reinterpret_cast<char*>(pool) + stride_bytes * (size_t)n_cand);  // 256 spare bytes behind the pool
  std::vector<std::pair<float, int>> t((size_t)n_cand);
  bool ok = hipMemsetAsync(pool, 0, stride_bytes * (size_t)n_cand, st) == hipSuccess;
  for (int pass = 0; pass < 2 && ok; pass++)  // pass 0 warms code and TLBs
    for (int c = 0; c < n_cand && ok; c++) {
      double* slot = reinterpret_cast<double*>(reinterpret_cast<char*>(pool) + stride_bytes * (size_t)c);
      ok = hipEventRecord(e0, st) == hipSuccess;
      VL_RAW_LAUNCH(k_lm_sync_probe, dim3(kCoop), dim3(kLmThreads), 0, st, slot, pass ? 192 : 8, sink);
      ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
      float ms = 0.f;
      ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      t[(size_t)c] = std::make_pair(ms, c);
    }
  ok = ok && hipMemsetAsync(pool, 0, stride_bytes * (size_t)n_cand, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (!ok) return -1;
  std::sort(t.begin(), t.end());
  for (int c = 0; c < n_cand; c++) order_out[c] = t[(size_t)c].second;
  if (getenv("VLOAM_CALIB_DEBUG")) {
    fprintf(stderr, "lm_sync_calibrate (us per exchange, slot):");
    for (int c = 0; c < n_cand; c++) fprintf(stderr, " %.2f/%d", 1e3 * t[(size_t)c].first / 192.0, t[(size_t)c].second);
    fprintf(stderr, "\n");
  }
  return 0;
}

void lm_launch(hipStream_t st, Sess se, const FactorTable& F_in, int n_edge_slots, double* d_x, LMRecord* d_rec, int max_iters, double huber_a, bool quat,
               const int* d_enable, ProfHook* ph, LOState* fin_lo, double* fin_traj, hipEvent_t done) {
  const int edge_rows = n_edge_slots >> 6;
  const unsigned Z = (unsigned)se.B;
  // every launch gets a generation of its own (process-wide counter: a sync slot only has to tell its own successive solves apart)
  static std::atomic<unsigned> g_gen{1};
  FactorTable Fg = F_in;
  Fg.gen = g_gen.fetch_add(1);
  // (the patience of its workgroups, FactorTable::spin_limit, was fixed when the handle was created: vloam_create reads VLOAM_LM_SPIN_LIMIT once)
  const FactorTable& F = Fg;
  const bool direct = quat && F.cap == kLmThreads * (kCacheE + kCacheP) && n_edge_slots == kLmThreads * kCacheE;  // the odometry table
  const bool rowmask = quat && !direct && F.rowmask != nullptr && (F.cap >> 6) <= 2 * kLmThreads;   // the fit kernel left row masks: the solve compacts on its own
  // Batches keep the cooperative form because it is faster there: one workgroup per session and solve (VLOAM_BATCH_SINGLE_WG=1) measured
  // +3 % at B = 8 in round 3 and -10 % in round 4 (DESIGN_HISTORY.md).  (The f64 sums of the normal equations are added in workgroup order:
  // whatever the count, a batched session equals the same sequence run alone to round-off, not bit for bit — tests/test_gpu_batch.py.)
  static const int single_wg = getenv("VLOAM_BATCH_SINGLE_WG") ? atoi(getenv("VLOAM_BATCH_SINGLE_WG")) : 0;
  const bool coop = F.gsync != nullptr && !(single_wg && se.B > 1) && !se.no_coop;   // no_coop: a solve of this handle had to degrade once (vloam_sync)
  if (!direct && !rowmask) VLOAM_LAUNCH(ph, kKLmCompact, st, k_lm_compact, dim3(F.cap >> 6, 1, Z), dim3(64), 0, st, F, quat ? 1 : 0, d_enable, se.ss);
  static const int one_xcd = getenv("VLOAM_LM_ONE_XCD") ? atoi(getenv("VLOAM_LM_ONE_XCD")) : 1;   // A/B switch, see lm_coop_block
  const unsigned spread = (one_xcd && se.B == 1 && se.crowd < 2) ? 8u : 1u;   // (crowd: the third and later single-sequence handles of a process stay spread, c_api.cpp)
#define VL_SOLVE(Q, M, N)                                                                                                                   \
  VLOAM_LAUNCH_EV(ph, kKLmSolve, st, done, (k_lm_solve<Q, M, N>), dim3((N) > 1 ? (N) * spread : (N), 1, Z), dim3(kLmThreads), 0, st, F, edge_rows, d_x, \
                  d_rec, max_iters, huber_a, d_enable, fin_lo, fin_traj, se.ss)
  // Workgroups of a cooperative solve.  A single sequence: 8 for both problems — with the exchange inside one XCD (1 350 - 1 900 cycles
  // whatever the count) more compute units shorten the factor loops for free: 6 440 scans/s against 6 250 with 4 / 6 (four alternating
  // runs each, profiles/r04_solver_workgroups_ab.txt).  Batches keep 4 / 6: B = 8 loses 5 % with 8 / 8 (128 compute units pinned by
  // one-wavefront-per-SIMD workgroups).  The partial sums are added in workgroup order, so the count is part of the summation order:
  // a batched session agrees with the same sequence run alone to round-off (1e-13 on the poses), not bit for bit (tests/test_gpu_batch.py).
  if (direct && coop) {
    static const int lo_env = getenv("VLOAM_LM_LO_WGS") ? atoi(getenv("VLOAM_LM_LO_WGS")) : 0;   // A/B override
    const int lo_wgs = lo_env ? lo_env : (se.B == 1 ? 8 : kCoop);
    if (lo_wgs == 8) VL_SOLVE(true, kLmDirect, 8);
    else if (lo_wgs == 6) VL_SOLVE(true, kLmDirect, 6);
    else VL_SOLVE(true, kLmDirect, kCoop);
  }
  else if (direct) VL_SOLVE(true, kLmDirect, 1);
  else if (rowmask && coop) {
    static const int map_env = getenv("VLOAM_LM_MAP_WGS") ? atoi(getenv("VLOAM_LM_MAP_WGS")) : 0;   // A/B override
    const int wgs_env = map_env ? map_env : (se.B == 1 ? 8 : kCoopMap);
    if (wgs_env == 8) VL_SOLVE(true, kLmRowMask, 8);
    else if (wgs_env == 4) VL_SOLVE(true, kLmRowMask, 4);
    else VL_SOLVE(true, kLmRowMask, kCoopMap);
  }
  else if (rowmask) VL_SOLVE(true, kLmRowMask, 1);
  else if (quat && coop) VL_SOLVE(true, kLmPacked, kCoopMap);
  else if (quat) VL_SOLVE(true, kLmPacked, 1);
  else VL_SOLVE(false, kLmPacked, 1);
#undef VL_SOLVE
}

}  // namespace vloam
