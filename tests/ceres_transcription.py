"""A SECOND, independent transcription of the trust-region machinery the reference gets from Ceres 2.0 — test infrastructure, CPU only.

Written from the published Ceres 2.0 sources as the author knows them (trust_region_minimizer.cc, levenberg_marquardt_strategy.{h,cc},
trust_region_step_evaluator.cc, dense_qr_solver.cc, corrector.cc, loss_function.cc, local_parameterization.cc, residual_block.cc,
rotation.h, jet.h) and SURVEY.md Appendix A — NOT from oracle/orc_ceres.cpp, which it exists to cross-check
(tests/test_oracle_math.py::test_lm_trace_vs_python_transcription).  Structure and names follow Ceres so that a reader can hold the two
side by side; numpy only.  The cost functors are the reference's (lidar_odometry_mapping/include/lidar_odometry_mapping/lidarFactor.hpp:14-139,
visual_odometry/include/visual_odometry/ceres_cost_function.h:54-96,147-185), differentiated by forward-mode dual numbers like
ceres::AutoDiffCostFunction does, all residual blocks of one type at once (arrays of Jets).

PARITY UNPINNED: Ceres itself is not in this image; two transcriptions agreeing says the restatement was transcribed consistently twice,
not that either equals the library."""
import math

import numpy as np

DBL_MIN = np.finfo(np.float64).tiny
DBL_MAX = np.finfo(np.float64).max
EPS = np.finfo(np.float64).eps


# ---------------------------------------------------------------------------------------------------------------- jet.h
class Jet:
    """ceres::Jet<double, K> for N residual blocks at once: a [N], v [N, K]."""
    __slots__ = ("a", "v")

    def __init__(self, a, v):
        self.a = a
        self.v = v

    @staticmethod
    def const(a, k):
        a = np.asarray(a, dtype=np.float64)
        return Jet(a, np.zeros(a.shape + (k,)))

    def _lift(self, o):
        return o if isinstance(o, Jet) else Jet(np.broadcast_to(np.asarray(o, dtype=np.float64), self.a.shape), np.zeros_like(self.v))

    def __add__(self, o):
        o = self._lift(o)
        return Jet(self.a + o.a, self.v + o.v)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        return Jet(self.a - o.a, self.v - o.v)

    def __rsub__(self, o):
        return self._lift(o) - self

    def __neg__(self):
        return Jet(-self.a, -self.v)

    def __mul__(self, o):
        o = self._lift(o)
        return Jet(self.a * o.a, self.a[..., None] * o.v + self.v * o.a[..., None])

    __rmul__ = __mul__

    def __truediv__(self, o):   # jet.h operator/: g_a_inverse = 1 / g.a; f_a_by_g_a = f.a * g_a_inverse; (f.v - f_a_by_g_a * g.v) * g_a_inverse
        o = self._lift(o)
        gi = 1.0 / o.a
        fg = self.a * gi
        return Jet(fg, (self.v - fg[..., None] * o.v) * gi[..., None])

    def __rtruediv__(self, o):
        return self._lift(o) / self


def jsqrt(f):       # tmp = sqrt(f.a); two_a_inverse = 1 / (2 tmp); Jet(tmp, f.v * two_a_inverse)
    t = np.sqrt(f.a)
    return Jet(t, f.v * (1.0 / (2.0 * t))[..., None])


def jsin(f):
    return Jet(np.sin(f.a), np.cos(f.a)[..., None] * f.v)


def jcos(f):
    return Jet(np.cos(f.a), (-np.sin(f.a))[..., None] * f.v)


def jacos(f):       # -1 / sqrt(1 - a^2)
    return Jet(np.arccos(f.a), (-1.0 / np.sqrt(1.0 - f.a * f.a))[..., None] * f.v)


def jabs(f):        # jet.h abs: f.a < 0 ? -f : f
    s = np.where(f.a < 0.0, -1.0, 1.0)
    return Jet(s * f.a, s[..., None] * f.v)


def jwhere(mask, f, g):
    return Jet(np.where(mask, f.a, g.a), np.where(mask[..., None], f.v, g.v))


# ---------------------------------------------------------------------------------------------------------------- Eigen pieces, on Jets
def cross3(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def quat_transform(q_wxyz, v):
    """Eigen QuaternionBase::_transformVector: uv = vec x v; uv += uv; v + w uv + vec x uv."""
    w, x, y, z = q_wxyz
    u = [x, y, z]
    uv = cross3(u, v)
    uv = [c + c for c in uv]
    uuv = cross3(u, uv)
    return [v[i] + w * uv[i] + uuv[i] for i in range(3)]


def identity_slerp(t, q_wxyz, k):
    """Eigen::Quaternion::slerp of the identity towards q (lidarFactor.hpp:31-33): coefficient-wise scale0 * identity + scale1 * q."""
    w, x, y, z = q_wxyz
    one = 1.0 - EPS
    d = w                       # identity . q
    absd = jabs(d)
    near = absd.a >= one
    # the branch not taken must stay finite for the rows that do take it: evaluate acos on a safe copy
    safe = Jet(np.where(near, 0.5, absd.a), absd.v)
    theta = jacos(safe)
    sin_theta = jsin(theta)
    s0 = jsin((1.0 - t) * theta) / sin_theta
    s1 = jsin(t * theta) / sin_theta
    lin0 = Jet.const(np.full_like(d.a, 1.0 - t), k)
    lin1 = Jet.const(np.full_like(d.a, t), k)
    scale0 = jwhere(near, lin0, s0)
    scale1 = jwhere(near, lin1, s1)
    scale1 = jwhere(d.a < 0.0, -scale1, scale1)
    return [scale0 * 1.0 + scale1 * w, scale1 * x, scale1 * y, scale1 * z]


def angle_axis_rotate_point(aa, pt, k):
    """ceres::AngleAxisRotatePoint (rotation.h)."""
    theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]
    big = theta2.a > EPS
    safe = Jet(np.where(big, theta2.a, 1.0), theta2.v)
    theta = jsqrt(safe)
    costheta, sintheta = jcos(theta), jsin(theta)
    theta_inverse = 1.0 / theta
    w = [aa[i] * theta_inverse for i in range(3)]
    w_cross_pt = cross3(w, pt)
    tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - costheta)
    far = [pt[i] * costheta + w_cross_pt[i] * sintheta + w[i] * tmp for i in range(3)]
    a_cross_pt = cross3(aa, pt)
    near = [pt[i] + a_cross_pt[i] for i in range(3)]
    return [jwhere(big, far[i], near[i]) for i in range(3)]


# ---------------------------------------------------------------------------------------------------------------- the residual blocks
# Factor rows as tests feed them to orc.solve: [type, payload...]
#   0 LidarEdgeFactor       curr(3) a(3) b(3)
#   1 LidarPlaneFactor      curr(3) j(3) l(3) m(3)
#   2 LidarPlaneNormFactor  curr(3) n(3) d
#   3 CostFunctor32         X0(3) x1_bar y1_bar
#   4 CostFunctor22         x0_bar y0_bar x1_bar y1_bar
RESIDUALS_OF_TYPE = {0: 3, 1: 1, 2: 1, 3: 2, 4: 1}


class Problem:
    """ceres::Problem with one shared HuberLoss and either {q (EigenQuaternionParameterization), t} or {angle-axis, t} parameter blocks."""

    def __init__(self, factors, quaternion, huber_a, s=1.0):
        self.quaternion = quaternion
        self.huber_a = huber_a
        self.s = s
        self.types = np.array([int(f[0]) for f in factors])
        width = max(len(f) for f in factors) - 1
        self.payload = np.zeros((len(factors), max(width, 12)))
        for i, f in enumerate(factors):
            self.payload[i, :len(f) - 1] = f[1:]
        self.n_ambient = 7 if quaternion else 6

    # -- ResidualBlock::Evaluate for all blocks of one type: raw residuals [n, r] and ambient Jacobians [n, r, n_ambient]
    def _functor(self, ftype, P, x):
        k = self.n_ambient
        n = P.shape[0]

        def par(i):
            v = np.zeros((n, k))
            v[:, i] = 1.0
            return Jet(np.full(n, x[i]), v)

        def cst(col):
            return Jet.const(P[:, col], k)

        if self.quaternion:
            q = [par(3), par(0), par(1), par(2)]       # Eigen::Quaternion<T>{q[3], q[0], q[1], q[2]}: (w, x, y, z)
            t = [par(4), par(5), par(6)]
            cp = [cst(0), cst(1), cst(2)]
            if ftype in (0, 1):
                qs = identity_slerp(self.s, q, k)
                ts = [self.s * c for c in t]
                lp = quat_transform(qs, cp)
                lp = [lp[i] + ts[i] for i in range(3)]
                if ftype == 0:
                    lpa = [cst(3), cst(4), cst(5)]
                    lpb = [cst(6), cst(7), cst(8)]
                    nu = cross3([lp[i] - lpa[i] for i in range(3)], [lp[i] - lpb[i] for i in range(3)])
                    de = P[:, 3:6] - P[:, 6:9]
                    de_norm = np.sqrt(de[:, 0] * de[:, 0] + de[:, 1] * de[:, 1] + de[:, 2] * de[:, 2])
                    res = [nu[i] / Jet.const(de_norm, k) for i in range(3)]
                else:
                    j, l, m = P[:, 3:6], P[:, 6:9], P[:, 9:12]
                    nrm = np.cross(j - l, j - m)
                    nn = np.sqrt(nrm[:, 0] * nrm[:, 0] + nrm[:, 1] * nrm[:, 1] + nrm[:, 2] * nrm[:, 2])
                    nrm = nrm / nn[:, None]                # ljm_norm.normalize()
                    ljm = [Jet.const(nrm[:, i], k) for i in range(3)]
                    lpj = [cst(3), cst(4), cst(5)]
                    dlt = [lp[i] - lpj[i] for i in range(3)]
                    res = [dlt[0] * ljm[0] + dlt[1] * ljm[1] + dlt[2] * ljm[2]]
            elif ftype == 2:
                pw = quat_transform(q, cp)
                pw = [pw[i] + t[i] for i in range(3)]
                norm = [cst(3), cst(4), cst(5)]
                res = [norm[0] * pw[0] + norm[1] * pw[1] + norm[2] * pw[2] + cst(6)]
            else:
                raise ValueError(ftype)
        else:
            aa = [par(0), par(1), par(2)]
            t = [par(3), par(4), par(5)]
            if ftype == 3:
                X0 = [cst(0), cst(1), cst(2)]
                RX = angle_axis_rotate_point(aa, X0, k)
                RX = [RX[i] + t[i] for i in range(3)]
                res = [RX[0] - RX[2] * cst(3), RX[1] - RX[2] * cst(4)]
            elif ftype == 4:
                X0 = [cst(0), cst(1), Jet.const(np.ones(n), k)]
                X1 = [cst(2), cst(3), Jet.const(np.ones(n), k)]
                RX = angle_axis_rotate_point(aa, X0, k)
                c = cross3(t, RX)
                res = [X1[0] * c[0] + X1[1] * c[1] + X1[2] * c[2]]
            else:
                raise ValueError(ftype)
        r = np.stack([c.a for c in res], axis=1)
        J = np.stack([c.v for c in res], axis=1)
        return r, J

    def plus_jacobian(self, x):
        """LocalParameterization::ComputeJacobian of the whole parameter vector: [n_ambient, 6]."""
        if not self.quaternion:
            return np.eye(6)
        q = x[:4]           # (x, y, z, w)
        Jq = np.array([[q[3], q[2], -q[1]], [-q[2], q[3], q[0]], [q[1], -q[0], q[3]], [-q[0], -q[1], -q[2]]])
        M = np.zeros((7, 6))
        M[:4, :3] = Jq
        M[4:, 3:] = np.eye(3)
        return M

    def plus(self, x, delta):
        """Evaluator::Plus: EigenQuaternionParameterization::Plus on block 0, identity parameterisation on the rest."""
        if not self.quaternion:
            return x + delta
        out = x.copy()
        d = delta[:3]
        norm_delta = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        if norm_delta > 0.0:
            s = math.sin(norm_delta) / norm_delta
            w1, x1, y1, z1 = math.cos(norm_delta), s * d[0], s * d[1], s * d[2]     # delta_q(w, x, y, z)
            x2, y2, z2, w2 = x[0], x[1], x[2], x[3]
            out[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2       # Eigen quaternion product delta_q * x
            out[1] = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
            out[2] = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
            out[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
        out[4:] = x[4:] + delta[3:]
        return out

    def evaluate(self, x, want_jacobian=True):
        """ProgramEvaluator::Evaluate: cost, residuals (after the corrector), Jacobian in the tangent space (after the corrector), gradient."""
        nb = len(self.types)
        blocks_r = [None] * nb
        blocks_J = [None] * nb
        PJ = self.plus_jacobian(x)
        for ftype in sorted(set(self.types.tolist())):
            idx = np.nonzero(self.types == ftype)[0]
            r, J = self._functor(ftype, self.payload[idx], x)
            Jl = J @ PJ                                   # ResidualBlock::Evaluate: global Jacobian x local parameterization Jacobian
            for n_, i in enumerate(idx):
                blocks_r[i] = r[n_].copy()
                blocks_J[i] = Jl[n_].copy()
        cost = 0.0
        a = self.huber_a
        b = a * a
        for i in range(nb):
            r = blocks_r[i]
            squared_norm = float(np.dot(r, r))
            if a > 0.0:
                # HuberLoss::Evaluate
                if squared_norm > b:
                    rr = math.sqrt(squared_norm)
                    rho = (2.0 * a * rr - b, max(DBL_MIN, a / rr), 0.0)
                    rho = (rho[0], rho[1], -rho[1] / (2.0 * squared_norm))
                else:
                    rho = (squared_norm, 1.0, 0.0)
                cost += 0.5 * rho[0]
                # Corrector: rho'' <= 0 for Huber -> residual_scaling = sqrt(rho'), alpha_sq_norm = 0
                sqrt_rho1 = math.sqrt(rho[1])
                assert squared_norm == 0.0 or rho[2] <= 0.0
                blocks_J[i] = blocks_J[i] * sqrt_rho1
                blocks_r[i] = r * sqrt_rho1
            else:
                cost += 0.5 * squared_norm
        residuals = np.concatenate(blocks_r)
        if not want_jacobian:
            return cost, residuals, None, None
        jac = np.concatenate(blocks_J, axis=0)
        gradient = np.zeros(6)
        row = 0
        for i in range(nb):
            nr = blocks_r[i].shape[0]
            gradient += blocks_J[i].T @ blocks_r[i]
            row += nr
        return cost, residuals, jac, gradient


# ---------------------------------------------------------------------------------------------------------------- dense_qr_solver.cc
def dense_qr_solve(A, b, D):
    """min |A y - b|^2 + |D y|^2 by Householder QR of the stacked system [A; diag(D)] y = [b; 0]."""
    m, n = A.shape
    lhs = np.zeros((m + n, n))
    lhs[:m] = A
    lhs[m:] = np.diag(D)
    rhs = np.zeros(m + n)
    rhs[:m] = b
    Q, R = np.linalg.qr(lhs)
    y = np.linalg.solve(R, Q.T @ rhs) if np.all(np.abs(np.diag(R)) > 0) else np.full(n, np.nan)
    return y


# ---------------------------------------------------------------------------------------------------------------- levenberg_marquardt_strategy.cc
class LevenbergMarquardtStrategy:
    def __init__(self, initial_radius=1e4, max_radius=1e16, min_lm_diagonal=1e-6, max_lm_diagonal=1e32):
        self.radius = initial_radius
        self.max_radius = max_radius
        self.min_diagonal = min_lm_diagonal
        self.max_diagonal = max_lm_diagonal
        self.decrease_factor = 2.0
        self.reuse_diagonal = False
        self.diagonal = None

    def compute_step(self, jacobian, residuals):
        """Returns (step or None on LINEAR_SOLVER_FAILURE)."""
        if not self.reuse_diagonal:
            self.diagonal = np.minimum(np.maximum((jacobian * jacobian).sum(axis=0), self.min_diagonal), self.max_diagonal)
        lm_diagonal = np.sqrt(self.diagonal / self.radius)
        # "Instead of solving Jx = -r, solve Jy = r. Then x can be found as x = -y"
        y = dense_qr_solve(jacobian, residuals, lm_diagonal)
        self.reuse_diagonal = True
        if not np.all(np.isfinite(y)):
            return None
        return -y

    def step_accepted(self, step_quality):
        self.radius = self.radius / max(1.0 / 3.0, 1.0 - (2.0 * step_quality - 1.0) ** 3)
        self.radius = min(self.max_radius, self.radius)
        self.decrease_factor = 2.0
        self.reuse_diagonal = False

    def step_rejected(self, step_quality):
        self.radius = self.radius / self.decrease_factor
        self.decrease_factor *= 2.0
        self.reuse_diagonal = True

    def step_is_invalid(self):
        # levenberg_marquardt_strategy.h: "Treat the current step as a rejected step with no increase in solution quality."
        self.step_rejected(0.0)


# ---------------------------------------------------------------------------------------------------------------- trust_region_step_evaluator.cc
class TrustRegionStepEvaluator:
    def __init__(self, initial_cost, max_consecutive_nonmonotonic_steps=0):
        self.max_consecutive_nonmonotonic_steps = max_consecutive_nonmonotonic_steps
        self.minimum_cost = initial_cost
        self.current_cost = initial_cost
        self.reference_cost = initial_cost
        self.candidate_cost = initial_cost
        self.accumulated_reference_model_cost_change = 0.0
        self.accumulated_candidate_model_cost_change = 0.0
        self.num_consecutive_nonmonotonic_steps = 0

    def step_quality(self, cost, model_cost_change):
        # trust_region_step_evaluator.cc: "If the function evaluation for this step was a failure, in which case the
        # TrustRegionMinimizer would have set the cost to std::numeric_limits<double>::max() ... the division by model_cost_change can
        # result in an overflow.  To prevent that from happening, we will deal with this case explicitly."
        if cost >= DBL_MAX:
            return -DBL_MAX
        relative_decrease = (self.current_cost - cost) / model_cost_change
        historical_relative_decrease = (self.reference_cost - cost) / (self.accumulated_reference_model_cost_change + model_cost_change)
        return max(relative_decrease, historical_relative_decrease)

    def step_accepted(self, cost, model_cost_change):
        self.current_cost = cost
        self.accumulated_candidate_model_cost_change += model_cost_change
        self.accumulated_reference_model_cost_change += model_cost_change
        if self.current_cost < self.minimum_cost:
            self.minimum_cost = self.current_cost
            self.num_consecutive_nonmonotonic_steps = 0
            self.candidate_cost = self.current_cost
            self.accumulated_candidate_model_cost_change = 0.0
        else:
            self.num_consecutive_nonmonotonic_steps += 1
            if self.current_cost > self.candidate_cost:
                self.candidate_cost = self.current_cost
                self.accumulated_candidate_model_cost_change = 0.0
        if self.num_consecutive_nonmonotonic_steps == self.max_consecutive_nonmonotonic_steps:
            self.reference_cost = self.candidate_cost
            self.accumulated_reference_model_cost_change = self.accumulated_candidate_model_cost_change


# ---------------------------------------------------------------------------------------------------------------- trust_region_minimizer.cc
NO_CONVERGENCE, CONVERGENCE, FAILURE = 0, 1, 2


class TrustRegionMinimizer:
    def __init__(self, problem, max_num_iterations, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
                 min_relative_decrease=1e-3, min_trust_region_radius=1e-32, max_num_consecutive_invalid_steps=5):
        self.p = problem
        self.max_num_iterations = max_num_iterations
        self.function_tolerance = function_tolerance
        self.gradient_tolerance = gradient_tolerance
        self.parameter_tolerance = parameter_tolerance
        self.min_relative_decrease = min_relative_decrease
        self.min_trust_region_radius = min_trust_region_radius
        self.max_num_consecutive_invalid_steps = max_num_consecutive_invalid_steps

    def _evaluate_gradient_and_jacobian(self):
        self.x_cost, self.residuals, self.jacobian, self.gradient = self.p.evaluate(self.x, True)
        self.it["cost"] = self.x_cost
        if self.it["iteration"] == 0:
            self.jacobian_scaling = 1.0 / (1.0 + np.sqrt((self.jacobian * self.jacobian).sum(axis=0)))
        self.jacobian = self.jacobian * self.jacobian_scaling[None, :]
        projected_gradient_step = self.p.plus(self.x, -self.gradient)
        self.it["gradient_max_norm"] = float(np.max(np.abs(self.x - projected_gradient_step)))

    def minimize(self, x0):
        self.x = np.array(x0, dtype=np.float64)
        self.iterations = []
        self.termination = None
        self.message = ""
        self.num_consecutive_invalid_steps = 0
        self.strategy = LevenbergMarquardtStrategy()
        self.parameters = self.x.copy()
        self.minimum_cost = DBL_MAX
        # ---- IterationZero
        self.it = dict(iteration=0, step_is_valid=False, step_is_successful=False, cost_change=0.0, gradient_max_norm=0.0, step_norm=0.0,
                       relative_decrease=0.0)
        self.x_norm = float(np.linalg.norm(self.x))
        self._evaluate_gradient_and_jacobian()
        self.initial_cost = self.x_cost
        self.it["step_is_valid"] = True
        self.it["step_is_successful"] = True
        self.step_evaluator = TrustRegionStepEvaluator(self.x_cost, 0)
        while self._finalize_iteration_and_check_if_minimizer_can_continue():
            self.it = dict(iteration=self.iterations[-1]["iteration"] + 1, step_is_valid=False, step_is_successful=False, cost=0.0, cost_change=0.0,
                           gradient_max_norm=0.0, step_norm=0.0, relative_decrease=0.0)
            self._compute_trust_region_step()
            if not self.it["step_is_valid"]:
                if not self._handle_invalid_step():
                    break
                continue
            # ---- ComputeCandidatePointAndEvaluateCost
            self.candidate_x = self.p.plus(self.x, self.delta)
            self.candidate_cost = self.p.evaluate(self.candidate_x, False)[0]
            if not math.isfinite(self.candidate_cost):
                self.candidate_cost = DBL_MAX
            if self._parameter_tolerance_reached():
                break
            if self._function_tolerance_reached():
                break
            if self._is_step_successful():
                self._handle_successful_step()
            else:
                self.it["step_is_successful"] = False
                self.it["cost"] = self.candidate_cost
                self.it["gradient_max_norm"] = self.iterations[-1]["gradient_max_norm"]
                self.strategy.step_rejected(self.it["relative_decrease"])
        return dict(x=self.parameters.copy(), iterations=self.iterations, termination=self.termination, message=self.message,
                    initial_cost=self.initial_cost, final_cost=self.minimum_cost)

    def _finalize_iteration_and_check_if_minimizer_can_continue(self):
        if self.it["step_is_successful"]:
            if self.x_cost < self.minimum_cost:
                self.minimum_cost = self.x_cost
                self.parameters = self.x.copy()
        self.it["trust_region_radius"] = self.strategy.radius
        self.iterations.append(dict(self.it))
        if self.it["iteration"] >= self.max_num_iterations:
            self.termination, self.message = NO_CONVERGENCE, "Maximum number of iterations reached."
            return False
        if self.it["gradient_max_norm"] <= self.gradient_tolerance:
            self.termination, self.message = CONVERGENCE, "Gradient tolerance reached."
            return False
        if self.it["trust_region_radius"] <= self.min_trust_region_radius:
            self.termination, self.message = CONVERGENCE, "Minimum trust region radius reached."
            return False
        return True

    def _compute_trust_region_step(self):
        self.it["step_is_valid"] = False
        step = self.strategy.compute_step(self.jacobian, self.residuals)
        if step is None:
            return
        self.trust_region_step = step
        # model_cost_change = -(J step)' (f + J step / 2)
        model_residuals = self.jacobian @ step
        self.model_cost_change = -float(np.dot(model_residuals, self.residuals + model_residuals / 2.0))
        self.it["step_is_valid"] = self.model_cost_change > 0.0
        if self.it["step_is_valid"]:
            self.delta = step * self.jacobian_scaling      # undo the Jacobian column scaling
            self.num_consecutive_invalid_steps = 0

    def _handle_invalid_step(self):
        self.num_consecutive_invalid_steps += 1
        if self.num_consecutive_invalid_steps >= self.max_num_consecutive_invalid_steps:
            self.termination, self.message = FAILURE, "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps"
            return False
        self.strategy.step_is_invalid()
        self.it["cost"] = self.x_cost
        self.it["cost_change"] = 0.0
        self.it["gradient_max_norm"] = self.iterations[-1]["gradient_max_norm"]
        self.it["step_norm"] = 0.0
        self.it["relative_decrease"] = 0.0
        return True

    def _parameter_tolerance_reached(self):
        self.it["step_norm"] = float(np.linalg.norm(self.x - self.candidate_x))
        step_size_tolerance = self.parameter_tolerance * (self.x_norm + self.parameter_tolerance)
        if self.it["step_norm"] > step_size_tolerance:
            return False
        self.termination, self.message = CONVERGENCE, "Parameter tolerance reached."
        return True

    def _function_tolerance_reached(self):
        self.it["cost_change"] = self.x_cost - self.candidate_cost
        absolute_function_tolerance = self.function_tolerance * self.x_cost
        if abs(self.it["cost_change"]) > absolute_function_tolerance:
            return False
        self.termination, self.message = CONVERGENCE, "Function tolerance reached."
        return True

    def _is_step_successful(self):
        self.it["relative_decrease"] = self.step_evaluator.step_quality(self.candidate_cost, self.model_cost_change)
        return self.it["relative_decrease"] > self.min_relative_decrease

    def _handle_successful_step(self):
        self.x = self.candidate_x
        self.x_norm = float(np.linalg.norm(self.x))
        self._evaluate_gradient_and_jacobian()
        self.it["step_is_successful"] = True
        self.strategy.step_accepted(self.it["relative_decrease"])
        self.step_evaluator.step_accepted(self.candidate_cost, self.model_cost_change)


def solve(factors, p0, p1, quaternion=True, huber_a=0.1, max_iters=4):
    """Same call shape as orc.solve: returns the trace as rows {cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius,
    step_is_valid, step_is_successful}, the termination type and the parameters."""
    prob = Problem(factors, quaternion, huber_a)
    x0 = np.concatenate([np.asarray(p0, dtype=np.float64), np.asarray(p1, dtype=np.float64)])
    r = TrustRegionMinimizer(prob, max_iters).minimize(x0)
    trace = np.array([[it["cost"], it["cost_change"], it["gradient_max_norm"], it["step_norm"], it["relative_decrease"], it["trust_region_radius"],
                       float(it["step_is_valid"]), float(it["step_is_successful"])] for it in r["iterations"]])
    n0 = len(p0)
    return dict(p0=r["x"][:n0], p1=r["x"][n0:], trace=trace, termination=r["termination"], message=r["message"], initial_cost=r["initial_cost"],
                final_cost=r["final_cost"])
