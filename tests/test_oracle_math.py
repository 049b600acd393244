"""CPU: the oracle's restatement of the third-party arithmetic (Ceres 2.0 LM, PCL VoxelGrid, exact kNN, autodiff
functors) checked against known answers and independent numpy / scipy implementations.
PARITY UNPINNED with respect to the real libraries — none of them exists in this image."""
import numpy as np
import pytest


def test_lm_hello_world_known_answer(orc):
    """Ceres' own trace quoted in the reference (src/visual_odometry/README.md:40-50): f(x) = 10 - x from x = 0.5.
    iter cost cost_change |gradient| |step| tr_ratio tr_radius:
      0 4.512500e+01 0 9.50e+00 0 0 1.00e+04 / 1 4.511598e-07 4.51e+01 9.50e-04 9.50e+00 1.00e+00 3.00e+04 /
      2 5.012552e-16 4.51e-07 3.17e-08 9.50e-04 1.00e+00 9.00e+04, termination CONVERGENCE, x: 0.5 -> 10."""
    r = orc.solve([[5, 10.0]], [0.5, 0, 0], [0, 0, 0], quaternion=False, huber_a=0.0, max_iters=50)
    tr = r["trace"]
    assert tr.shape[0] == 3 and r["termination"] == 1
    for got, want in zip(tr[:, 0], [4.512500e+01, 4.511598e-07, 5.012552e-16]):
        assert abs(got - want) <= 5e-7 * want
    assert np.allclose(tr[:, 5], [1e4, 3e4, 9e4], rtol=1e-12)               # trust-region radius schedule
    assert np.allclose(tr[:, 2], [9.50, 9.50e-04, 3.17e-08], rtol=2e-3)     # |gradient|
    assert np.allclose(tr[1:, 3], [9.50, 9.50e-04], rtol=2e-3)              # |step|
    assert np.allclose(tr[1:, 4], [1.0, 1.0], atol=1e-4)                    # tr_ratio
    assert abs(r["p0"][0] - 10.0) < 1e-6


def quat_plus(q, d):
    n = np.linalg.norm(d)
    if n == 0:
        return q.copy()
    dq = np.concatenate([np.sin(n) / n * d, [np.cos(n)]])
    x1, y1, z1, w1 = dq
    x2, y2, z2, w2 = q
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])


@pytest.mark.parametrize("ftype,ngeom", [(0, 6), (1, 9), (2, 4)])
def test_factor_jacobians_vs_finite_differences(orc, ftype, ngeom):
    """Autodiff (Jets through the restated functors, lidarFactor.hpp:14-139) x EigenQuaternionParameterization
    Jacobian == central differences along Plus(x, delta)."""
    rng = np.random.default_rng(ftype)
    for _ in range(20):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = rng.normal(size=3)
        curr = rng.normal(size=3) * 10
        geom = rng.normal(size=ngeom) * 5
        if ftype == 2:
            geom[:3] /= np.linalg.norm(geom[:3])
        r0, J = orc.eval_lidar_factor(ftype, curr, geom, q, t)
        h = 1e-6
        for a in range(6):
            d = np.zeros(6); d[a] = h
            rp, _ = orc.eval_lidar_factor(ftype, curr, geom, quat_plus(q, d[:3]), t + d[3:])
            rm, _ = orc.eval_lidar_factor(ftype, curr, geom, quat_plus(q, -d[:3]), t - d[3:])
            fd = (rp - rm) / (2 * h)
            assert np.allclose(J[:, a], fd, rtol=1e-5, atol=1e-5 * (1 + np.abs(J).max()))
        # closed form used on the device: d lp / d delta = -2 [R p]x, d lp / d t = I  
        if ftype == 2:
            x, y, z, w = q
            u = np.array([x, y, z])
            uv = 2 * np.cross(u, curr)
            rp_ = curr + w * uv + np.cross(u, uv)
            n = geom[:3]
            assert np.allclose(J[0, :3], -2 * np.cross(n, rp_), rtol=1e-9, atol=1e-9)
            assert np.allclose(J[0, 3:], n, rtol=1e-12)


def test_lm_recovers_known_se3(orc):
    """Noise-free point-to-plane + point-to-line constraints generated from a known transform: LM run to convergence
    must return it (known-answer test for Plus / Jacobian layout / step acceptance)."""
    rng = np.random.default_rng(7)
    ang = 0.05 * rng.normal(size=3)
    th = np.linalg.norm(ang)
    q_true = np.concatenate([np.sin(th / 2) * ang / th, [np.cos(th / 2)]])
    t_true = np.array([0.8, -0.1, 0.05])
    x, y, z, w = q_true
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    rows = []
    for _ in range(300):
        p = rng.normal(size=3) * 10
        lp = R @ p + t_true
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        rows.append([2, *p, *n, -float(n @ lp)])
    for _ in range(100):
        p = rng.normal(size=3) * 10
        lp = R @ p + t_true
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        rows.append([0, *p, *(lp + 0.1 * d), *(lp - 0.1 * d)])
    r = orc.solve(rows, [0, 0, 0, 1], [0, 0, 0], quaternion=True, huber_a=0.1, max_iters=50)
    # Ceres semantics: the run stops on function_tolerance and the last (tiny) step is not applied
    assert min(np.linalg.norm(r["p0"] - q_true), np.linalg.norm(r["p0"] + q_true)) < 1e-7
    assert np.linalg.norm(r["p1"] - t_true) < 1e-6
    assert r["final_cost"] < 1e-12 and r["termination"] == 1
    # iteration cap (the reference runs max_num_iterations = 4, laser_odometry.cpp:460): rows 0..cap, NO_CONVERGENCE
    r1 = orc.solve(rows, [0, 0, 0, 1], [0, 0, 0], quaternion=True, huber_a=0.1, max_iters=1)
    assert r1["trace"].shape[0] == 2 and r1["termination"] == 0
    assert r1["trace"][1, 0] < r1["trace"][0, 0] and r1["trace"][1, 7] == 1


def test_lm_minimiser_agrees_with_scipy_on_a_robust_problem(orc):
    """Independent pin of the restated solver + HuberLoss(0.1): on noisy point-to-plane constraints with gross outliers the
    oracle's LM, run to convergence, must reach the same minimiser of the same robust cost as scipy's trust-region
    least_squares(loss='huber', f_scale=0.1) — a different algorithm, parameterisation (rotation vector) and code base.
    (Plane factors only: Ceres robustifies per residual BLOCK, scipy per component; they coincide for 1-row blocks.)"""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(11)
    rv_true = np.array([0.02, -0.03, 0.05])
    t_true = np.array([0.6, -0.2, 0.1])
    R = Rotation.from_rotvec(rv_true).as_matrix()
    P, N, D = [], [], []
    for k in range(400):
        p = rng.normal(size=3) * 8
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        noise = rng.normal() * 0.02 + (rng.normal() * 1.5 if k % 9 == 0 else 0.0)  # every 9th constraint is an outlier
        P.append(p); N.append(n); D.append(-float(n @ (R @ p + t_true)) + noise)
    P, N, D = np.array(P), np.array(N), np.array(D)
    rows = [[2, *P[k], *N[k], D[k]] for k in range(len(P))]
    r = orc.solve(rows, [0, 0, 0, 1], [0, 0, 0], quaternion=True, huber_a=0.1, max_iters=200)

    def resid(x):
        return np.einsum("ij,ij->i", N, Rotation.from_rotvec(x[:3]).apply(P) + x[3:]) + D

    sol = least_squares(resid, np.zeros(6), loss="huber", f_scale=0.1, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
    q = Rotation.from_rotvec(sol.x[:3]).as_quat()  # x, y, z, w
    # the restated solver stops like Ceres does, on function_tolerance = 1e-6 (relative cost change), scipy runs to 1e-15:
    # the two minima agree to what that tolerance leaves open
    assert r["termination"] == 1
    assert min(np.linalg.norm(r["p0"] - q), np.linalg.norm(r["p0"] + q)) < 1e-5
    assert np.linalg.norm(r["p1"] - sol.x[3:]) < 1e-4
    # same objective: Ceres reports sum rho(r^2) / 2, scipy 0.5 * sum rho(f^2); scipy's (converged) value is the lower one
    assert 0.0 <= r["final_cost"] - sol.cost < 1e-6 * sol.cost
    # and the robust fit is close to the truth although 11 % of the constraints are gross outliers
    assert np.linalg.norm(r["p1"] - t_true) < 0.02


def numpy_voxel_grid(pts, leaf):
    """Independent restatement of pcl::VoxelGrid (voxel_grid.hpp): floor(p * (1/leaf)) cells, output by linearised index."""
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64)
    mn = ijk.min(axis=0)
    div = ijk.max(axis=0) - mn + 1
    idx = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    start = 0
    s_idx = idx[order]
    while start < len(order):
        end = start
        acc = np.zeros(4, dtype=np.float32)
        while end < len(order) and s_idx[end] == s_idx[start]:
            acc = (acc + pts[order[end]]).astype(np.float32)
            end += 1
        out.append(acc / np.float32(end - start))
        start = end
    return np.array(out, dtype=np.float32)


@pytest.mark.parametrize("leaf", [0.2, 0.4, 0.8])
def test_voxel_grid_vs_numpy(orc, leaf):
    rng = np.random.default_rng(3)
    pts = (rng.normal(size=(5000, 4)) * [8, 8, 1.5, 10]).astype(np.float32)
    got = orc.voxel_grid(pts, leaf)
    want = numpy_voxel_grid(pts, leaf)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert orc.voxel_grid(np.zeros((0, 4), np.float32), leaf).shape[0] == 0
    one = orc.voxel_grid(pts[:1], leaf)
    assert np.array_equal(one, pts[:1])  # single point: centroid / 1 is exact


def test_voxel_grid_std_sort_variant_agrees(orc):
    """PCL calls std::sort (unstable): the f32 sum order inside a voxel is implementation-defined.  The canonical
    (stable) oracle and the literally-std::sort build must agree to f32 rounding."""
    rng = np.random.default_rng(5)
    pts = (rng.normal(size=(20000, 4)) * [5, 5, 1, 10]).astype(np.float32)
    a = orc.voxel_grid(pts, 0.8)
    b = orc.voxel_grid(pts, 0.8, variant="liborc_stdsort.so")
    assert a.shape == b.shape and np.max(np.abs(a - b)) < 2e-5


def test_kdtree_exact_vs_brute_force_and_scipy(orc):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    pts = (rng.normal(size=(30000, 4)) * [20, 20, 2, 1]).astype(np.float32)
    q = (rng.normal(size=(500, 3)) * [20, 20, 2]).astype(np.float32)
    for k in (1, 5):
        it, dt = orc.knn(pts, q, k, use_tree=True)
        ib, db = orc.knn(pts, q, k, use_tree=False)
        assert np.array_equal(it, ib) and np.array_equal(dt.view(np.uint32), db.view(np.uint32))
        ds, isc = cKDTree(pts[:, :3].astype(np.float64)).query(q.astype(np.float64), k=k)
        isc = isc.reshape(len(q), k)
        assert np.mean(it == isc) > 0.999  # f32 vs f64 metric may swap near-ties
        assert np.allclose(np.sqrt(dt), ds.reshape(len(q), k), rtol=1e-5, atol=1e-5)
    # ties: duplicated points -> lowest index wins
    dup = np.concatenate([pts[:100], pts[:100]])
    it, _ = orc.knn(dup, pts[:100, :3], 1, use_tree=True)
    assert np.array_equal(it[:, 0], np.arange(100))


# ---------------------------------------------------------------------------------------------------------------- second transcription of the LM loop
def _rand_rot(rng, angle):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    h = 0.5 * angle
    return np.concatenate([np.sin(h) * ax, [np.cos(h)]])      # (x, y, z, w)


def _qrot(q, v):
    u = q[:3]
    uv = 2.0 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


def _random_se3_problem(rng, case):
    """Edge + plane (+ plane-norm) factors around a true pose.  Eight classes: ordinary scan-matching problems (noise, 5-30 % gross outliers:
    the Huber corrector is active), noise-free ones (they end on the gradient / parameter tolerances), outlier-heavy ones, and THIN ones — a
    handful of factors with long lever arms started ~pi away — which are what makes Levenberg-Marquardt steps get rejected."""
    cls = case % 8
    # far_rot, far_t, noise, outlier share, outlier size, edges, planes, plane-norms, spread
    cfg = [(rng.uniform(0.01, 0.6), 1.0, 2e-2, rng.uniform(0.05, 0.30), rng.uniform(0.5, 3.0), int(rng.integers(6, 30)), int(rng.integers(10, 50)), int(rng.integers(0, 20)), 20.0),
           (rng.uniform(0.5, 1.6), 5.0, 5e-2, rng.uniform(0.05, 0.50), rng.uniform(0.5, 30.0), int(rng.integers(6, 30)), int(rng.integers(10, 50)), int(rng.integers(0, 20)), 20.0),
           (0.3, 1.0, 0.0, 0.0, 0.0, 10, 20, 5, 20.0),
           (3.0, 40.0, 0.0, 0.0, 0.0, 6, 12, 0, 20.0),
           (3.1, 50.0, 0.0, 0.0, 0.0, 2, 3, 0, 100.0),
           (3.1, 50.0, 0.02, 0.1, 2.0, 2, 3, 0, 100.0),
           (3.1, 200.0, 0.01, 0.0, 0.0, 0, 8, 0, 100.0),
           (3.1, 100.0, 0.05, 0.2, 5.0, 2, 4, 0, 100.0)][cls]
    far_rot, far_t, noise, outlier_share, outlier_size, n_edge, n_plane, n_pn, spread = cfg
    qt, tt = _rand_rot(rng, rng.uniform(0.0, 0.3)), rng.normal(size=3) * 0.5
    rows = []
    for k in range(n_edge + n_plane + n_pn):
        p = rng.uniform(-spread, spread, size=3)
        lp = _qrot(qt, p) + tt + rng.normal(size=3) * noise
        if rng.uniform() < outlier_share:
            lp = lp + rng.normal(size=3) * outlier_size
        if k < n_edge:
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            rows.append([0, *p, *(lp + 0.5 * u), *(lp - 0.7 * u)])
        elif k < n_edge + n_plane:
            n = rng.normal(size=3); n /= np.linalg.norm(n)
            e1 = np.cross(n, [1.0, 0.3, -0.2]); e1 /= np.linalg.norm(e1)
            e2 = np.cross(n, e1)
            rows.append([1, *p, *(lp + 0.4 * e1), *(lp - 0.5 * e1 + 0.6 * e2), *(lp - 0.3 * e1 - 0.7 * e2)])
        else:
            n = rng.normal(size=3); n /= np.linalg.norm(n)
            rows.append([2, *p, *n, -float(np.dot(n, lp))])
    q0 = _rand_rot(rng, far_rot)
    # compose the start from the truth so that "far" means far from the minimiser
    x1, y1, z1, w1 = q0; x2, y2, z2, w2 = qt
    qs = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
                   w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    ts = tt + rng.normal(size=3) * far_t
    return rows, qs, ts


def _random_vo_problem(rng, case):
    """CostFunctor32 / CostFunctor22 blocks (angle-axis + translation, identity parameterisation)."""
    n32, n22 = int(rng.integers(8, 40)), int(rng.integers(0, 25))
    noise = [0.0, 1e-4, 2e-3][case % 3]
    w = rng.normal(size=3) * 0.05
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
    t = rng.normal(size=3) * 0.3
    rows = []
    for k in range(n32 + n22):
        X0 = np.array([rng.uniform(-8, 8), rng.uniform(-3, 3), rng.uniform(4, 40)])
        X1 = R @ X0 + t
        obs = X1[:2] / X1[2] + rng.normal(size=2) * noise
        if rng.uniform() < 0.15:
            obs = obs + rng.normal(size=2) * 0.2
        if k < n32:
            rows.append([3, *X0, *obs])
        else:
            rows.append([4, X0[0] / X0[2], X0[1] / X0[2], *obs])
    far = [0.0, 0.05, 0.4][(case // 3) % 3]
    return rows, w + rng.normal(size=3) * far * 0.3, t + rng.normal(size=3) * far * 2.0


def test_lm_trace_vs_python_transcription(orc):
    """The trust-region loop twice: oracle/orc_ceres.cpp (C++, DENSE_QR on the stacked Jacobian, Jets) against tests/ceres_transcription.py — a
    literal Python transcription of Ceres 2.0's TrustRegionMinimizer + LevenbergMarquardtStrategy + TrustRegionStepEvaluator + Corrector +
    EigenQuaternionParameterization written from the published sources, not from the C++ file.  288 randomised problems (SE3 with edge /
    plane / plane-norm factors and angle-axis problems with the visual-odometry functors; 5-30 % gross outliers so that the Huber corrector is
    active; starts far enough away that steps get REJECTED; max_num_iterations 4 like the LiDAR solves and 100 like the visual odometry):
    the whole iteration table — cost, cost change, gradient max-norm, step norm, relative decrease, radius, valid / successful flags — and
    the termination must agree: flags and termination exactly, the numbers to 1e-11 relative over the first five rows (all a 4-iteration
    solve produces) on the well-posed classes, see the tolerance comments below for longer runs and for the deliberately ill-conditioned
    classes that provoke the rejections; parameters to 1e-10.  The case mix is asserted: >= 20 % contain a
    rejected step, every termination rule fires at least once."""
    import ceres_transcription as ct
    rng = np.random.default_rng(20260927)
    n_rejected, seen, n_cases = 0, set(), 0
    worst = 0.0
    for case in range(288):
        quaternion = case % 6 != 5
        max_iters = 4 if (case // 8) % 2 == 0 else 100
        if quaternion and case % 8 >= 4 and max_iters == 100:
            max_iters = 10   # (the ill-conditioned classes: two round-off-different runs eventually take a different accept / reject decision)
        if quaternion:
            rows, p0, p1 = _random_se3_problem(rng, case)
        else:
            rows, p0, p1 = _random_vo_problem(rng, case)
        a = orc.solve(rows, p0, p1, quaternion=quaternion, huber_a=0.1, max_iters=max_iters)
        b = ct.solve(rows, p0, p1, quaternion=quaternion, huber_a=0.1, max_iters=max_iters)
        ta, tb = a["trace"], b["trace"]
        assert ta.shape == tb.shape, (case, ta.shape, tb.shape, a["termination"], b["message"])
        assert a["termination"] == b["termination"], (case, a["termination"], b["message"])
        assert np.array_equal(ta[:, 6:8], tb[:, 6:8]), (case, ta[:, 6:8], tb[:, 6:8])
        # per column: relative 1e-11 for the rows a 4-iteration solve can reach (every LiDAR solve of the reference; a quadratically converging
        # noise-free problem turns one ulp of x into 1e-12 of its cost, so 1e-12 itself is not reachable by two correct programs); two round-off-level
        # different trajectories of a THIN problem drift apart once steps get rejected (each accept / reject decision is a discontinuity), so
        # rows beyond the fifth may differ 4x more per row, never more than 1e-6 — the flags and the termination must still be identical.
        # Absolute floors for quantities that are DIFFERENCES of O(1) numbers in both implementations: a cost at round-off level (noise-free
        # problems converge to ~1e-25), cost change = cost - cost, gradient max-norm and step norm = |x - Plus(x, .)| (one ulp of |x| is
        # 1e-16 whatever the result's size; the gradient itself is a sum of ~100 terms J r of mixed sign), relative decrease = cost change /
        # model cost change.
        nrow = tb.shape[0]
        rel = np.minimum(1e-11 * 4.0 ** np.maximum(0, np.arange(nrow) - 4), 1e-6)[:, None]
        if quaternion and case % 8 >= 4:   # the THIN classes: 5 - 8 factors, 100 m lever arms, started ~pi away — every step amplifies round-off ~100x (two Householder QRs of an ill-conditioned 15 x 6 system agree to 1e-16 x condition number)
            rel = np.minimum(1e-12 * 100.0 ** np.arange(nrow), 1e-6)[:, None]
        pm = max(abs(v) for r in rows for v in r[1:])
        xmag = max(1.0, float(np.max(np.abs(np.concatenate([p0, p1])))))
        cost_floor = len(rows) * (2e-13 * (1.0 + pm + xmag)) ** 2
        cost = np.abs(tb[:, 0])
        prev = np.maximum(cost, np.abs(np.roll(tb[:, 0], 1)))
        d = np.abs(ta[:, :6] - tb[:, :6])
        tol = rel * np.abs(tb[:, :6])
        # one ulp of x moves a cost 0.5 sum r^2 by |r| |J| |x| 1e-16: dominant once a noise-free problem converges quadratically
        amp = np.sqrt(2.0 * cost * len(rows)) * 2.0 * (1.0 + pm) * 1e-16 * (1.0 + pm + xmag) * (rel[:, 0] / 1e-11)
        amp_prev = np.maximum(amp, np.roll(amp, 1))
        tol[:, 0] += cost_floor + amp
        tol[:, 1] += 4e-15 * prev + cost_floor + rel[:, 0] * prev + amp_prev
        tol[:, 2] += 4e-15 * xmag + 2e-13 * tb[0, 2] + len(rows) * 2.0 * (1.0 + pm) * 4e-15 * (1.0 + pm + xmag)   # (last term: J^T r with every r at round-off level)
        tol[:, 3] += 4e-15 * xmag
        mcc = np.abs(tb[:, 1]) / np.maximum(np.abs(tb[:, 4]), 1e-300)       # model cost change of the row
        tol[:, 4] += (8e-15 * prev + cost_floor + rel[:, 0] * prev + amp_prev) / np.maximum(mcc, 1e-300)
        excess = d / np.maximum(tol, 1e-300)
        worst = max(worst, float(excess.max()))
        assert excess.max() <= 1.0, (case, b["message"], float(excess.max()), np.unravel_index(excess.argmax(), excess.shape), ta, tb)
        assert np.allclose(np.concatenate([a["p0"], a["p1"]]), np.concatenate([b["p0"], b["p1"]]), rtol=0, atol=1e-10 if nrow <= 5 and not (quaternion and case % 8 >= 4) else 1e-6 * xmag), case
        n_cases += 1
        n_rejected += int(np.any(tb[1:, 7] == 0.0))
        seen.add(b["message"])
    assert n_rejected >= 0.2 * n_cases, (n_rejected, n_cases)
    for msg in ("Maximum number of iterations reached.", "Gradient tolerance reached.", "Parameter tolerance reached.", "Function tolerance reached."):
        assert msg in seen, (msg, seen)
