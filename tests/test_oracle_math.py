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
