"""-m gpu: scan-to-scan laserOdometry through the C ABI vs the CPU oracle.

Reference: src/lidar_odometry_mapping/src/laser_odometry.cpp:187-536 + lidarFactor.hpp:14-106 +
the Ceres 2.0 trust-region loop (oracle/orc_ceres.cpp).  Bars (north_star): SE3 pose within 1e-4,
per-point residuals within 1e-6; asserted here two to four orders tighter.  Correspondence indices
(exact 1-NN + adjacent-ring walks) must be identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-8       # |dt| [m] and |dq| (north_star bar: 1e-4)
RESID_TOL = 1e-9      # per-point residuals (north_star bar: 1e-6)


def qdist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def compare_outer(d, o, outer):
    oc, op = o.lo_corr(outer)
    assert np.array_equal(d["corner"], oc), "corner correspondences differ (outer %d)" % outer
    assert np.array_equal(d["plane"], op), "plane correspondences differ (outer %d)" % outer
    s = o.lo_solve(outer)
    rec = d["rec"]
    assert rec["n_factors"] == oc.shape[0] + op.shape[0]
    assert qdist(rec["x_in"][:4], s["q_in"]) < 1e-12 and np.linalg.norm(rec["x_in"][4:] - s["t_in"]) < 1e-12
    # raw residuals at the initial point, factor order = corner then plane (the oracle's AddResidualBlock order)
    r_dev = np.concatenate([d["resid"][:, d["corner_slots"]].T.reshape(-1), d["resid"][0, d["plane_slots"]]])
    assert r_dev.shape == s["residuals0"].shape
    assert np.max(np.abs(r_dev - s["residuals0"]), initial=0) < RESID_TOL
    # closed-form tangent-space Jacobians vs Ceres-style autodiff: J^T J and J^T r at the initial point
    scale = np.sqrt(np.outer(np.diag(s["H0"]), np.diag(s["H0"]))) + 1e-30
    assert np.max(np.abs(rec["H0"] - s["H0"]) / scale) < 1e-9
    assert np.max(np.abs(rec["g0"] - s["g0"])) < 1e-9 * (1 + np.max(np.abs(s["g0"])))
    assert abs(rec["initial_cost"] - s["initial_cost"]) < 1e-10 * (1 + s["initial_cost"])
    # trust-region trace: same number of iterations, same accept/reject pattern, same radius schedule
    assert rec["trace"].shape == s["trace"].shape, (rec["trace"][:, 0], s["trace"][:, 0])
    assert np.array_equal(rec["trace"][:, 6:8], s["trace"][:, 6:8])
    assert np.allclose(rec["trace"][:, 0], s["trace"][:, 0], rtol=1e-8, atol=1e-12)
    assert np.allclose(rec["trace"][:, 5], s["trace"][:, 5], rtol=1e-6)
    assert rec["termination"] == s["termination"]
    assert qdist(rec["x_out"][:4], s["q_out"]) < POSE_TOL and np.linalg.norm(rec["x_out"][4:] - s["t_out"]) < POSE_TOL


@pytest.mark.parametrize("shape,nframes", [((64, 512), 20), ((64, 2048), 4)])
def test_laser_odometry_parity(vl, orc, sweeps, shape, nframes):
    h = vl.Handle(0, scan_line=shape[0], debug=1, with_mapping=0)
    o = orc.Oracle(scan_line=shape[0], with_mapping=False)
    for k in range(nframes):
        cloud = sweeps(shape[0], shape[1], k)
        h.reset_frame()
        h.scan_registration(cloud)
        qw, tw, ql, tl = h.laser_odometry()
        assert o.process(cloud) == 0
        oqw, otw, oql, otl = o.lo_pose()
        if k == 0:
            assert o.lo_num_outer() == 0
            assert np.array_equal(ql, [0, 0, 0, 1]) and np.array_equal(tl, [0, 0, 0])
        else:
            assert o.lo_num_outer() == 2
            for outer in range(2):
                compare_outer(h.lo_debug(outer), o, outer)
        assert qdist(ql, oql) < POSE_TOL and np.linalg.norm(tl - otl) < POSE_TOL, "frame %d f2f pose" % k
        assert qdist(qw, oqw) < POSE_TOL * (k + 1) and np.linalg.norm(tw - otw) < POSE_TOL * (k + 1), "frame %d world pose" % k
        # laserCloudCornerLast / laserCloudSurfLast handed to mapping (laser_odometry.cpp:610-629)
        for which in (5, 6):
            dv, rf = h.features(which), o.cloud(which)
            assert dv.shape == rf.shape and np.array_equal(dv[:, :4].view(np.uint32), rf[:, :4].view(np.uint32))


def test_async_path_matches_stagewise(vl, sweeps):
    """vloam_process_scan (no host sync between stages) == stage-wise calls; replay is bit-reproducible."""
    clouds = [sweeps(64, 512, k) for k in range(5)]
    h1 = vl.Handle(0, with_mapping=0)
    for c in clouds:
        h1.reset_frame(); h1.scan_registration(c); h1.laser_odometry()
    h2 = vl.Handle(0, with_mapping=0)
    for c in clouds:
        h2.process_scan(c)
    h2.sync()
    h3 = vl.Handle(0, with_mapping=0)
    for c in clouds:
        h3.process_scan(c)
    h3.sync()
    t1, t2, t3 = h1.trajectory(), h2.trajectory(), h3.trajectory()
    assert t1.shape == (5, 14)
    assert np.array_equal(t1, t2) and np.array_equal(t2, t3)


def test_lo_recovers_ground_truth_motion(vl, synth):
    """Known-answer: noise-free sweeps, the estimated frame-to-frame motion approaches the generator's SE3."""
    seq = synth.SynthSequence(n_rings=64, n_azimuth=1024, n_sweeps=8, noise_sigma=0.0)
    h = vl.Handle(0, with_mapping=0)
    errs = []
    for k in range(8):
        h.reset_frame(); h.scan_registration(seq.sweep(k))
        _, _, ql, tl = h.laser_odometry()
        if k >= 3:
            gq, gt = seq.gt_relative(k)
            errs.append((np.linalg.norm(tl - gt), qdist(ql, gq)))
    assert max(e[0] for e in errs) < 0.05 and max(e[1] for e in errs) < 2e-3, errs
