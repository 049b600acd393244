"""CPU: the oracle's coupled VLOAM frame loop (oracle/orc_vloam.py) — tf2 algebra against scipy's Rotation, and the
combined-mode behaviour the reference has (MAIN/src/vloam_main_node.cpp:125-180, laser_odometry.cpp:223-236,563-567,
vloam_tf.cpp:59-75, visual_odometry.cpp:258-281,425-430)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import orc_vloam as ov


def rand_tf(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return ov.TF.from_qt(q, rng.normal(size=3) * 3), q


def test_tf2_restatement_against_scipy():
    rng = np.random.default_rng(5)
    for _ in range(50):
        A, qa = rand_tf(rng)
        B, qb = rand_tf(rng)
        assert np.allclose(A.m, Rotation.from_quat(qa).as_matrix(), atol=1e-14)
        C = A * B
        assert np.allclose(C.m, A.m @ B.m, atol=1e-14) and np.allclose(C.o, A.m @ B.o + A.o, atol=1e-13)
        I = A * A.inverse()
        assert np.allclose(I.m, np.eye(3), atol=1e-13) and np.allclose(I.o, 0, atol=1e-12)
        q = A.rotation()   # Matrix3x3::getRotation: same rotation, unit norm, either sign
        assert abs(np.linalg.norm(q) - 1) < 1e-14 and min(np.linalg.norm(q - qa), np.linalg.norm(q + qa)) < 1e-13
        # non-unit quaternion: setRotation normalises through s = 2 / |q|^2
        assert np.allclose(ov.basis_from_quat(2.5 * qa), A.m, atol=1e-14)
    # the largest-diagonal branches of getRotation (trace <= 0)
    for axis in np.eye(3):
        R = Rotation.from_rotvec(np.pi * 0.98 * axis).as_matrix()
        q = ov.quat_from_basis(R)
        assert np.allclose(ov.basis_from_quat(q), R, atol=1e-13)


def test_axis_angle_helpers_match_tf2_semantics():
    rng = np.random.default_rng(6)
    for _ in range(20):
        aa = rng.normal(size=3) * 0.3
        ang = np.linalg.norm(aa)
        q = ov.quat_axis_angle(aa / ang, ang)
        assert np.allclose(q, Rotation.from_rotvec(aa).as_quat(), atol=1e-14)
        assert np.allclose(ov.quat_get_axis(q) * ov.quat_get_angle(q), aa, atol=1e-12)
    ident = np.array([0.0, 0.0, 0.0, 1.0])
    assert np.array_equal(ov.quat_get_axis(ident), [1, 0, 0]) and ov.quat_get_angle(ident) == 0.0   # -> zero initial guess at count == 1
    with np.errstate(all="ignore"):
        assert np.all(np.isnan(ov.quat_axis_angle(np.zeros(3) / 0.0, 0.0)[:3]))                      # the reference's zero-angle NaN


@pytest.fixture(scope="module")
def coupled_run(synth):
    from vloam_amd import kitti_io  # noqa: F401  (package import path set up by conftest)
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=6)
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
    o = ov.VloamOracle(cam_T_velo, rect0_T_cam, P, base_T_cam0, velo_T_cam0, detach_VO_LO=False, with_mapping=True)
    log = []
    for k in range(5):
        m = synth.synth_matches(seq, k) if k > 0 else (None, None)
        assert o.process(seq.sweep(k), m[0], m[1]) == 0
        rec = dict(prior=o.lo_prior(), vo=o.vo_result, lo=o.lidar.lo_pose(), n_outer=o.lidar.lo_num_outer())
        rec["solves"] = [o.lidar.lo_solve(r) for r in range(rec["n_outer"])]
        log.append(rec)
    return seq, o, log, (base_T_cam0, velo_T_cam0)


def test_combined_mode_overwrites_the_warm_start_in_both_outer_rounds(coupled_run):
    """Quirk A.8-4 (laser_odometry.cpp:223-236): with detach_VO_LO == 0 the result of outer round 0 is discarded —
    round 1 starts from velo_last_VOT_velo_curr again."""
    seq, o, log, _ = coupled_run
    for k in range(1, 5):
        pq, pt = log[k]["prior"]
        assert log[k]["n_outer"] == 2
        for r in range(2):
            s = log[k]["solves"][r]
            assert np.array_equal(s["q_in"], pq) and np.array_equal(s["t_in"], pt)
        assert np.linalg.norm(log[k]["solves"][0]["q_out"] - log[k]["solves"][0]["q_in"]) > 0   # round 0 did move ...
        # ... and the final pose is round 1's answer
        assert np.array_equal(log[k]["lo"][2], log[k]["solves"][1]["q_out"]) and np.array_equal(log[k]["lo"][3], log[k]["solves"][1]["t_out"])


def test_coupling_carries_the_motion_both_ways(coupled_run):
    seq, o, log, (base_T_cam0, velo_T_cam0) = coupled_run
    # frame 0: no VO solve, identity prior; frame 1: VO starts from the identity cam0_curr_LOT_cam0_prev of frame 0 (zero guess)
    assert log[0]["vo"] is None and np.allclose(log[0]["prior"][0], [0, 0, 0, 1]) and np.allclose(log[0]["prior"][1], 0)
    # (B^-1 * I * B through tf2's transpose-inverse is the identity to rounding only, and getAngle() = 2 acos(w) turns a last-bit
    # error in w into ~3e-8 rad: the guess is zero to that level, in the reference too)
    assert np.linalg.norm(log[1]["vo"]["init_angles"]) < 1e-7 and np.linalg.norm(log[1]["vo"]["init_t"]) < 1e-12
    for k in range(1, 5):
        # the VO prior handed to laser odometry is the generator's velodyne motion to VO accuracy
        q_gt, t_gt = seq.gt_relative(k)
        pq, pt = log[k]["prior"]
        assert min(np.linalg.norm(pq - q_gt), np.linalg.norm(pq + q_gt)) < 5e-3 and np.linalg.norm(pt - t_gt) < 0.15
        if k >= 2:  # VO's initial guess = LO's previous frame-to-frame estimate, conjugated into the camera frame and inverted
            B = ov.TF.from_matrix4(base_T_cam0)
            prev = ov.TF.from_qt(log[k - 1]["lo"][2], log[k - 1]["lo"][3])
            want = B.inverse() * prev.inverse() * B
            q = want.rotation()
            assert np.allclose(log[k]["vo"]["init_angles"], ov.quat_get_axis(q) * ov.quat_get_angle(q), atol=1e-15)
            assert np.allclose(log[k]["vo"]["init_t"], want.o, atol=1e-15)
    # world_VOT_base_last accumulates base_last_VOT_base_curr: after 4 frames it is near the generator's world pose (base == velodyne
    # up to the static extrinsics used here)
    q, t = o.vo_world_pose()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(t)) and np.linalg.norm(t) > 2.0


def test_oracle_reproduces_the_coupled_loop_fixture():
    """tests/golden/vloam_64x256_5frames.npz (written by make_golden.py; inputs stored in the fixture) is still what the oracle computes."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vloam_64x256_5frames.npz"))
    o = ov.VloamOracle(g["cam_T_velo"], g["rect0_T_cam"], g["P_rect0"], g["base_T_cam0"], g["velo_T_cam0"], detach_VO_LO=False, with_mapping=True)
    for k in range(5):
        c = np.zeros((g["in_%d" % k].shape[0], 4), dtype=np.float32)
        c[:, :3] = g["in_%d" % k]
        assert o.process(c, g["prev_uv_%d" % k] if k else None, g["curr_uv_%d" % k] if k else None) == 0
        qw, tw, _, _ = o.lidar.lo_pose()
        qm, tm = o.lidar.map_published_pose()
        vq, vt = o.vo_world_pose()
        assert np.allclose(np.concatenate([qw, tw, qm, tm, vq, vt]), g["f%d_poses" % k], rtol=0, atol=1e-12)
        if k:
            assert np.allclose(np.concatenate([o.vo_result["angles"], o.vo_result["t"]]), g["f%d_vo" % k][:6], rtol=0, atol=1e-12)
            assert [o.vo_result["counter32"], o.vo_result["counter22"]] == list(g["f%d_vo" % k][6:].astype(int))
