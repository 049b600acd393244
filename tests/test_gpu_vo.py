"""-m gpu: depth-enhanced VO residual stack through the C ABI vs the CPU oracle (config 4).

Reference: src/visual_odometry/src/point_cloud_util.cpp:148-174,205-260,302-387 and
src/visual_odometry/src/visual_odometry.cpp:254-450 (+ ceres_cost_function.h:54-96,147-185).
Bucket maps and per-match depths / observations are f32 pipelines with a fixed evaluation order and
must match bit for bit; the 100-iteration Levenberg–Marquardt solve must agree to 1e-8.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vo_parity(vl, orc, synth):
    seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=4)
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    h = vl.Handle(0, with_mapping=0, debug=1)
    h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
    o = orc.VOOracle(cam_T_velo, rect0_T_cam, P, remove_outlier=100)
    for k in range(3):
        cloud = seq.sweep(k)
        h.vo_process_point_cloud(cloud)
        o.reset()
        o.process_point_cloud(cloud)
        if k == 0:
            continue
        prev_uv, curr_uv = synth.synth_matches(seq, k)
        # the LO prior in camera coordinates would be the init; start from zero motion like reset_VO_to_identity = false with an identity prior
        aa, t, c32, c22 = h.vo_solve(prev_uv, curr_uv, np.zeros(3), np.zeros(3))
        r = o.solve(prev_uv, curr_uv, np.zeros(3), np.zeros(3))
        d = h.vo_debug(prev_uv.shape[0])
        for which, (dm, om) in enumerate([(d["cur"], o.buckets(0)), (d["prev"], o.buckets(1))]):
            assert np.array_equal(dm[3], om[3]), "bucket_count map %d" % which
            for a in range(3):
                assert np.array_equal(dm[a].view(np.uint32), om[a].view(np.uint32)), "bucket array %d map %d" % (a, which)
        assert (c32, c22) == (r["counter32"], r["counter22"])
        assert c32 > 200 and c22 > 0
        md = r["match_debug"]
        assert np.array_equal(d["match_rows"][:, 0], md[:, 0])
        assert np.array_equal(d["match_rows"][:, 1].astype(np.float32).view(np.uint32), md[:, 1].astype(np.float32).view(np.uint32))
        assert np.array_equal(d["match_rows"][:, 2:], md[:, 2:]), "K^-1 observations must be bit-identical (same f32 QR)"
        rec = d["rec"]
        assert abs(rec["initial_cost"] - r["initial_cost"]) < 1e-9 * (1 + r["initial_cost"])
        scale = np.sqrt(np.outer(np.diag(r["H0"]), np.diag(r["H0"]))) + 1e-30
        assert np.max(np.abs(rec["H0"] - r["H0"]) / scale) < 1e-9
        assert rec["trace"].shape == r["trace"].shape, (rec["trace"][:, 0], r["trace"][:, 0])
        assert np.allclose(rec["trace"][:, 0], r["trace"][:, 0], rtol=1e-7, atol=1e-12)
        assert np.linalg.norm(aa - r["angles"]) < 1e-8 and np.linalg.norm(t - r["t"]) < 1e-8
        # sanity: the estimate is the camera-frame motion of the generator (p_curr = R p_prev + t convention of CostFunctor32)
        q, tg = seq.gt_relative(k)
        Rcv = cam_T_velo[:3, :3].astype(np.float64)
        Rg = Rcv @ synth.quat_to_rot(q).T @ Rcv.T
        ang = np.linalg.norm(aa)
        assert abs(ang - np.arccos(np.clip((np.trace(Rg) - 1) / 2, -1, 1))) < 5e-3
