"""-m gpu: the coupled per-frame VLOAM loop (configs[3], synthetic analogue) through the C ABI vs the CPU oracle.

Reference: MAIN/src/vloam_main_node.cpp:125-180 (callback), TF/src/vloam_tf.cpp:55-75 (VO2VeloAndBase),
LOM/src/laser_odometry.cpp:223-236 (combined mode: VO prior overwrites the warm start in BOTH outer rounds, quirk A.8-4) and
:563-567 (LO -> VO prior), VO/src/visual_odometry.cpp:258-281,425-430.  One vloam_process_frame per frame: VO solve ->
VO2VeloAndBase -> scan registration -> laser odometry (detach_VO_LO = 0) -> LO -> VO prior -> mapping, all on the device.
"""
import numpy as np
import pytest

from test_gpu_laser_odometry import compare_outer, qdist

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-8


def make(vl, synth, detach, **kw):
    import orc_vloam
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
    h = vl.Handle(0, detach_VO_LO=int(detach), **kw)
    h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
    h.set_extrinsics(base_T_cam0, velo_T_cam0)
    o = orc_vloam.VloamOracle(cam_T_velo, rect0_T_cam, P, base_T_cam0, velo_T_cam0, detach_VO_LO=detach,
                              with_mapping=bool(kw.get("with_mapping", 1)))
    return h, o


def test_process_frame_needs_calibration_and_extrinsics(vl, synth, sweeps):
    h = vl.Handle(0, with_mapping=0)
    with pytest.raises(vl.VloamError) as e:
        h.process_frame(sweeps(64, 512, 0))
    assert e.value.status == vl.ERR_ORDER


@pytest.mark.parametrize("shape,nframes", [((64, 2048), 22), ((64, 512), 12)])
def test_coupled_frame_loop_parity(vl, synth, shape, nframes):
    seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=nframes + 1)
    h, o = make(vl, synth, detach=False, debug=1, with_mapping=1)
    for k in range(nframes):
        cloud = seq.sweep(k)
        m = synth.synth_matches(seq, k) if k > 0 else (None, None)
        h.process_frame(cloud, m[0], m[1])
        assert o.process(cloud, m[0], m[1]) == 0
        r = h.vo_result()
        oq, ot = o.lo_prior()
        if k == 0:
            assert o.vo_result is None and np.allclose(r["prior_q"], [0, 0, 0, 1], atol=1e-15) and np.allclose(r["prior_t"], 0, atol=1e-15)
        else:
            v = o.vo_result
            assert (r["counter32"], r["counter22"]) == (v["counter32"], v["counter22"]) and r["counter32"] > 200
            # frame 1's initial guess goes through 2 acos(w) with w one ulp from 1 (see tests/test_oracle_vloam.py): a ~3e-8 rad
            # libm-dependent start; everything after that is well conditioned
            tol = 2e-7 if k == 1 else POSE_TOL
            assert np.linalg.norm(r["angles"] - v["angles"]) < tol and np.linalg.norm(r["t"] - v["t"]) < tol, "VO estimate, frame %d" % k
            assert qdist(r["prior_q"], oq) < tol and np.linalg.norm(r["prior_t"] - ot) < tol, "velo_last_VOT_velo_curr, frame %d" % k
            # laser odometry: identical correspondences / residuals / trust-region trace in BOTH outer rounds, and both rounds start
            # from the VO prior (the result of round 0 is discarded, quirk A.8-4)
            assert o.lidar.lo_num_outer() == 2
            if k > 1:
                for outer in range(2):
                    d = h.lo_debug(outer)
                    compare_outer(d, o.lidar, outer)
                    assert np.array_equal(d["rec"]["x_in"][:4], r["prior_q"]) and np.array_equal(d["rec"]["x_in"][4:], r["prior_t"])
                assert np.linalg.norm(h.lo_debug(0)["rec"]["x_out"] - h.lo_debug(0)["rec"]["x_in"]) > 0
        tol = 1e-6 if k == 1 else POSE_TOL * (k + 1)
        tj = h.trajectory()[k]
        qw, tw, _, _ = o.lidar.lo_pose()
        qm, tm = o.lidar.map_published_pose()
        assert qdist(tj[0:4], qw) < tol and np.linalg.norm(tj[4:7] - tw) < tol, "LO world pose, frame %d" % k
        assert qdist(tj[7:11], qm) < tol and np.linalg.norm(tj[11:14] - tm) < tol, "map pose, frame %d" % k
        vq, vt = o.vo_world_pose()
        vj = h.vo_trajectory()[k]
        assert qdist(vj[0:4], vq) < tol and np.linalg.norm(vj[4:7] - vt) < tol, "world_VOT_base_last, frame %d" % k
    h.sync()


def test_pipelined_frames_match_the_per_frame_run(vl, synth):
    """24 frames streamed with no host synchronisation in between (the odometry of frame k is enqueued while frame k + 1 arrives)
    give bit-identical trajectories to frame-by-frame calls with a sync after each; detached mode (D) ignores the prior in LO."""
    n = 24
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1)
    clouds = [seq.sweep(k) for k in range(n)]
    ms = [(None, None)] + [synth.synth_matches(seq, k) for k in range(1, n)]
    out = {}
    for detach in (False, True):
        h1, _ = make(vl, synth, detach=detach, with_mapping=1)
        for k in range(n):
            h1.process_frame(clouds[k], *ms[k])
        h1.sync()
        h2, o = make(vl, synth, detach=detach, with_mapping=1)
        for k in range(n):
            h2.process_frame(clouds[k], *ms[k])
            h2.sync()
        assert np.array_equal(h1.trajectory(), h2.trajectory()) and np.array_equal(h1.vo_trajectory(), h2.vo_trajectory())
        for k in range(n):
            o.process(clouds[k], *ms[k])
        qw, tw, _, _ = o.lidar.lo_pose()
        tj = h1.trajectory()[n - 1]
        assert qdist(tj[0:4], qw) < 1e-6 and np.linalg.norm(tj[4:7] - tw) < 1e-6
        out[detach] = h1.trajectory()
    assert not np.array_equal(out[False][5:, :7], out[True][5:, :7])   # the prior does change the laser-odometry result
