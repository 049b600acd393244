"""CPU: the oracle's side of the degenerate-frame cases the -m gpu file (tests/test_gpu_degenerate.py) holds the device to.

Each case must really take the reference's branch it is named after — otherwise the GPU comparison would pass on ordinary frames:
    laser_odometry.cpp:272,359 (every correspondence rejected), :452-455 (< 10 correspondences, the solve proceeds),
    laser_mapping.cpp:448,631-635 (no optimisation on a small but non-empty map), mapping_skip_frame = 5 (laser_odometry.cpp:618,
    laser_mapping.cpp:198-204), visual_odometry.cpp:309-314,345,393,419-421 (no usable match / only CostFunctor22 rows).
A Ceres problem without residual blocks: Program::RemoveFixedBlocks drops the two parameter blocks (no residual block uses them), the
trust-region preprocessor sees an empty reduced program and Solve returns CONVERGENCE with the parameters untouched — the restated
loop reaches the same state through its gradient test (|g| = 0 <= 1e-10) after iteration 0."""
import numpy as np

import degenerate_cases as dc


def qdist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def test_zero_correspondences_keep_the_warm_start(orc, synth):
    clouds = dc.lo_sequence(synth, n=7, far_at=(3,))
    o = orc.Oracle(with_mapping=False)
    last = None
    for k, c in enumerate(clouds):
        assert o.process(c) == 0
        _, _, ql, tl = o.lo_pose()
        if k in (3, 4):   # the far sweep against a normal one, then a normal sweep against the far one
            for outer in range(2):
                cc, pp = o.lo_corr(outer)
                s = o.lo_solve(outer)
                assert cc.shape[0] == 0 and pp.shape[0] == 0
                assert s["termination"] == 1 and s["trace"].shape[0] == 1 and s["initial_cost"] == 0.0
                assert np.array_equal(s["q_out"], s["q_in"]) and np.array_equal(s["t_out"], s["t_in"])
            assert np.array_equal(ql, last[0]) and np.array_equal(tl, last[1]), "q_last_curr / t_last_curr are the warm start, untouched"
        elif k > 0:
            assert o.lo_corr(1)[0].shape[0] > 100 and o.lo_corr(1)[1].shape[0] > 100
        last = (ql.copy(), tl.copy())
    assert np.isfinite(o.lo_pose()[1]).all()


def test_fewer_than_ten_correspondences_still_solve(orc, synth):
    clouds = dc.lo_sequence(synth, n=6, far_at=(), wedge_at=(3,))
    o = orc.Oracle(with_mapping=False)
    for k, c in enumerate(clouds):
        assert o.process(c) == 0
        if k == 3:
            for outer in range(2):
                cc, pp = o.lo_corr(outer)
                s = o.lo_solve(outer)
                assert 1 <= cc.shape[0] + pp.shape[0] <= 9, (cc.shape, pp.shape)   # laser_odometry.cpp:452: "less correspondence!"
                assert s["trace"].shape[0] > 1, "the solve runs all the same (:457-463)"
                assert not np.array_equal(s["t_out"], s["t_in"])


def test_small_map_skips_the_optimisation_but_not_the_update(orc, synth):
    clouds = dc.sparse_map_sequence(synth, n=7)
    o = orc.Oracle(scan_line=16, with_mapping=True)
    surf_at_gather, totals = [], []
    for k, c in enumerate(clouds):
        assert o.process(c) == 0
        surf_at_gather.append(o.cloud(10).shape[0])
        info = o.map_info()
        totals.append((info["total_corner"], info["total_surf"]))
        if k < 3:
            assert o.map_num_outer() == 0                      # laser_mapping.cpp:448 false: "Map corner and surf num are not enough"
            qo, to, _, _ = o.lo_pose()
            qm, tm, qwm, twm = o.map_pose()
            assert qdist(qm, qo) < 1e-12 and np.linalg.norm(tm - to) < 1e-12   # initial guess = odometry pose (q_wmap_wodom = identity) ...
            assert qdist(qwm, [0, 0, 0, 1]) < 1e-12 and np.linalg.norm(twm) < 1e-12   # ... and transformUpdate (:636) keeps it so
        else:
            assert o.map_num_outer() == 2
    assert surf_at_gather[:4] == [0, 38, 50, 59], surf_at_gather   # 1 .. 50 points AFTER frame 0: the gate with a non-empty map
    assert all(b[0] > a[0] for a, b in zip(totals[:3], totals[1:4])), "the map insert (:639-683) runs on un-optimised frames"


def test_mapping_skip_frame_five(orc, synth):
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=13)
    o = orc.Oracle(with_mapping=True, mapping_skip_frame=5)
    mapped = []
    for k in range(12):
        assert o.process(seq.sweep(k)) == 0
        info = o.map_info()
        mapped.append(info["total_surf"])
    # frameCount % 5 == 0 after the increment (laser_odometry.cpp:535,618): frames 4 and 9 are mapped, every other frame is skipped
    grew = [k for k in range(12) if mapped[k] > (mapped[k - 1] if k else 0)]
    assert grew == [4, 9], grew


def test_vo_without_usable_matches(orc, synth):
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=4)
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    o = orc.VOOracle(cam_T_velo, rect0_T_cam, P, remove_outlier=100)
    for k in range(2):
        o.reset()
        o.process_point_cloud(seq.sweep(k))
    a0, t0 = np.array([0.001, -0.002, 0.0005]), np.array([0.01, 0.02, -0.5])
    for name, pu, cu in dc.vo_cases(synth, seq, 1):
        r = o.solve(pu, cu, a0, t0)
        if name == "no_depth":
            assert r["counter32"] == 0 and r["counter22"] == pu.shape[0] > 1000   # only epipolar rows: the translation scale is unobservable
            assert r["trace"].shape[0] > 1
        else:
            assert r["counter32"] == 0 and r["counter22"] == 0 and r["termination"] == 1 and r["trace"].shape[0] == 1
            assert np.array_equal(r["angles"], a0) and np.array_equal(r["t"], t0), "no residual block: the initial guess comes back"
