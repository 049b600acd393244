"""-m gpu: the cube window of the map rolled along every axis in both directions.

The reference keeps a 21 x 21 x 11 array of 50 m cubes around the sensor and shifts it with six literal `while` loops — one per axis and
direction — whenever the centre cube index leaves [3, size - 3) (laser_mapping.cpp:218-402); the slab that wraps round is cleared.  A street
drive only ever runs one or two of the six.  Here LaserMapping::input is handed the odometry pose plus the offsets of
branch_cases.six_way_walk (the reference takes whatever pose its caller hands it, :167-196; on the device: vloam_set_mapping_input): every loop
runs, cubes leave the window at both ends of every axis (the device purges their voxels) and re-enter empty, and the sensor comes back through
places it mapped before.  tests/test_oracle_pipeline.py replays the same walk in literal Python against the oracle."""
import numpy as np
import pytest

import branch_cases
from test_gpu_laser_mapping import oracle_published_map, qdist

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-8


@pytest.mark.parametrize("rings,n_az", [(64, 256), (16, 1024)])
def test_window_rolls_in_all_six_directions(vl, orc, synth, rings, n_az):
    walk = branch_cases.six_way_walk()
    n = len(walk)
    seq = synth.SynthSequence(n_rings=rings, n_azimuth=n_az, n_sweeps=n)
    h = vl.Handle(0, scan_line=rings, with_mapping=1, max_frames=n + 8)
    o = orc.Oracle(scan_line=rings, with_mapping=True)
    cens, solved, worst = [], 0, 0.0
    for k in range(n):
        cloud = seq.sweep(k)
        h.reset_frame()
        h.scan_registration(cloud)
        qw, tw, _, _ = h.laser_odometry()
        assert o.stage_sr(cloud) == 0
        o.stage_lo()
        oq, ot, _, _ = o.lo_pose()
        assert qdist(qw, oq) < POSE_TOL and np.linalg.norm(tw - ot) < POSE_TOL, "odometry pose, sweep %d" % k
        # both sides are handed the SAME pose (the oracle's odometry + the walk): what is compared is the mapping stage
        h.set_mapping_input(q_wodom_curr=oq, t_wodom_curr=ot + walk[k])
        qm, tm = h.laser_mapping()
        assert o.stage_map(q=oq, t=ot + walk[k]) == 0
        mq, mt = o.map_published_pose()
        assert qdist(qm, mq) < POSE_TOL and np.linalg.norm(tm - mt) < POSE_TOL, "map pose, sweep %d" % k
        worst = max(worst, float(np.linalg.norm(tm - mt)))
        info = o.map_info()
        st = h.map_state()
        assert np.array_equal(st["cen"], info["cen"]), "window position, sweep %d" % k
        cens.append(tuple(int(c) for c in info["cen"]))
        solved += int(st["do_optimize"])   # the gate of laser_mapping.cpp:448 (points of the valid 5 x 5 x 3 block)
        if k % 8 == 7 or k == n - 1:   # points per cube (which voxels merged, which cubes were cleared) along the way, both kinds
            cc = h.debug_raw(2, 66, np.int32).reshape(2, 21 * 21 * 11)
            for kind in (0, 1):
                want = np.array([o.map_cube(kind, c).shape[0] for c in range(21 * 21 * 11)], np.int32)
                assert np.array_equal(cc[kind], want), "points per cube, kind %d, sweep %d" % (kind, k)
    c = np.array(cens)
    d = np.diff(c, axis=0)
    for a in range(3):
        assert (d[:, a] > 0).any() and (d[:, a] < 0).any(), "axis %d did not roll both ways: %s" % (a, sorted(set(c[:, a])))
    h.sync()
    got, want = h.get_map(), oracle_published_map(o)
    assert got.shape == want.shape and got.shape[0] > 1000
    # Same points in the same order, intensities bit for bit; coordinates: map points are f32(q p + t) of f64 poses that agree with the oracle's
    # to round-off (bar above: 1e-8; measured: up to 7e-10 m at |t| = 600 m over 59 sweeps of solves built with -ffp-contract=fast on 8 workgroups'
    # partial sums), so a coordinate may sit on the other side of an f32 rounding boundary (1 ulp: 3e-5 m at 440 m) and a coordinate near zero,
    # whose ulp is smaller than that round-off, by the round-off itself.  Everything that is index work — which cube, which voxel, how many
    # points, which order — is exact (checked above per cube and here by position).  Near the origin short runs stay bit for bit
    # (test_gpu_laser_mapping.py).
    g, w = np.ascontiguousarray(got[:, :4]), np.ascontiguousarray(want[:, :4])
    assert np.array_equal(g[:, 3].view(np.uint32), w[:, 3].view(np.uint32)), "intensities"
    ulp = np.abs(g[:, :3].view(np.int32).astype(np.int64) - w[:, :3].view(np.int32).astype(np.int64))
    absd = np.abs(g[:, :3].astype(np.float64) - w[:, :3].astype(np.float64))
    bad = (ulp > 1) & (absd > 1e-8)
    assert not bad.any(), "map coordinates beyond rounding: rows %s\n%s\n%s" % (np.nonzero(bad.any(axis=1))[0][:8], g[bad.any(axis=1)][:6], w[bad.any(axis=1)][:6])
    assert float(np.mean(ulp == 0)) > 0.999, float(np.mean(ulp == 0))
    print("six-way walk, %d x %d: %d sweeps, scan-to-map solved on %d, worst |dt| of the map pose %.2e m, %d of %d map coordinates not bit-equal (max %d ulp)"
          % (rings, n_az, n, solved, worst, int(np.count_nonzero(ulp)), ulp.size, int(ulp.max())))
    assert solved >= 10, "the scan-to-map optimisation ran on %d of %d sweeps only" % (solved, n)
    h.close()
