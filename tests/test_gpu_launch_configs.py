"""-m gpu: the LiDAR launch configurations the reference SHIPS, with the parameters of their own launch files.

    lidar_odometry_mapping/launch/loam_velodyne_VLP_16.launch:3-13   scan_line 16, minimum_range 0.3, mapping_line_resolution 0.2,
                                                                     mapping_plane_resolution 0.4, mapping_skip_frame 1
    lidar_odometry_mapping/launch/loam_velodyne_HDL_32.launch:3-13   scan_line 32, otherwise the same
    lidar_odometry_mapping/launch/loam_velodyne_HDL_64.launch:3-13   scan_line 64, minimum_range 5, 0.4 / 0.8 (the KITTI one; every other test)

LaserMapping::init takes any resolution (laser_mapping.cpp:95-101).  Until round 6 vloam_create refused leaves below 0.25 m (8 voxel bits per
axis in the map's voxel key); the key now has 9.  Per sweep: the laser-odometry pose and the map pose against the oracle; at the end the
whole /laser_cloud_map bit for bit — alone, and as one session of a batch of three (a batch shares one rig, i.e. one configuration).
A sensor at walking / slow-driving speed: the VLP-16 and HDL-32 are not car-roof sensors in the reference's launch files.
"""
import numpy as np
import pytest

from test_gpu_laser_mapping import lexsort_rows, oracle_map_points, qdist

pytestmark = pytest.mark.gpu

LAUNCH = {
    # name: (scan_line, columns per revolution at 10 Hz, launch-file parameters)
    "VLP_16": (16, 1800, dict(minimum_range=0.3, mapping_line_resolution=0.2, mapping_plane_resolution=0.4, mapping_skip_frame=1)),
    "HDL_32": (32, 2170, dict(minimum_range=0.3, mapping_line_resolution=0.2, mapping_plane_resolution=0.4, mapping_skip_frame=1)),
}
N_SWEEPS = 32
POSE_TOL = 1e-8


def oracle_for(orc, name):
    rings, _, p = LAUNCH[name]
    return orc.Oracle(scan_line=rings, minimum_range=p["minimum_range"], line_res=p["mapping_line_resolution"], plane_res=p["mapping_plane_resolution"],
                      mapping_skip_frame=p["mapping_skip_frame"], with_mapping=True)


def sequence(synth, name, n, speed, **seeds):
    """Sweeps of the sensor model + a patch of SELF-HITS (the carrier's body: two scan lines x 80 columns at 0.15 - 0.9 m along their beams,
    fixed in the sensor frame): returns on both sides of the 0.3 m threshold of removeClosedPointCloud (scan_registration.cpp:100-129)."""
    rings, az, _ = LAUNCH[name]
    seq = synth.SynthSequence(n_rings=rings, n_azimuth=az, n_sweeps=n + 1, speed=speed, **seeds)
    rng = np.random.default_rng(seeds.get("seed_noise", 0) + 99)
    out = []
    for k in range(n):
        c = seq.sweep(k)
        for ring in (1, 2):
            cols = ring * az + az // 3 + np.arange(80)
            r = rng.uniform(0.15, 0.9, 80)
            c[cols, :3] = (seq.dirs[cols] * r[:, None]).astype(np.float32)
        out.append(c)
    return out


def oracle_run(orc, name, clouds):
    o = oracle_for(orc, name)
    rows = []
    for c in clouds:
        assert o.process(c) == 0
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        rows.append(np.concatenate([qw, tw, qm, tm]))
    return o, np.array(rows)


def assert_poses(tj, ref, what):
    assert tj.shape == ref.shape, (tj.shape, ref.shape)
    for k in range(ref.shape[0]):
        assert qdist(tj[k, 0:4], ref[k, 0:4]) < POSE_TOL and np.linalg.norm(tj[k, 4:7] - ref[k, 4:7]) < POSE_TOL, "%s: LO pose, sweep %d" % (what, k)
        assert qdist(tj[k, 7:11], ref[k, 7:11]) < POSE_TOL and np.linalg.norm(tj[k, 11:14] - ref[k, 11:14]) < POSE_TOL, "%s: map pose, sweep %d" % (what, k)


def assert_map(h, o, what):
    for kind in (0, 1):
        _, pts = h.map_dump(kind)
        ref = oracle_map_points(o, kind)
        assert pts.shape == ref.shape, (what, kind, pts.shape, ref.shape)
        assert ref.shape[0] > 500, "the %s map should not be trivial" % what
        assert np.array_equal(lexsort_rows(pts)[:, :4].view(np.uint32), lexsort_rows(ref)[:, :4].view(np.uint32)), "%s: map kind %d" % (what, kind)
    # /laser_cloud_map in the reference's publishing order (laser_mapping.cpp:778-793)
    pub = h.get_map()
    ref = np.concatenate([np.concatenate([o.map_cube(0, c), o.map_cube(1, c)]) for c in range(21 * 21 * 11)])
    assert pub.shape == ref.shape and np.array_equal(pub[:, :4].view(np.uint32), ref[:, :4].view(np.uint32)), "%s: /laser_cloud_map order" % what


@pytest.mark.parametrize("name,speed", [("VLP_16", 1.5), ("HDL_32", 4.0)])
def test_launch_file_configuration_alone(vl, orc, synth, name, speed):
    rings, _, params = LAUNCH[name]
    clouds = sequence(synth, name, N_SWEEPS, speed)
    o, ref = oracle_run(orc, name, clouds)
    # the configuration must do real work: both solves run, with the near returns the 0.3 m range keeps
    assert o.map_num_outer() == 2
    near = np.linalg.norm(np.nan_to_num(clouds[5][:, :3], nan=100.0), axis=1)
    assert np.count_nonzero(near < 0.3) > 10 and np.count_nonzero((near > 0.3) & (near < 5.0)) > 100, "returns on both sides of minimum_range 0.3, and ones the KITTI setting (5 m) drops"
    h = vl.Handle(0, scan_line=rings, with_mapping=1, **params)
    for c in clouds:
        h.process_scan(c)
    h.sync()
    assert_poses(h.trajectory(), ref, name)
    assert_map(h, o, name)
    assert h.health()["fallback_solves"] == 0
    h.close()


@pytest.mark.parametrize("name,speed", [("VLP_16", 1.5), ("HDL_32", 4.0)])
def test_launch_file_configuration_stagewise_with_factor_sets(vl, orc, synth, name, speed):
    """The stage-wise calls (ScanRegistration / LaserOdometry / LaserMapping of the reference's façade) with the debug hooks: the stacks
    (VoxelGrid 0.2 / 0.4 of the scan features) bit for bit, factor sets, fitted lines / planes and both trust-region traces per sweep."""
    from test_gpu_laser_mapping import compare_map_round
    rings, _, params = LAUNCH[name]
    clouds = sequence(synth, name, 8, speed)
    o = oracle_for(orc, name)
    h = vl.Handle(0, scan_line=rings, with_mapping=1, debug=1, **params)
    for k, c in enumerate(clouds):
        h.reset_frame()
        h.scan_registration(c)
        h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(c) == 0
        for which in (7, 8):
            dv, rf = h.features(which), o.cloud(which)
            assert dv.shape == rf.shape and np.array_equal(dv[:, :4].view(np.uint32), rf[:, :4].view(np.uint32)), "stack %d, sweep %d" % (which, k)
        if k > 0:
            assert o.map_num_outer() == 2
            for outer in range(2):
                compare_map_round(h, o, outer)
        oq, ot, _, _ = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "map pose, sweep %d" % k
    h.close()


@pytest.mark.parametrize("name,speed", [("VLP_16", 1.5), ("HDL_32", 4.0)])
def test_launch_file_configuration_as_a_session_of_a_batch(vl, orc, synth, name, speed):
    rings, _, params = LAUNCH[name]
    B = 3
    seqs = [sequence(synth, name, N_SWEEPS, speed * (1.0 + 0.2 * b), seed_scene=1234 + 31 * b, seed_traj=42 + b, seed_noise=5678 + 1000 * b) for b in range(B)]
    hb = vl.Handle(0, n_sessions=B, scan_line=rings, with_mapping=1, **params)
    for k in range(N_SWEEPS):
        hb.batch_process_scan([seqs[b][k] for b in range(B)])
    hb.sync()
    for b in range(B):
        o, ref = oracle_run(orc, name, seqs[b])
        hb.select(b)
        assert_poses(hb.trajectory(), ref, "%s session %d" % (name, b))
        assert_map(hb, o, "%s session %d" % (name, b))
    hb.close()


def test_leaf_bound(vl):
    """What vloam_create still refuses: a leaf whose 75 x radix^3 voxel positions do not fit the 32-bit tie rank (< 0.132 m)."""
    for leaf in (0.2, 0.15, 0.135):
        vl.Handle(0, with_mapping=1, mapping_line_resolution=leaf, mapping_plane_resolution=leaf, map_capacity_log2=12).close()
    for leaf in (0.13, 0.1, 0.0, -1.0):
        with pytest.raises(vl.VloamError) as e:
            vl.Handle(0, mapping_line_resolution=leaf)
        assert e.value.status == vl.ERR_INVALID
