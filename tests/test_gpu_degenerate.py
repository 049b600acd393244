"""-m gpu: the reference's degenerate-frame branches through the C ABI vs the CPU oracle, on a handle of their own AND as one session of
a batch of four whose other sessions are ordinary drives.

    (a) laser_odometry.cpp:272,359      every correspondence rejected (d^2 >= 25): a solve without residual blocks — pose = warm start
    (b) laser_odometry.cpp:452-455      1..9 correspondences: "less correspondence" is only a message, the solve proceeds
    (c) laser_mapping.cpp:448,631-635   a NON-empty map with <= 50 surf points: no optimisation, transformUpdate + map insert still run
    (d) mapping_skip_frame = 5          laser_odometry.cpp:618, laser_mapping.cpp:198-204
    (e) visual_odometry.cpp:309-314,345,393,419-421   VO with only CostFunctor22 rows / without any usable match
That the inputs really take those branches is asserted on the oracle in tests/test_oracle_degenerate.py (CPU).  Bars as everywhere:
integer / index work bit for bit, poses 1e-8 (north_star: 1e-4)."""
import numpy as np
import pytest

import degenerate_cases as dc
from test_gpu_laser_odometry import compare_outer, qdist
from test_gpu_laser_mapping import lexsort_rows, oracle_map_points

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-8


def check_row(row, o, with_mapping, tol, what):
    qw, tw, _, _ = o.lo_pose()
    assert qdist(row[0:4], qw) < tol and np.linalg.norm(row[4:7] - tw) < tol, what + ": odometry pose"
    if with_mapping:
        qm, tm = o.map_published_pose()
        assert qdist(row[7:11], qm) < tol and np.linalg.norm(row[11:14] - tm) < tol, what + ": map pose"


def normal_drives(synth, count, n, shape=(64, 512)):
    out = []
    for b in range(count):
        seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=n + 1, seed_scene=2000 + 31 * b, seed_traj=77 + b, seed_noise=900 + 13 * b)
        out.append([np.ascontiguousarray(seq.sweep(k), dtype=np.float32) for k in range(n)])
    return out


@pytest.mark.parametrize("case", ["zero", "few"])
def test_lo_with_no_or_few_correspondences(vl, orc, synth, case):
    """(a), (b) stage by stage on a single handle: correspondences, residuals, normal equations, trust-region trace and termination of
    BOTH outer rounds equal the oracle's — an empty factor table included — and no solver error is raised."""
    clouds = dc.lo_sequence(synth, n=7, far_at=(3,) if case == "zero" else (), wedge_at=() if case == "zero" else (3,))
    h = vl.Handle(0, debug=1, with_mapping=0)
    o = orc.Oracle(with_mapping=False)
    for k, c in enumerate(clouds):
        h.reset_frame()
        h.scan_registration(c)
        qw, tw, ql, tl = h.laser_odometry()
        assert o.process(c) == 0
        oqw, otw, oql, otl = o.lo_pose()
        if k > 0:
            for outer in range(2):
                d = h.lo_debug(outer)
                compare_outer(d, o, outer)
                nf = d["rec"]["n_factors"]
                if case == "zero" and k in (3, 4):
                    assert nf == 0 and d["rec"]["termination"] == 1 and d["rec"]["trace"].shape[0] == 1
                    assert np.array_equal(d["rec"]["x_out"], d["rec"]["x_in"]), "no residual block: the warm start comes back bit for bit"
                elif case == "few" and k == 3:
                    assert 1 <= nf <= 9 and d["rec"]["trace"].shape[0] > 1
                elif case == "few" and k == 4:
                    assert 10 <= nf < 200   # an ordinary sweep against the wedge sweep: only the wedge is there to be matched
                else:
                    assert nf > 200
        assert qdist(ql, oql) < POSE_TOL and np.linalg.norm(tl - otl) < POSE_TOL, "frame %d frame-to-frame pose" % k
        assert qdist(qw, oqw) < POSE_TOL * (k + 1) and np.linalg.norm(tw - otw) < POSE_TOL * (k + 1), "frame %d world pose" % k
    h.sync()   # no sticky error (kErrSolverSync in particular)
    h.close()


@pytest.mark.parametrize("case,with_mapping", [("zero", 0), ("zero", 1), ("few", 1)])
def test_degenerate_session_inside_a_batch(vl, orc, synth, case, with_mapping):
    """(a), (b) as session 2 of a batch of four: the cooperative solves of all sessions share their launches, one of them with an empty
    (or nearly empty) factor table.  Every session — the degenerate one and its three neighbours — against its own oracle run."""
    n = 7
    # (with mapping: 256 columns — the far sweep puts every point into a voxel of its own, and 64 x 512 would exceed the surf voxels
    # a sweep may bring, a stated capacity of the handle; the far sweep's map solve then runs without a single factor as well)
    deg = dc.lo_sequence(synth, n=n, shape=(64, 256) if with_mapping else (64, 512), far_at=(3,) if case == "zero" else (),
                         wedge_at=() if case == "zero" else (3,))
    others = normal_drives(synth, 3, n)
    seqs = [others[0], others[1], deg, others[2]]
    hb = vl.Handle(0, n_sessions=4, with_mapping=with_mapping)
    oracles = [orc.Oracle(with_mapping=bool(with_mapping)) for _ in range(4)]
    refs = [[] for _ in range(4)]
    for k in range(n):
        hb.batch_process_scan([seqs[b][k] for b in range(4)])
        for b in range(4):
            assert oracles[b].process(seqs[b][k]) == 0
            qw, tw, _, _ = oracles[b].lo_pose()
            qm, tm = oracles[b].map_published_pose() if with_mapping else (qw, tw)
            refs[b].append(np.concatenate([qw, tw, qm, tm]))
    hb.sync()   # raises on any sticky error
    for b in range(4):
        tj = hb.select(b).trajectory()
        assert tj.shape == (n, 14)
        for k in range(n):
            r = refs[b][k]
            assert qdist(tj[k, 0:4], r[0:4]) < POSE_TOL * (k + 1) and np.linalg.norm(tj[k, 4:7] - r[4:7]) < POSE_TOL * (k + 1), (b, k)
            if with_mapping:
                assert qdist(tj[k, 7:11], r[7:11]) < POSE_TOL * (k + 1) and np.linalg.norm(tj[k, 11:14] - r[11:14]) < POSE_TOL * (k + 1), (b, k)
    if with_mapping:   # the degenerate session's map: same voxel centroids as the oracle's (far returns included: raw points beyond the valid block)
        hb.select(2)
        got, want = hb.get_map(), np.concatenate([p for c in range(21 * 21 * 11) for kind in (0, 1) for p in [oracles[2].map_cube(kind, c)] if p.shape[0]])
        assert got.shape == want.shape
        ulp = np.abs(got[:, :3].view(np.int32).astype(np.int64) - want[:, :3].view(np.int32).astype(np.int64))
        assert int(ulp.max(initial=0)) <= 1 and float(np.mean(ulp == 0)) > 0.999
    hb.close()


def test_small_map_gate_after_frame_zero(vl, orc, synth):
    """(c) 16 lines x 128 columns: frames 1 and 2 find a map with 38 / 50 surf points — laser_mapping.cpp:448 is false on a non-empty map:
    do_optimize == 0, the pose is the odometry's, transformUpdate and the insert run, the map equals the oracle's bit for bit after every
    frame; from frame 3 on both outer rounds run."""
    clouds = dc.sparse_map_sequence(synth, n=7)
    h = vl.Handle(0, scan_line=16, debug=1, with_mapping=1)
    o = orc.Oracle(scan_line=16, with_mapping=True)
    gate = []
    for k, c in enumerate(clouds):
        h.reset_frame()
        h.scan_registration(c)
        h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(c) == 0
        st = h.map_state()
        gate.append((st["do_optimize"], st["n_map_corner"], st["n_map_surf"]))
        assert st["n_map_corner"] == o.cloud(9).shape[0] and st["n_map_surf"] == o.cloud(10).shape[0]
        assert st["do_optimize"] == (1 if o.map_num_outer() == 2 else 0)
        oq, ot, oqm, otm = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "frame %d map pose" % k
        assert qdist(st["q_wmap_wodom"], oqm) < POSE_TOL and np.linalg.norm(st["t_wmap_wodom"] - otm) < POSE_TOL
        for which in (7, 8):
            dv, rf = h.features(which), o.cloud(which)
            assert dv.shape == rf.shape and np.array_equal(dv[:, :4].view(np.uint32), rf[:, :4].view(np.uint32)), "stack %d" % which
        for kind in (0, 1):
            cnt, pts = h.map_dump(kind)
            ref = oracle_map_points(o, kind)
            assert pts.shape == ref.shape, "frame %d map kind %d: %s vs %s" % (k, kind, pts.shape, ref.shape)
            assert np.array_equal(lexsort_rows(pts)[:, :4].view(np.uint32), lexsort_rows(ref)[:, :4].view(np.uint32)), "frame %d map kind %d" % (k, kind)
    assert [g[0] for g in gate[:4]] == [0, 0, 0, 1] and 0 < gate[1][2] <= 50 and 0 < gate[2][2] <= 50, gate
    h.sync()
    h.close()


def test_small_map_session_inside_a_batch(vl, orc, synth):
    """(c) in a batch: the 16-line session next to three 16-line sessions of other, denser scenes (one handle = one scan_line)."""
    n = 7
    sparse = dc.sparse_map_sequence(synth, n=n)
    others = []
    for b in range(3):
        seq = synth.SynthSequence(n_rings=16, n_azimuth=512, n_sweeps=n + 1, seed_scene=300 + b, seed_traj=5 + b)
        others.append([np.ascontiguousarray(seq.sweep(k), dtype=np.float32) for k in range(n)])
    seqs = [others[0], sparse, others[1], others[2]]
    hb = vl.Handle(0, n_sessions=4, scan_line=16, with_mapping=1)
    oracles = [orc.Oracle(scan_line=16, with_mapping=True) for _ in range(4)]
    for k in range(n):
        hb.batch_process_scan([seqs[b][k] for b in range(4)])
        for b in range(4):
            assert oracles[b].process(seqs[b][k]) == 0
    hb.sync()
    for b in range(4):
        check_row(hb.select(b).trajectory()[-1], oracles[b], True, POSE_TOL * n, "session %d" % b)
    hb.select(1)
    for kind in (0, 1):
        _, pts = hb.map_dump(kind)
        ref = oracle_map_points(oracles[1], kind)
        assert pts.shape == ref.shape and np.array_equal(lexsort_rows(pts)[:, :4].view(np.uint32), lexsort_rows(ref)[:, :4].view(np.uint32))
    hb.close()


def test_mapping_skip_frame_five(vl, orc, synth):
    """(d) mapping_skip_frame = 5 on a single handle and in a batch of two: frames 4 and 9 are mapped, the others report the high-frequency
    pose q_wmap_wodom * q_wodom_curr (laser_mapping.cpp:718-757)."""
    n = 12
    drives = normal_drives(synth, 2, n)
    hs = vl.Handle(0, with_mapping=1, mapping_skip_frame=5)
    hb = vl.Handle(0, n_sessions=2, with_mapping=1, mapping_skip_frame=5)
    oracles = [orc.Oracle(with_mapping=True, mapping_skip_frame=5) for _ in range(2)]
    refs = [[], []]
    for k in range(n):
        hs.process_scan(drives[0][k])
        hb.batch_process_scan([drives[0][k], drives[1][k]])
        for b in range(2):
            assert oracles[b].process(drives[b][k]) == 0
            qw, tw, _, _ = oracles[b].lo_pose()
            qm, tm = oracles[b].map_published_pose()
            refs[b].append(np.concatenate([qw, tw, qm, tm]))
    hs.sync(); hb.sync()
    for name, tj, b in (("single", hs.trajectory(), 0), ("batch 0", hb.select(0).trajectory(), 0), ("batch 1", hb.select(1).trajectory(), 1)):
        for k in range(n):
            r = refs[b][k]
            tol = POSE_TOL * (k + 1)
            assert qdist(tj[k, 0:4], r[0:4]) < tol and np.linalg.norm(tj[k, 4:7] - r[4:7]) < tol, (name, k)
            assert qdist(tj[k, 7:11], r[7:11]) < tol and np.linalg.norm(tj[k, 11:14] - r[11:14]) < tol, (name, k)
    got = hs.get_map()
    want = np.concatenate([p for c in range(21 * 21 * 11) for kind in (0, 1) for p in [oracles[0].map_cube(kind, c)] if p.shape[0]])
    assert got.shape == want.shape and got.shape[0] > 1000
    ulp = np.abs(got[:, :3].view(np.int32).astype(np.int64) - want[:, :3].view(np.int32).astype(np.int64))
    assert int(ulp.max(initial=0)) <= 1 and float(np.mean(ulp == 0)) > 0.999
    hs.close(); hb.close()


def test_vo_without_usable_matches(vl, orc, synth):
    """(e) stand-alone VO solve: every match without LiDAR depth (1 400 CostFunctor22 rows, no CostFunctor32), no match at all, every match
    beyond remove_VO_outlier — counters, trust-region trace and estimate equal the oracle's; an empty problem hands the initial guess back."""
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=4)
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    h = vl.Handle(0, with_mapping=0, debug=1)
    h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
    o = orc.VOOracle(cam_T_velo, rect0_T_cam, P, remove_outlier=100)
    for k in range(2):
        c = seq.sweep(k)
        h.vo_process_point_cloud(c)
        o.reset()
        o.process_point_cloud(c)
    a0, t0 = np.array([0.001, -0.002, 0.0005]), np.array([0.01, 0.02, -0.5])
    for name, pu, cu in dc.vo_cases(synth, seq, 1):
        aa, t, c32, c22 = h.vo_solve(pu, cu, a0, t0)
        r = o.solve(pu, cu, a0, t0)
        assert (c32, c22) == (r["counter32"], r["counter22"]), name
        rec = h.vo_debug(pu.shape[0])["rec"]
        assert rec["trace"].shape == r["trace"].shape and rec["termination"] == r["termination"], name
        assert np.array_equal(rec["trace"][:, 6:8], r["trace"][:, 6:8]), name
        if name == "no_depth":
            assert c32 == 0 and c22 > 1000
            assert np.allclose(rec["trace"][:, 0], r["trace"][:, 0], rtol=1e-7, atol=1e-12)
            assert np.linalg.norm(aa - r["angles"]) < 1e-8 and np.linalg.norm(t - r["t"]) < 1e-8
        else:
            assert c32 == 0 and c22 == 0 and rec["n_factors"] == 0
            assert np.array_equal(aa, a0) and np.array_equal(t, t0), name
    h.sync()
    h.close()


def test_coupled_frames_with_a_matchless_frame(vl, synth):
    """(e) inside the coupled loop: frame 3 arrives without a single pixel match (and frame 5 with outliers only) — the VO solve returns its
    initial guess (cam0_curr_LOT_cam0_prev), VO2VeloAndBase, the combined-mode odometry and the mapping go on; single handle and batch."""
    import orc_vloam
    n = 8
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1)
    clouds = [np.ascontiguousarray(seq.sweep(k), dtype=np.float32) for k in range(n)]
    empty = np.zeros((0, 2), dtype=np.int32)
    matches = []
    for k in range(n):
        m = synth.synth_matches(seq, k) if k > 0 else (None, None)
        if k == 3:
            m = (empty, empty)
        if k == 5:
            m = (m[0], m[1] + 300)
        matches.append(m)

    def setup(h):
        h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
        h.set_extrinsics(base_T_cam0, velo_T_cam0)
        return h

    o = orc_vloam.VloamOracle(cam_T_velo, rect0_T_cam, P, base_T_cam0, velo_T_cam0, detach_VO_LO=False, with_mapping=True)
    hs = setup(vl.Handle(0, with_mapping=1, detach_VO_LO=0))
    hb = setup(vl.Handle(0, n_sessions=2, with_mapping=1, detach_VO_LO=0))
    normal = [synth.synth_matches(seq, k) if k > 0 else (None, None) for k in range(n)]
    for k in range(n):
        hs.process_frame(clouds[k], matches[k][0], matches[k][1])
        hb.batch_process_frame([clouds[k], clouds[k]], [normal[k], matches[k]])
        assert o.process(clouds[k], matches[k][0], matches[k][1]) == 0
        if k in (3, 5):
            assert o.vo_result["counter32"] == 0 and o.vo_result["counter22"] == 0
            assert np.array_equal(o.vo_result["angles"], o.vo_result["init_angles"])
            r = hs.vo_result()
            assert (r["counter32"], r["counter22"]) == (0, 0)
            assert np.linalg.norm(r["angles"] - o.vo_result["angles"]) < 1e-7 and np.linalg.norm(r["t"] - o.vo_result["t"]) < 1e-7
    hs.sync(); hb.sync()
    qw, tw, _, _ = o.lidar.lo_pose()
    qm, tm = o.lidar.map_published_pose()
    vq, vt = o.vo_world_pose()
    for name, h in (("single", hs), ("batch", hb.select(1))):
        tj, vj = h.trajectory()[-1], h.vo_trajectory()[-1]
        tol = 1e-6   # frame 1's 2 acos(w) start (tests/test_gpu_vloam.py) rides along
        assert qdist(tj[0:4], qw) < tol and np.linalg.norm(tj[4:7] - tw) < tol, name
        assert qdist(tj[7:11], qm) < tol and np.linalg.norm(tj[11:14] - tm) < tol, name
        assert qdist(vj[0:4], vq) < tol and np.linalg.norm(vj[4:7] - vt) < tol, name
    hs.close(); hb.close()
