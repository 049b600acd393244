"""-m gpu: batched execution — one handle, B independent sequences per launch chain (blockIdx.z = session).

The sessions share the launch chain and nothing else.  Everything that is integer / index / f32 point work — the scan-registration
clouds, the picks, the down-sampled scan features, the key points and tracked matches of the image front-end — is bit-identical to the
same sequence run alone on a single-session handle.  The f64 poses agree to round-off: since round 4 a single sequence adds the partial
sums of its Levenberg-Marquardt solves over 8 workgroups and a batch over 4 / 6 (csrc/lm_solve.hip: lm_launch), i.e. in another order
— asserted here at 1e-9 against the single-session run AND at 1e-8 against the CPU oracle (north_star bar: 1e-4); the map, f32
transforms of those poses, is the same cloud up to a unit in the last place of a few coordinates.
Reference for what one session computes: lidar_odometry_mapping.cpp:65-154."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-9


def same_poses(a, b, tol=POSE_TOL):
    """Trajectory rows (q, t [, q_map, t_map]) equal to round-off; quaternions up to sign."""
    if a.shape != b.shape:
        return False
    ok = True
    for lo in range(0, a.shape[1], 7):
        qa, qb = a[:, lo:lo + 4], b[:, lo:lo + 4]
        dq = np.minimum(np.abs(qa - qb).max(axis=1), np.abs(qa + qb).max(axis=1))
        ok = ok and float(dq.max(initial=0)) < tol and float(np.abs(a[:, lo + 4:lo + 7] - b[:, lo + 4:lo + 7]).max(initial=0)) < tol
    return ok


def same_map(a, b):
    """Same points in the same order; coordinates equal or neighbouring floats (f32(q p + t) of poses that agree to 1e-13), almost all equal."""
    if a.shape != b.shape:
        return False
    ulp = np.abs(a[:, :3].view(np.int32).astype(np.int64) - b[:, :3].view(np.int32).astype(np.int64))
    return int(ulp.max(initial=0)) <= 1 and float(np.mean(ulp == 0)) > 0.999


def sequences(synth, B, n, shape=(64, 512)):
    out = []
    for b in range(B):
        seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=n + 1, seed_scene=1234 + 17 * b, seed_traj=42 + b,
                                  seed_noise=5678 + 1000 * b)
        out.append([seq.sweep(k) for k in range(n)])
    return out


@pytest.mark.parametrize("B,skip", [(4, 1), (3, 2)])
def test_batched_sessions_equal_single_session_runs(vl, orc, synth, B, skip):
    n = 14
    seqs = sequences(synth, B, n)
    hb = vl.Handle(0, n_sessions=B, with_mapping=1, mapping_skip_frame=skip)
    for k in range(n):
        hb.batch_process_scan([seqs[b][k] for b in range(B)])
    hb.sync()
    trajs = []
    for b in range(B):
        hs = vl.Handle(0, with_mapping=1, mapping_skip_frame=skip)
        for k in range(n):
            hs.process_scan(seqs[b][k])
        hs.sync()
        hb.select(b)
        tb, ts = hb.trajectory(), hs.trajectory()
        mb, ms = hb.get_map(), hs.get_map()
        assert tb.shape == (n, 14) and mb.shape == ms.shape and mb.shape[0] > 1000
        assert same_poses(tb, ts), "session %d trajectory" % b
        assert same_map(mb, ms), "session %d map" % b
        # ... and against the oracle (north_star bar 1e-4)
        o = orc.Oracle(with_mapping=True, mapping_skip_frame=skip)
        for k in range(n):
            assert o.process(seqs[b][k]) == 0
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        assert same_poses(tb[-1:, :7], np.concatenate([qw, tw])[None, :], 1e-8) and same_poses(tb[-1:, 7:], np.concatenate([qm, tm])[None, :], 1e-8), b
        for which in (0, 2, 4, 7, 8):   # integer / index / f32 work: bit for bit
            fb, fs = hb.features(which), hs.features(which)
            assert fb.shape == fs.shape and np.array_equal(fb.view(np.uint32), fs.view(np.uint32)), "session %d cloud %d" % (b, which)
        assert hb.counts() == hs.counts()
        trajs.append(tb)
    for b in range(1, B):
        assert not np.array_equal(trajs[0], trajs[b]), "the sessions are different sequences"


def test_batched_sessions_with_different_sweep_sizes(vl, synth):
    """Sessions may bring sweeps of different sizes (the launch geometry follows the largest, blocks beyond a session's n idle)."""
    n = 8
    a = sequences(synth, 1, n, (64, 512))[0]
    seq16 = synth.SynthSequence(n_rings=64, n_azimuth=256, n_sweeps=n + 1, seed_scene=99)
    b_ = [seq16.sweep(k) for k in range(n)]
    hb = vl.Handle(0, n_sessions=2, with_mapping=1)
    for k in range(n):
        hb.batch_process_scan([a[k], b_[k]])
    hb.sync()
    for b, clouds in enumerate((a, b_)):
        hs = vl.Handle(0, with_mapping=1)
        for c in clouds:
            hs.process_scan(c)
        hs.sync()
        assert same_poses(hb.select(b).trajectory(), hs.trajectory()), b


def test_single_sequence_entry_points_refuse_a_batched_handle(vl, sweeps):
    hb = vl.Handle(0, n_sessions=2, with_mapping=0)
    with pytest.raises(vl.VloamError) as e:
        hb.process_scan(sweeps(64, 512, 0))
    assert e.value.status == vl.ERR_INVALID
    with pytest.raises(vl.VloamError):
        hb.select(2)
    with pytest.raises(vl.VloamError):
        vl.Handle(0, n_sessions=0)


def test_batched_coupled_frames_equal_single_session_runs(vl, synth):
    """vloam_batch_process_frame[_device]: B coupled VLOAM sessions (depth-enhanced VO + VO2VeloAndBase + scan registration + odometry in
    combined mode + mapping, MAIN/src/vloam_main_node.cpp:125-180) advanced by one launch chain per frame — every session equal to
    the same sequence through vloam_process_frame on a handle of its own to round-off (trajectory, VO trajectory, VO estimate, map)."""
    B, n = 4, 8
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
    seqs = [synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1, seed_scene=1234 + 17 * b, seed_traj=42 + b, seed_noise=5678 + 1000 * b)
            for b in range(B)]
    clouds = [[np.ascontiguousarray(s.sweep(k), dtype=np.float32) for k in range(n)] for s in seqs]
    matches = [[synth.synth_matches(s, k) if k > 0 else (None, None) for k in range(n)] for s in seqs]

    def setup(h):
        h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
        h.set_extrinsics(base_T_cam0, velo_T_cam0)
        return h

    hb = setup(vl.Handle(0, n_sessions=B, with_mapping=1, detach_VO_LO=0))
    for k in range(n):
        hb.batch_process_frame([clouds[b][k] for b in range(B)], [matches[b][k] for b in range(B)])
    hb.sync()
    for b in range(B):
        hs = setup(vl.Handle(0, with_mapping=1, detach_VO_LO=0))
        for k in range(n):
            hs.process_frame(clouds[b][k], matches[b][k][0], matches[b][k][1])
        hs.sync()
        hb.select(b)
        assert same_poses(hb.trajectory(), hs.trajectory()), "session %d trajectory" % b
        assert same_poses(hb.vo_trajectory(), hs.vo_trajectory()), "session %d VO trajectory" % b
        rb, rs = hb.vo_result(), hs.vo_result()
        assert np.allclose(rb["angles"], rs["angles"], rtol=0, atol=POSE_TOL) and np.allclose(rb["t"], rs["t"], rtol=0, atol=POSE_TOL) and rb["counter32"] == rs["counter32"] > 100
        assert same_map(hb.get_map(), hs.get_map()), "session %d map" % b
        hs.close()
    hb.close()


@pytest.mark.parametrize("orb", [False, True], ids=["optical_flow", "orb_brute_force"])
def test_batched_frames_from_raw_images_equal_single_session_runs(vl, synth, orb):
    """vloam_batch_process_frame_image: every session gets its own sweep AND its own grey image (corners + pyramidal LK — or, with an ORB pattern
    set, ORB descriptors + brute-force matches — on the device feed the session's VO solve).  Each session must equal the same inputs through
    vloam_process_frame_image on a handle of its own: trajectory, VO trajectory, key points, matches."""
    B, n, W, H = 3, 4, 1242, 375
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
    seqs = [synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1, seed_scene=1234 + 17 * b, seed_traj=42 + b, seed_noise=5678 + 1000 * b)
            for b in range(B)]
    clouds = [[np.ascontiguousarray(s.sweep(k), dtype=np.float32) for k in range(n)] for s in seqs]
    images = [[synth.render_image(s, k, width=W, height=H) for k in range(n)] for s in seqs]

    def setup(h):
        h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
        h.set_extrinsics(base_T_cam0, velo_T_cam0)
        if orb:
            h.vo_set_orb_pattern(synth.orb_test_pattern())
        return h

    hb = setup(vl.Handle(0, n_sessions=B, with_mapping=1, detach_VO_LO=0, image_width=W, image_height=H))
    for k in range(n):
        hb.batch_process_frame_image([clouds[b][k] for b in range(B)], [images[b][k] for b in range(B)])
    hb.sync()
    for b in range(B):
        hs = setup(vl.Handle(0, with_mapping=1, detach_VO_LO=0, image_width=W, image_height=H))
        for k in range(n):
            hs.process_frame_image(clouds[b][k], images[b][k])
        hs.sync()
        hb.select(b)
        assert same_poses(hb.trajectory(), hs.trajectory()), "session %d trajectory" % b
        assert same_poses(hb.vo_trajectory(), hs.vo_trajectory()), "session %d VO trajectory" % b
        assert np.array_equal(hb.vo_keypoints(), hs.vo_keypoints()) and hs.vo_keypoints().shape[0] > 50, "session %d key points" % b
        mb, ms = hb.vo_flow_matches(), hs.vo_flow_matches()
        assert np.array_equal(mb[0], ms[0]) and np.array_equal(mb[1], ms[1]) and ms[0].shape[0] > 20, "session %d matches" % b
        if orb:
            db, ds = hb.vo_descriptors(), hs.vo_descriptors()
            assert np.array_equal(db[0], ds[0]) and np.array_equal(db[1], ds[1]), "session %d descriptors" % b
        hs.close()
    # distinct sessions really had distinct images
    hb.select(0); k0 = hb.vo_keypoints()
    hb.select(1); k1 = hb.vo_keypoints()
    assert k0.shape != k1.shape or not np.array_equal(k0, k1)
    hb.close()
