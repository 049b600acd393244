"""GPU: the HIP image front-end (corners + pyramidal Lucas-Kanade, optical_flow_match configuration of visual_odometry.cpp:91-132)
against the oracle's restatement on the same images, through the C ABI: corners, their order, the eigenvalue map and the pyramids are
bit-exact (integer / correctly rounded arithmetic); the tracked positions are f32 results of the same operation sequence."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(96, 40, 5), (320, 96, 1), (641, 203, 2), (1242, 375, 3)])   # 96 x 40: a 20-row top level, two levels only
def test_image_frontend_parity(vl, orc, synth, w, h, seed):
    prev, nxt, _ = synth.synth_image_pair(w, h, seed=seed, shift=(4.3, -1.7), rot=0.005, scale=1.003)
    hd = vl.Handle(0, with_mapping=0, image_width=w, image_height=h)
    hd.vo_process_image(prev)
    c0 = hd.vo_keypoints()
    ref0, eig0 = orc.good_features(prev, want_eig=True)
    eig_d, lv_d = hd.img_debug(w, h)
    assert np.array_equal(eig_d, eig0), np.abs(eig_d - eig0).max()
    assert np.array_equal(c0, ref0), (c0.shape, ref0.shape)
    lv_o = orc.pyramid_levels(prev)
    assert len(lv_d) == len(lv_o)
    for (a, da), (b, db) in zip(lv_d, lv_o):
        assert np.array_equal(a, b) and np.array_equal(da, db)
    a, b, st = hd.vo_flow()
    assert a.shape[0] == 0                                 # first image: nothing to track from
    hd.vo_process_image(nxt)
    c1 = hd.vo_keypoints()
    ref1 = orc.good_features(nxt)
    assert np.array_equal(c1, ref1)
    a, b, st = hd.vo_flow()
    out, st_o = orc.pyr_lk(prev, nxt, ref1)                # the NEW image's corners, tracked from the previous image into the new one
    assert np.array_equal(a, ref1) and np.array_equal(st, st_o)
    assert np.array_equal(b, out), np.abs(b - out).max()
    pu, cu = hd.vo_flow_matches()
    pu_o, cu_o = orc.flow_matches(ref1, out, st_o)
    assert np.array_equal(pu, pu_o) and np.array_equal(cu, cu_o)
    if w == 1242:
        assert c1.shape[0] == 1024                          # maxCorners cut exercised
    hd.close()


@pytest.mark.gpu
def test_image_frontend_edge_cases(vl, orc):
    hd = vl.Handle(0, with_mapping=0, image_width=256, image_height=128)
    flat = np.full((128, 256), 90, dtype=np.uint8)
    hd.vo_process_image(flat)
    assert hd.vo_keypoints().shape[0] == 0                  # no gradient: no corner, and the flow of nothing is nothing
    hd.vo_process_image(flat)
    assert hd.vo_flow()[0].shape[0] == 0 and hd.vo_flow_matches()[0].shape[0] == 0
    with pytest.raises(vl.VloamError) as e:
        hd.vo_process_image(np.zeros((64, 256), dtype=np.uint8))   # one size per sequence
    assert e.value.status == vl.ERR_INVALID
    hd.close()
    # same area, other aspect ratio: the pyramid levels of 128 x 256 do not fit the buffers laid out for 256 x 128 — refused, not overrun
    hd = vl.Handle(0, with_mapping=0, image_width=256, image_height=128)
    with pytest.raises(vl.VloamError) as e:
        hd.vo_process_image(np.zeros((256, 128), dtype=np.uint8))
    assert e.value.status == vl.ERR_INVALID
    hd.vo_process_image(np.zeros((96, 160), dtype=np.uint8))   # smaller in both dimensions is fine
    hd.close()
    # a corner tracked out of the image: a bright square near the right border that moves out
    a = np.full((96, 160), 40, dtype=np.uint8)
    b = a.copy()
    a[30:50, 120:140] = 220
    b[30:50, 150:160] = 220
    h2 = vl.Handle(0, with_mapping=0, image_width=160, image_height=96)
    h2.vo_process_image(a)
    h2.vo_process_image(b)
    pa, pb, st = h2.vo_flow()
    out, st_o = orc.pyr_lk(a, b, orc.good_features(b))
    assert np.array_equal(st, st_o) and np.array_equal(pb, out)
    h2.close()
    h3 = vl.Handle(0, with_mapping=0)                       # no image capacity configured
    with pytest.raises(vl.VloamError) as e:
        h3.vo_process_image(flat)
    assert e.value.status == vl.ERR_ORDER
    h3.close()


@pytest.mark.gpu
def test_coupled_frame_loop_from_raw_images(vl, synth):
    """configs[3] from raw inputs: every frame hands a sweep AND a grey image to vloam_process_frame_image; corners, flow, depth map,
    VO solve, VO -> LO prior, scan registration, odometry, LO -> VO prior and mapping all stay on the device.  Same frames through
    the oracle (image restatement + coupled pipeline): matches identical, poses <= 1e-8."""
    from test_gpu_laser_odometry import qdist
    from test_gpu_vloam import make
    nframes, W, H = 7, 1242, 375
    seq = synth.SynthSequence(n_rings=64, n_azimuth=1024, n_sweeps=nframes + 1)
    h, o = make(vl, synth, detach=False, with_mapping=1, image_width=W, image_height=H)
    frames = [(seq.sweep(k), synth.render_image(seq, k, W, H)) for k in range(nframes)]
    for k in range(nframes):
        cloud, img = frames[k]
        h.process_frame_image(cloud, img)
        assert o.process_image(cloud, img) == 0
        assert np.array_equal(h.vo_keypoints(), o.keypoints)
        if k > 0:
            a, b, st = h.vo_flow()
            assert np.array_equal(a, o.flow[0]) and np.array_equal(b, o.flow[1]) and np.array_equal(st, o.flow[2])
            r, v = h.vo_result(), o.vo_result
            assert (r["counter32"], r["counter22"]) == (v["counter32"], v["counter22"]) and r["counter32"] + r["counter22"] > 100
            tol = 2e-7 if k == 1 else 1e-8
            assert np.linalg.norm(r["angles"] - v["angles"]) < tol and np.linalg.norm(r["t"] - v["t"]) < tol, "VO estimate, frame %d" % k
        tol = 1e-6 if k == 1 else 1e-8 * (k + 1)
        tj = h.trajectory()[k]
        qw, tw, _, _ = o.lidar.lo_pose()
        qm, tm = o.lidar.map_published_pose()
        assert qdist(tj[0:4], qw) < tol and np.linalg.norm(tj[4:7] - tw) < tol, "LO world pose, frame %d" % k
        assert qdist(tj[7:11], qm) < tol and np.linalg.norm(tj[11:14] - tm) < tol, "map pose, frame %d" % k
        vq, vt = o.vo_world_pose()
        vj = h.vo_trajectory()[k]
        assert qdist(vj[0:4], vq) < tol and np.linalg.norm(vj[4:7] - vt) < tol, "world_VOT_base_last, frame %d" % k
    # the VO chain built from tracked image corners follows the LiDAR chain (geometry-consistent images)
    assert np.linalg.norm(h.vo_trajectory()[nframes - 1][4:7] - h.trajectory()[nframes - 1][4:7]) < 0.6
    # a frame with a wrong image size is refused before anything of it is enqueued: the handle carries on with the next good frame
    with pytest.raises(vl.VloamError) as e:
        h.process_frame_image(frames[0][0], frames[0][1][:200])
    assert e.value.status == vl.ERR_INVALID and h.frame_count() == nframes
    # the same frames streamed without reading anything back in between give the same trajectory
    h2, _ = make(vl, synth, detach=False, with_mapping=1, image_width=W, image_height=H)
    for k in range(nframes):
        cloud, img = frames[k][0].copy(), np.pad(frames[k][1], ((0, 0), (0, 38)))   # padded rows (stride != width); buffers die right after the call
        h2.L.vloam_process_frame_image(h2.h, cloud.ctypes.data_as(vl.C.c_void_p), cloud.shape[0], img.ctypes.data_as(vl.C.c_void_p), W, H, img.shape[1])
        cloud[:] = 0; img[:] = 0
        del cloud, img
    h2.sync()
    assert np.array_equal(h2.trajectory(), h.trajectory()) and np.array_equal(h2.vo_trajectory(), h.vo_trajectory())
    h.close(); h2.close()


@pytest.mark.gpu
def test_image_frontend_capacity_is_reported(vl, orc):
    """More 3 x 3 local maxima above the quality threshold than the candidate list holds: VLOAM_ERR_CAPACITY, not a silent cut."""
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, size=(1024, 2048), dtype=np.uint8)
    assert orc.good_features(noise, max_corners=0).shape[0] > 0          # the oracle itself has no such limit
    h = vl.Handle(0, with_mapping=0, image_width=2048, image_height=1024)
    h.vo_process_image(noise)
    with pytest.raises(vl.VloamError) as e:
        h.vo_keypoints()
    assert e.value.status == vl.ERR_CAPACITY and "candidates" in str(e.value)
    with pytest.raises(vl.VloamError) as e:     # vloam_sync reports it as well (the coupled loop has no getter in its path)
        h.sync()
    assert e.value.status == vl.ERR_CAPACITY and "image front-end" in str(e.value)
    h.close()
    # the same texture at a size whose candidates fit is processed normally (and equals the oracle)
    h = vl.Handle(0, with_mapping=0, image_width=512, image_height=256)
    h.vo_process_image(noise[:256, :512])
    assert np.array_equal(h.vo_keypoints(), orc.good_features(noise[:256, :512]))
    h.close()


@pytest.mark.gpu
def test_bf_hamming_matcher_parity(vl, orc):
    """vloam_vo_match_descriptors == the oracle's ImageUtil::matchDescriptors (BF, NORM_HAMMING), both selector types, ties included."""
    from test_oracle_image import _rand_desc
    rng = np.random.default_rng(1)
    h = vl.Handle(0, with_mapping=0, image_width=64, image_height=64)
    for (n0, n1, nbytes) in ((300, 250, 32), (1024, 1024, 32), (77, 500, 64), (5, 1, 32)):
        a, b = _rand_desc(rng, max(n0, 12), max(n1, 8), nbytes)
        a, b = a[:n0], b[:n1]
        for knn in (True, False):
            q, t = h.vo_match_descriptors(a, b, knn)
            qo, to = orc.bf_match_hamming(a, b, knn)
            assert np.array_equal(q, qo) and np.array_equal(t, to), (n0, n1, nbytes, knn)
    q, t = h.vo_match_descriptors(np.zeros((0, 32), dtype=np.uint8), b, True)
    assert q.size == 0
    with pytest.raises(vl.VloamError) as e:
        h.vo_match_descriptors(np.zeros((4, 30), dtype=np.uint8), np.zeros((4, 30), dtype=np.uint8))
    assert e.value.status == vl.ERR_INVALID
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(320, 96, 1), (333, 101, 2), (1242, 375, 3)])
def test_clahe_parity(vl, orc, synth, w, h, seed):
    """cfg.CLAHE = 1: the equalised image is bit-identical to the oracle's cv::CLAHE restatement, and corners / flow are those of the
    equalised images (visual_odometry.cpp:97-105)."""
    prev, nxt, _ = synth.synth_image_pair(w, h, seed=seed, shift=(3.3, 1.4))
    hd = vl.Handle(0, with_mapping=0, image_width=w, image_height=h, CLAHE=1)
    hd.vo_process_image(prev)
    e0 = orc.clahe(prev)
    assert np.array_equal(hd.debug_raw(4, 11, np.uint8).reshape(h, w), e0)
    assert np.array_equal(hd.vo_keypoints(), orc.good_features(e0))
    hd.vo_process_image(nxt)
    e1 = orc.clahe(nxt)
    c1 = orc.good_features(e1)
    out, st = orc.pyr_lk(e0, e1, c1)
    a, b, s = hd.vo_flow()
    assert np.array_equal(a, c1) and np.array_equal(b, out) and np.array_equal(s, st)
    hd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(320, 96, 1), (641, 203, 2), (1242, 375, 3)])
def test_orb_configuration_parity(vl, orc, synth, w, h, seed):
    """optical_flow_match = false — the reference's launch default (vloam_main.launch:10): ORB descriptors on the Shi-Tomasi corners
    (image_util.cpp:162-212) + brute-force Hamming 2-NN with the 0.8 ratio test (:214-296), the sampling pattern handed in by the caller.
    Blurred image, border-filtered keypoints and their order, all 256 bits of every descriptor, the matches and the match loop's integer
    pixel pairs: bit for bit against the oracle."""
    pat = synth.orb_test_pattern()
    prev, nxt, _ = synth.synth_image_pair(w, h, seed=seed, shift=(2.0, -1.0), rot=0.002, scale=1.001)
    hd = vl.Handle(0, with_mapping=0, image_width=w, image_height=h)
    hd.vo_set_orb_pattern(pat)
    ref = []
    for k, img in enumerate((prev, nxt)):
        hd.vo_process_image(img)
        corners = orc.good_features(img)
        assert np.array_equal(hd.vo_keypoints(), corners)
        kept, desc = orc.orb_descriptors(img, corners, pat)
        xy, d = hd.vo_descriptors()
        assert np.array_equal(xy, corners[kept]) and kept.size < corners.shape[0], "border filter, order kept"
        assert np.array_equal(d, desc), "descriptor bits"
        ref.append((corners, kept, desc))
        pu, cu = hd.vo_flow_matches()
        assert hd.vo_flow()[0].shape[0] == 0, "nothing is tracked in the ORB configuration (calcOpticalFlowPyrLK is not called, visual_odometry.cpp:105-116)"
        if k == 0:
            assert pu.shape[0] == 0
        else:
            pu_o, cu_o = orc.orb_matches(ref[0][0], ref[0][1], ref[0][2], corners, kept, desc)
            assert np.array_equal(pu, pu_o) and np.array_equal(cu, cu_o) and pu.shape[0] > 10
            assert np.median(np.abs((cu - pu)[:, 0] - 2)) <= 1 and np.median(np.abs((cu - pu)[:, 1] + 1)) <= 1, "the matches follow the 2 px / -1 px shift"
    with pytest.raises(vl.VloamError) as e:
        hd.vo_set_orb_pattern(None)          # not in the middle of a sequence
    assert e.value.status == vl.ERR_ORDER
    hd.close()
    bad = pat.copy()
    bad[7] = [40, 0, 1, 1]                   # a test point outside the 31-pixel border the keypoints keep
    hb = vl.Handle(0, with_mapping=0, image_width=w, image_height=h)
    with pytest.raises(vl.VloamError) as e:
        hb.vo_set_orb_pattern(bad)
    assert e.value.status == vl.ERR_INVALID
    hb.close()


@pytest.mark.gpu
def test_coupled_frame_loop_with_orb_matches(vl, synth):
    """configs[3] in the reference's DEFAULT image configuration, from raw inputs: sweep + grey image per frame, corners -> ORB -> brute-force
    matches -> depth-enhanced VO -> VO prior -> scan registration / odometry / mapping, all on the device, against the coupled-loop oracle."""
    from test_gpu_laser_odometry import qdist
    from test_gpu_vloam import make
    nframes, W, H = 6, 1242, 375
    pat = synth.orb_test_pattern()
    seq = synth.SynthSequence(n_rings=64, n_azimuth=1024, n_sweeps=nframes + 1)
    h, o = make(vl, synth, detach=False, with_mapping=1, image_width=W, image_height=H)
    h.vo_set_orb_pattern(pat)
    o.orb_pattern = pat
    for k in range(nframes):
        cloud, img = seq.sweep(k), synth.render_image(seq, k, W, H)
        h.process_frame_image(cloud, img)
        assert o.process_image(cloud, img) == 0
        xy, d = h.vo_descriptors()
        assert np.array_equal(xy, o.orb[0]) and np.array_equal(d, o.orb[1])
        if k > 0:
            r, v = h.vo_result(), o.vo_result
            assert (r["counter32"], r["counter22"]) == (v["counter32"], v["counter22"]) and r["counter32"] + r["counter22"] > 50
            tol = 2e-7 if k == 1 else 1e-8
            assert np.linalg.norm(r["angles"] - v["angles"]) < tol and np.linalg.norm(r["t"] - v["t"]) < tol, "VO estimate, frame %d" % k
        tol = 1e-6 if k == 1 else 1e-8 * (k + 1)
        tj = h.trajectory()[k]
        qw, tw, _, _ = o.lidar.lo_pose()
        qm, tm = o.lidar.map_published_pose()
        assert qdist(tj[0:4], qw) < tol and np.linalg.norm(tj[4:7] - tw) < tol, "LO world pose, frame %d" % k
        assert qdist(tj[7:11], qm) < tol and np.linalg.norm(tj[11:14] - tm) < tol, "map pose, frame %d" % k
    h.close()
