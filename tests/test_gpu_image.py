"""GPU: the HIP image front-end (corners + pyramidal Lucas-Kanade, optical_flow_match configuration of visual_odometry.cpp:91-132)
against the oracle's restatement on the same images, through the C ABI: corners, their order, the eigenvalue map and the pyramids are
bit-exact (integer / correctly rounded arithmetic); the tracked positions are f32 results of the same operation sequence."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(320, 96, 1), (641, 203, 2), (1242, 375, 3)])
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
