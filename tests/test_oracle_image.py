"""CPU: the oracle's restatement of the image front-end (cv::goodFeaturesToTrack / cv::calcOpticalFlowPyrLK as the reference calls
them, image_util.cpp:13-36, 351-372) behaves like the published algorithms on synthetic images with a known warp."""
import numpy as np


def test_good_features_properties(orc, synth):
    img, _, _ = synth.synth_image_pair(640, 200, seed=11)
    c, eig = orc.good_features(img, want_eig=True)
    assert 50 < c.shape[0] <= 1024 and eig.shape == img.shape
    xi, yi = c[:, 0].astype(int), c[:, 1].astype(int)
    assert np.all(c == np.stack([xi, yi], 1))                                  # integer pixel positions
    assert xi.min() >= 1 and xi.max() <= 638 and yi.min() >= 1 and yi.max() <= 198   # the 1-pixel frame is never a corner
    v = eig[yi, xi]
    assert np.all(np.diff(v) <= 0)                                             # strongest first
    assert v.min() > np.float32(np.float64(eig.max()) * 0.03)                  # quality level
    for k in range(c.shape[0]):                                                # 3 x 3 local maxima
        assert v[k] == eig[yi[k] - 1:yi[k] + 2, xi[k] - 1:xi[k] + 2].max()
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=2) + 1e9 * np.eye(c.shape[0])
    assert d.min() >= 7.5                                                      # minDistance = block_size * 1.5
    # greedy: every local maximum above the threshold that was NOT taken lies closer than minDistance to a stronger taken corner
    thr = np.float32(np.float64(eig.max()) * 0.03)
    taken = set(zip(xi.tolist(), yi.tolist()))
    rng = np.random.default_rng(0)
    ys, xs = np.nonzero(eig[1:-1, 1:-1] > thr)
    for j in rng.choice(ys.size, size=400, replace=False):
        x, y = int(xs[j]) + 1, int(ys[j]) + 1
        if (x, y) in taken or eig[y, x] != eig[y - 1:y + 2, x - 1:x + 2].max():
            continue
        dd = np.hypot(c[:, 0] - x, c[:, 1] - y)
        blockers = (dd < 7.5) & (v >= eig[y, x])
        assert blockers.any(), (x, y)
    # maxCorners cuts the sorted list
    c16 = orc.good_features(img, max_corners=16)
    assert np.array_equal(c16, c[:16])
    assert orc.good_features(np.full((64, 64), 77, dtype=np.uint8)).shape[0] == 0   # no gradient, no corner


def test_pyramid_levels(orc, synth):
    img, _, _ = synth.synth_image_pair(321, 97, seed=5)
    lv = orc.pyramid_levels(img)
    assert [a.shape for a, _ in lv] == [(97, 321), (49, 161), (25, 81)]
    assert np.array_equal(lv[0][0], img)
    # pyrDown = 5-tap binomial in both directions, rounded: compare an interior pixel with the closed form
    k = np.array([1, 4, 6, 4, 1])
    y, x = 20, 33
    ref = (np.outer(k, k) * img[2 * y - 2:2 * y + 3, 2 * x - 2:2 * x + 3].astype(int)).sum()
    assert lv[1][0][y, x] == (ref + 128) >> 8
    # Scharr: Ix = 3 / 10 / 3 smoothing of the horizontal difference
    a = img.astype(int)
    y, x = 40, 100
    ix = 3 * (a[y - 1, x + 1] - a[y - 1, x - 1]) + 10 * (a[y, x + 1] - a[y, x - 1]) + 3 * (a[y + 1, x + 1] - a[y + 1, x - 1])
    iy = 3 * (a[y + 1, x - 1] - a[y - 1, x - 1]) + 10 * (a[y + 1, x] - a[y - 1, x]) + 3 * (a[y + 1, x + 1] - a[y - 1, x + 1])
    assert lv[0][1][y, x, 0] == ix and lv[0][1][y, x, 1] == iy
    small = orc.pyramid_levels(img[:40, :60])     # the next level must be larger than the 15 x 15 window
    assert len(small) == 2 and small[1][0].shape == (20, 30)


def test_pyr_lk_recovers_a_known_warp(orc, synth):
    prev, nxt, flow = synth.synth_image_pair(640, 200, seed=3, shift=(5.2, -2.4), rot=0.006, scale=1.004)
    c = orc.good_features(prev)
    out, st = orc.pyr_lk(prev, nxt, c)
    assert st.mean() > 0.97
    err = np.linalg.norm(out - flow(c), axis=1)[st == 1]
    assert np.median(err) < 0.06 and np.percentile(err, 95) < 0.3
    # identical images: nothing moves (the first iteration's update is below epsilon)
    same, st2 = orc.pyr_lk(prev, prev, c)
    assert np.abs(same - c).max() < 1e-3 and st2.all()
    # a point whose window leaves the image loses its status; a flat patch fails the min-eigenvalue test
    flat = np.full((200, 640), 128, dtype=np.uint8)
    _, st3 = orc.pyr_lk(flat, flat, np.array([[100.0, 100.0], [700.0, 50.0]], dtype=np.float32))
    assert st3.tolist() == [0, 0]
    pu, cu = orc.flow_matches(c, out, st)
    assert pu.dtype == np.int32 and pu.shape == cu.shape and pu.shape[0] == int(st.sum())
    assert np.array_equal(pu, c[st == 1].astype(np.int32)) and np.all(np.abs(cu - out[st == 1]) < 1.0)
