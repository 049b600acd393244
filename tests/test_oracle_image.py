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


# ---- a second, independent transcription in numpy (array formulation, written from OpenCV's algorithm description, not from
# oracle/orc_img.cpp) must reproduce the C++ restatement exactly
def _np_good_features(img, max_corners=1024, quality=0.03, min_distance=7.5, block=5):
    from scipy.ndimage import maximum_filter
    a = np.pad(img.astype(np.int64), 1, mode="reflect")                      # BORDER_REFLECT_101
    dx = (a[:-2, 2:] - a[:-2, :-2]) + 2 * (a[1:-1, 2:] - a[1:-1, :-2]) + (a[2:, 2:] - a[2:, :-2])
    dy = (a[2:, :-2] - a[:-2, :-2]) + 2 * (a[2:, 1:-1] - a[:-2, 1:-1]) + (a[2:, 2:] - a[:-2, 2:])
    r = block // 2

    def box(p):
        q = np.pad(p, r, mode="reflect")
        c = np.cumsum(np.cumsum(np.pad(q, ((1, 0), (1, 0))), axis=0), axis=1)   # exact integers
        return c[block:, block:] - c[:-block, block:] - c[block:, :-block] + c[:-block, :-block]
    sxx, sxy, syy = box(dx * dx), box(dx * dy), box(dy * dy)
    scale = 1.0 / (4.0 * block * 255.0)
    d = sxx - syy
    eig = (((sxx + syy).astype(np.float64) - np.sqrt((d * d + 4 * sxy * sxy).astype(np.float64))) * (0.5 * scale * scale)).astype(np.float32)
    thr = np.float32(np.float64(max(eig.max(), np.float32(0))) * quality)
    tz = np.where(eig > thr, eig, np.float32(0))
    dil = maximum_filter(tz, size=3, mode="constant", cval=0.0)
    ok = (tz != 0) & (tz == dil)
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    ys, xs = np.nonzero(ok)
    addr = ys * img.shape[1] + xs
    order = np.lexsort((-addr, -eig[ys, xs].astype(np.float64)))            # value descending, then the larger address first
    out = []
    md2 = np.float32(min_distance * min_distance)
    for j in order:
        x, y = int(xs[j]), int(ys[j])
        if all((np.float32(x - u) * np.float32(x - u) + np.float32(y - v) * np.float32(y - v)) >= md2 for u, v in out):
            out.append((x, y))
            if len(out) == max_corners:
                break
    return np.array(out, dtype=np.float32).reshape(-1, 2), eig


def test_good_features_vs_numpy_transcription(orc, synth):
    for (w, h, seed) in ((320, 96, 4), (401, 131, 9)):
        img, _, _ = synth.synth_image_pair(w, h, seed=seed)
        c, eig = orc.good_features(img, want_eig=True)
        c_np, eig_np = _np_good_features(img)
        assert np.array_equal(eig, eig_np)
        assert np.array_equal(c, c_np)


def _np_pyr_down(a):
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    h, w = a.shape
    nh, nw = (h + 1) // 2, (w + 1) // 2
    p = np.pad(a.astype(np.int64), 2, mode="reflect")
    if p.shape[0] < 2 * nh + 3:
        p = np.pad(p, ((0, 2 * nh + 3 - p.shape[0]), (0, 0)), mode="reflect")
    if p.shape[1] < 2 * nw + 3:
        p = np.pad(p, ((0, 0), (0, 2 * nw + 3 - p.shape[1])), mode="reflect")
    rows = sum(k[i] * p[:, i:i + 2 * nw:2] for i in range(5))
    full = sum(k[j] * rows[j:j + 2 * nh:2, :] for j in range(5))
    return ((full + 128) >> 8).astype(np.uint8)


def _np_scharr(a):
    p = np.pad(a.astype(np.int64), 1, mode="reflect")
    t0 = 3 * (p[:-2] + p[2:]) + 10 * p[1:-1]          # vertical smoothing, all columns incl. the padded ones
    t1 = p[2:] - p[:-2]
    ix = t0[:, 2:] - t0[:, :-2]
    iy = 3 * (t1[:, 2:] + t1[:, :-2]) + 10 * t1[:, 1:-1]
    return np.stack([ix, iy], -1).astype(np.int16)


def _np_lk_point(I, J, dI, pt, guess, level, top, win=15):
    """One point on one level (lkpyramid.cpp LKTrackerInvoker).  Returns (next point, lost_at_this_level)."""
    h, w = I.shape
    half = np.float32((win - 1) * 0.5)
    f32 = np.float32
    inv = f32(1.0 / (1 << level))
    px, py = f32(pt[0]) * inv, f32(pt[1]) * inv
    nx, ny = (px, py) if level == top else (f32(guess[0]) * f32(2), f32(guess[1]) * f32(2))
    out = (nx, ny)
    px, py = px - half, py - half
    ipx, ipy = int(np.floor(px)), int(np.floor(py))
    if ipx < -win or ipx >= w or ipy < -win or ipy >= h:
        return out, True
    R = win                                               # REFLECT_101 border of winSize pixels; zero border for the derivatives
    Ip = np.pad(I.astype(np.int64), R + 1, mode="reflect")
    Jp = np.pad(J.astype(np.int64), R + 1, mode="reflect")
    Dp = np.pad(dI.astype(np.int64), ((R + 1, R + 1), (R + 1, R + 1), (0, 0)))

    def weights(a, b):
        one = f32(1)
        w00 = int(np.rint((one - a) * (one - b) * f32(16384)))
        w01 = int(np.rint(a * (one - b) * f32(16384)))
        w10 = int(np.rint((one - a) * b * f32(16384)))
        return w00, w01, w10, 16384 - w00 - w01 - w10

    def interp(P, x0, y0, ws, n):
        s = P[y0 + R + 1:y0 + R + 1 + win, x0 + R + 1:x0 + R + 1 + win] * ws[0] + P[y0 + R + 1:y0 + R + 1 + win, x0 + R + 2:x0 + R + 2 + win] * ws[1] + \
            P[y0 + R + 2:y0 + R + 2 + win, x0 + R + 1:x0 + R + 1 + win] * ws[2] + P[y0 + R + 2:y0 + R + 2 + win, x0 + R + 2:x0 + R + 2 + win] * ws[3]
        return (s + (1 << (n - 1))) >> n
    ws = weights(px - f32(ipx), py - f32(ipy))
    Iw = interp(Ip, ipx, ipy, ws, 9)
    Ix = interp(Dp[:, :, 0], ipx, ipy, ws, 14)
    Iy = interp(Dp[:, :, 1], ipx, ipy, ws, 14)
    sc = f32(1.0 / (1 << 20))
    A11, A12, A22 = f32(int((Ix * Ix).sum())) * sc, f32(int((Ix * Iy).sum())) * sc, f32(int((Iy * Iy).sum())) * sc
    D = A11 * A22 - A12 * A12
    min_eig = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + f32(4) * A12 * A12)) / f32(2 * win * win)
    if float(min_eig) < 1e-4 or D < np.finfo(np.float32).eps:
        return out, True
    D = f32(1) / D
    nx, ny = nx - half, ny - half
    pdx = pdy = f32(0)
    lost = False
    for j in range(10):
        inx, iny = int(np.floor(nx)), int(np.floor(ny))
        if inx < -win or inx >= w or iny < -win or iny >= h:
            lost = True
            break
        ws = weights(nx - f32(inx), ny - f32(iny))
        diff = interp(Jp, inx, iny, ws, 9) - Iw
        b1, b2 = f32(int((diff * Ix).sum())) * sc, f32(int((diff * Iy).sum())) * sc
        ddx, ddy = (A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D
        nx, ny = nx + ddx, ny + ddy
        out = (nx + half, ny + half)
        if float(ddx) * float(ddx) + float(ddy) * float(ddy) <= 0.03 * 0.03:
            break
        if j > 0 and abs(float(ddx + pdx)) < 0.01 and abs(float(ddy + pdy)) < 0.01:
            out = (out[0] - ddx * f32(0.5), out[1] - ddy * f32(0.5))
            break
        pdx, pdy = ddx, ddy
    if not lost and level == 0:
        ix, iy = int(np.floor(out[0] - half)), int(np.floor(out[1] - half))
        lost = ix < -win or ix >= w or iy < -win or iy >= h
    return out, lost


def test_pyramid_and_lk_vs_numpy_transcription(orc, synth):
    prev, nxt, _ = synth.synth_image_pair(320, 96, seed=6, shift=(6.4, -2.8), rot=0.01, scale=1.006)
    lv = orc.pyramid_levels(prev)
    P, N = [prev], [nxt]
    for _ in range(2):
        P.append(_np_pyr_down(P[-1])); N.append(_np_pyr_down(N[-1]))
    for l in range(3):
        assert np.array_equal(lv[l][0], P[l]) and np.array_equal(lv[l][1], _np_scharr(P[l]))
    c = orc.good_features(nxt)
    extra = np.array([[1.5, 2.25], [318.0, 94.0], [160.3, 0.2], [0.0, 50.0]], dtype=np.float32)   # windows hanging over every border
    pts = np.concatenate([c[:60], extra])
    out, st = orc.pyr_lk(prev, nxt, pts)
    for i, p in enumerate(pts):
        g, lost0 = (0, 0), False
        for level in (2, 1, 0):
            g, lost = _np_lk_point(P[level], N[level], _np_scharr(P[level]), p, g, level, 2)
            lost0 = lost if level == 0 else lost0
        assert np.float32(g[0]) == out[i, 0] and np.float32(g[1]) == out[i, 1], (i, p, g, out[i])
        assert int(not lost0) == int(st[i]), (i, p)


def test_golden_image_fixture(orc):
    """tests/golden/image_320x96_3frames.npz (tests/golden/make_golden.py): the oracle still reproduces its committed outputs."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_320x96_3frames.npz"))
    for k in range(3):
        c = orc.good_features(g["img_%d" % k])
        assert np.array_equal(c, g["corners_%d" % k])
        if k > 0:
            t, s = orc.pyr_lk(g["img_%d" % (k - 1)], g["img_%d" % k], c)
            assert np.array_equal(t, g["tracked_%d" % k]) and np.array_equal(s, g["status_%d" % k])


def _rand_desc(rng, n0, n1, nbytes):
    a = rng.integers(0, 256, (n0, nbytes), dtype=np.uint8)
    b = a[rng.permutation(n0)[:n1] % n0].copy() if n1 <= n0 else rng.integers(0, 256, (n1, nbytes), dtype=np.uint8)
    flip = rng.integers(0, 256, b.shape, dtype=np.uint8) & rng.integers(0, 256, b.shape, dtype=np.uint8) & rng.integers(0, 256, b.shape, dtype=np.uint8)
    b ^= flip
    b[3] = b[7]                      # duplicates: ties must resolve to the lower index
    a[5] = a[11]
    return a, b


def test_bf_hamming_matcher_vs_numpy(orc):
    """image_util.cpp:221-296 (BF, NORM_HAMMING): 2-NN + ratio 0.8 and NN + cross check against a stable-argsort numpy formulation."""
    rng = np.random.default_rng(0)
    for nbytes in (32, 64):
        a, b = _rand_desc(rng, 300, 250, nbytes)
        D = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
        o = np.argsort(D, axis=1, kind="stable")[:, :2]
        d0, d1 = D[np.arange(300), o[:, 0]].astype(np.float32), D[np.arange(300), o[:, 1]].astype(np.float32)
        keep = d0.astype(np.float64) < 0.8 * d1.astype(np.float64)
        q, t = orc.bf_match_hamming(a, b, True)
        assert np.array_equal(q, np.nonzero(keep)[0]) and np.array_equal(t, o[keep, 0]) and 100 < q.size < 300
        q2, t2 = orc.bf_match_hamming(a, b, False)
        bq, bt = np.argmin(D, axis=1), np.argmin(D, axis=0)
        k2 = bt[bq] == np.arange(300)
        assert np.array_equal(q2, np.nonzero(k2)[0]) and np.array_equal(t2, bq[k2])
    assert orc.bf_match_hamming(a, b[:1], True)[0].size == 0     # one train descriptor: no second neighbour, no match


def _np_clahe(img, clip_limit=2.0, tiles=8):
    """cv::CLAHE::apply in array form (integral histogram per tile, fancy-indexed LUT interpolation), independent of orc_img.cpp."""
    h, w = img.shape
    ext = img if (w % tiles == 0 and h % tiles == 0) else np.pad(img, ((0, tiles - h % tiles), (0, tiles - w % tiles)), mode="reflect")
    eh, ew = ext.shape
    th, tw = eh // tiles, ew // tiles
    area = tw * th
    scale = np.float32(255) / np.float32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    lut = np.zeros((tiles, tiles, 256), np.float32)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                exc = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = exc // 256
                res = exc - batch * 256
                hist = hist + batch
                if res:
                    hist[np.arange(0, 256, max(256 // res, 1))[:res]] += 1
            lut[ty, tx] = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * scale), 0, 255)
    f32 = np.float32
    ys = np.arange(h, dtype=f32) * f32(f32(1) / f32(th)) - f32(0.5)
    xs = np.arange(w, dtype=f32) * f32(f32(1) / f32(tw)) - f32(0.5)
    ty1, tx1 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    ya, xa = (ys - ty1.astype(f32))[:, None], (xs - tx1.astype(f32))[None, :]
    ya1, xa1 = f32(1) - ya, f32(1) - xa
    ty2, tx2 = np.minimum(ty1 + 1, tiles - 1), np.minimum(tx1 + 1, tiles - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = img.astype(int)
    l11, l12 = lut[ty1[:, None], tx1[None, :], v], lut[ty1[:, None], tx2[None, :], v]
    l21, l22 = lut[ty2[:, None], tx1[None, :], v], lut[ty2[:, None], tx2[None, :], v]
    return np.clip(np.rint((l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya), 0, 255).astype(np.uint8)


def test_clahe_vs_numpy_transcription(orc, synth):
    """cv::createCLAHE(2.0)->apply (visual_odometry.cpp:31,97-100): sizes that divide the 8 x 8 tile grid and sizes that do not (padding)."""
    for (w, h, seed) in ((320, 96, 1), (333, 101, 2), (1242, 375, 3)):
        img, _, _ = synth.synth_image_pair(w, h, seed=seed)
        out = orc.clahe(img)
        assert np.array_equal(out, _np_clahe(img))
        assert out.std() > img.std()                      # it does equalise
    flat = np.full((64, 64), 100, dtype=np.uint8)
    assert np.array_equal(orc.clahe(flat), _np_clahe(flat))


def numpy_gaussian_blur(img):
    """cv::GaussianBlur(img, (7, 7), 2, 2, BORDER_REFLECT_101) in OpenCV's fixed-point form, array-formulated (independent of oracle/orc_img.cpp):
    the Q8 kernel from the error-diffusion rule (softdouble arithmetic == IEEE f64 here), separable integer passes, one rounding."""
    x = np.arange(-3, 4, dtype=np.float64)
    g = np.exp(-(x * x) / (2.0 * 2.0 * 2.0))
    g /= g.sum()
    q, err = np.zeros(7, np.int64), 0.0
    for i in range(3):                       # getGaussianKernelFixedPoint_ED: outer taps first, the rounding error carried inward
        adj = g[i] * 256.0 + err
        v = int(np.rint(adj))
        err = adj - v
        q[i] = q[6 - i] = v
    q[3] = 256 - 2 * int(q[:3].sum())       # the centre takes what is left of 1.0
    assert list(q) == [18, 34, 48, 56, 48, 34, 18]
    a = img.astype(np.int64)
    h, w = a.shape
    p = np.pad(a, ((0, 0), (3, 3)), mode="reflect")
    rows = sum(q[k] * p[:, k:k + w] for k in range(7))
    p = np.pad(rows, ((3, 3), (0, 0)), mode="reflect")
    out = sum(q[k] * p[k:k + h, :] for k in range(7))
    return ((out + (1 << 15)) >> 16).astype(np.uint8)


def numpy_orb(img, kps, pattern):
    """ORB::compute on provided goodFeaturesToTrack keypoints (octave 0, angle -1), array-formulated: border filter, blur, steered tests."""
    h, w = img.shape
    kps = np.asarray(kps, np.float32).reshape(-1, 2)
    keep = (kps[:, 0] >= 31) & (kps[:, 0] < w - 31) & (kps[:, 1] >= 31) & (kps[:, 1] < h - 31)
    kept = np.nonzero(keep)[0]
    blur = numpy_gaussian_blur(img)
    ang = np.float32(-1.0) * np.float32(np.pi / 180.0)
    a, b = np.float32(np.cos(np.float64(ang))), np.float32(np.sin(np.float64(ang)))
    pts = np.asarray(pattern, np.int8).reshape(512, 2).astype(np.float32)
    ox = np.rint(pts[:, 0] * a - pts[:, 1] * b).astype(np.int64)
    oy = np.rint(pts[:, 0] * b + pts[:, 1] * a).astype(np.int64)
    cx = np.rint(kps[kept, 0]).astype(np.int64)[:, None]
    cy = np.rint(kps[kept, 1]).astype(np.int64)[:, None]
    v = blur[cy + oy[None, :], cx + ox[None, :]]            # [kept, 512]
    bits = (v[:, 0::2] < v[:, 1::2]).astype(np.uint8)      # [kept, 256]
    desc = np.packbits(bits.reshape(-1, 32, 8), axis=2, bitorder="little").reshape(-1, 32)
    return kept, desc


def test_orb_descriptors_vs_numpy_transcription_and_properties(orc, synth):
    """The ORB + brute-force configuration (optical_flow_match = false, the launch default): blur, border filter and the 256 steered tests of the
    C++ restatement against an array-formulated numpy transcription, bit for bit; and what the construction promises — keypoints within 31 px of
    the border are dropped (order kept), a featureless image gives the all-zero descriptor, every keypoint of an image matches ITSELF at distance 0
    against the same image and survives the ratio test, a 1-pixel-shifted copy of a smooth texture still matches most keypoints to their shifted
    selves."""
    pat = synth.orb_test_pattern()
    assert pat.shape == (256, 4) and np.abs(pat).max() <= 13
    prev, nxt, _ = synth.synth_image_pair(320, 160, seed=4)
    blur = orc.gaussian_blur(nxt)
    assert np.array_equal(blur, numpy_gaussian_blur(nxt))
    flat = np.full((64, 96), 77, np.uint8)
    assert np.array_equal(orc.gaussian_blur(flat), flat), "the Q8 kernel sums to exactly 256"
    kps = orc.good_features(nxt)
    extra = np.array([[5.0, 80.0], [300.0, 80.0], [160.0, 3.0], [31.0, 31.0], [288.0, 128.0], [289.0, 128.0], [30.75, 64.0]], np.float32)   # border cases of Rect::contains
    kps = np.concatenate([kps, extra])
    kept, desc = orc.orb_descriptors(nxt, kps, pat)
    nk, nd = numpy_orb(nxt, kps, pat)
    assert np.array_equal(kept, nk) and np.array_equal(desc, nd) and kept.size > 20
    n0 = kps.shape[0] - extra.shape[0]
    assert [int(k - n0) for k in kept if k >= n0] == [3, 4], "x in [31, w - 31): (31, 31) and (288, 128) stay, (289, .), (30.75, .) go"
    _, dflat = orc.orb_descriptors(flat, np.array([[48.0, 32.0]], np.float32), pat)
    assert dflat.shape == (1, 32) and not dflat.any()
    q, t = orc.bf_match_hamming(desc, desc, knn=True)
    uniq = np.array([np.count_nonzero((desc == d).all(axis=1)) == 1 for d in desc])
    assert np.array_equal(q[np.isin(q, np.nonzero(uniq)[0])], t[np.isin(q, np.nonzero(uniq)[0])]) and q.size >= 0.8 * uniq.sum()
    shifted = np.roll(nxt, 1, axis=1)
    k2 = orc.good_features(shifted)
    kept2, desc2 = orc.orb_descriptors(shifted, k2, pat)
    pu, cu = orc.orb_matches(kps, kept, desc, k2, kept2, desc2)
    assert pu.shape[0] > 10 and np.mean((cu[:, 0] - pu[:, 0] == 1) & (cu[:, 1] == pu[:, 1])) > 0.8
