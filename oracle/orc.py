"""ctypes driver for the CPU oracle (liborc.so).

ORACLE — TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  PARITY UNPINNED (see oracle/orc_math.h).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile the oracle with the committed Makefile (g++ only, no reference sources involved)."""
    so = os.path.join(_DIR, "liborc.so")
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".cpp", ".h"))]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _DIR, "-s", "all"])
    return so


_libs = {}


def lib(variant="liborc.so"):
    if variant not in _libs:
        build()
        L = C.CDLL(os.path.join(_DIR, variant))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_vo_create.restype = C.c_void_p
        L.orc_vo_query_depth.restype = C.c_float
        _libs[variant] = L
    return _libs[variant]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


F, I, D = C.c_float, C.c_int, C.c_double


class Oracle:
    """One LidarOdometryMapping-equivalent session on the CPU oracle."""

    def __init__(self, scan_line=64, minimum_range=5.0, line_res=0.4, plane_res=0.8, mapping_skip_frame=1,
                 detach_vo_lo=True, with_mapping=True, variant="liborc.so"):
        self.L = lib(variant)
        self.h = C.c_void_p(self.L.orc_create(scan_line, minimum_range, line_res, plane_res, mapping_skip_frame,
                                              int(detach_vo_lo), int(with_mapping)))

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    # ---- driving
    def process(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        return self.L.orc_process(self.h, _p(cloud, F), cloud.shape[0])

    def scan_registration(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        return self.L.orc_scan_registration(self.h, _p(cloud, F), cloud.shape[0])

    # ---- the façade stage by stage, hand-overs editable in between (lidar_odometry_mapping.cpp:73-154)
    def stage_sr(self, cloud):
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        return self.L.orc_stage_sr(self.h, c.ctypes.data_as(C.c_void_p), c.shape[0])

    def set_sr_cloud(self, which, pts):
        c = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
        assert self.L.orc_set_sr_cloud(self.h, which, c.ctypes.data_as(C.c_void_p), c.shape[0]) == 0

    def stage_lo(self):
        return self.L.orc_stage_lo(self.h)

    def stage_map(self, corner=None, surf=None, full=None, q=None, t=None):
        def arr(a):
            if a is None:
                return None, None, 0
            a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
            return a, a.ctypes.data_as(C.c_void_p), a.shape[0]
        keep = [arr(corner), arr(surf), arr(full)]
        qa = None if q is None else np.ascontiguousarray(q, dtype=np.float64)
        ta = None if t is None else np.ascontiguousarray(t, dtype=np.float64)
        return self.L.orc_stage_map(self.h, keep[0][1], keep[0][2], keep[1][1], keep[1][2], keep[2][1], keep[2][2],
                                    None if qa is None else qa.ctypes.data_as(C.c_void_p), None if ta is None else ta.ctypes.data_as(C.c_void_p))

    def set_vo_prior(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        self.L.orc_set_vo_prior(self.h, _p(q, D), _p(t, D))

    def stage_ms(self):
        ms = np.zeros(3)
        self.L.orc_stage_ms(self.h, _p(ms, D))
        return ms

    # ---- getters
    def cloud(self, which):
        n = self.L.orc_get_cloud(self.h, which, None, 0)
        buf = np.zeros((max(n, 1), 4), dtype=np.float32)
        self.L.orc_get_cloud(self.h, which, _p(buf, F), n)
        return buf[:n]

    def sr_ints(self, which):
        n = self.L.orc_get_sr_ints(self.h, which, None, 0)
        buf = np.zeros(max(n, 1), dtype=np.int32)
        self.L.orc_get_sr_ints(self.h, which, _p(buf, I), n)
        return buf[:n]

    def sr_curvature(self):
        n = self.L.orc_get_sr_curvature(self.h, None, 0)
        buf = np.zeros(max(n, 1), dtype=np.float32)
        self.L.orc_get_sr_curvature(self.h, _p(buf, F), n)
        return buf[:n]

    def sr_scalars(self):
        ori = np.zeros(2, dtype=np.float32)
        hp, n1 = I(0), I(0)
        self.L.orc_get_sr_scalars(self.h, _p(ori, F), C.byref(hp), C.byref(n1))
        return dict(startOri=ori[0], endOri=ori[1], halfPassedAt=hp.value, n_after_s1=n1.value)

    def lo_pose(self):
        qw, tw, ql, tl = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        self.L.orc_get_lo_pose(self.h, _p(qw, D), _p(tw, D), _p(ql, D), _p(tl, D))
        return qw, tw, ql, tl

    def lo_num_outer(self):
        return self.L.orc_lo_num_outer(self.h)

    def lo_corr(self, outer):
        nc, npl = I(0), I(0)
        if self.L.orc_get_lo_corr(self.h, outer, None, 0, C.byref(nc), None, 0, C.byref(npl)) != 0:
            return None
        c = np.zeros((max(nc.value, 1), 3), dtype=np.int32)
        p = np.zeros((max(npl.value, 1), 4), dtype=np.int32)
        self.L.orc_get_lo_corr(self.h, outer, _p(c, I), nc.value, C.byref(nc), _p(p, I), npl.value, C.byref(npl))
        return c[:nc.value], p[:npl.value]

    def _solve_dict(self, fn, outer, extra_counts=False):
        qi, qo = np.zeros(7), np.zeros(7)
        trace = np.zeros((128, 8))
        ni, term, nres = I(0), I(0), I(0)
        H0, g0, costs = np.zeros((6, 6)), np.zeros(6), np.zeros(2)
        res = np.zeros(1 << 17)
        args = [self.h, outer, _p(qi, D), _p(qo, D), _p(trace, D), 128, C.byref(ni), _p(H0, D), _p(g0, D), C.byref(term), _p(costs, D)]
        counts = np.zeros(2, dtype=np.int32)
        if extra_counts:
            args.append(_p(counts, I))
        args += [_p(res, D), res.shape[0], C.byref(nres)]
        if fn(*args) != 0:
            return None
        d = dict(q_in=qi[:4], t_in=qi[4:], q_out=qo[:4], t_out=qo[4:], trace=trace[:ni.value], H0=H0, g0=g0,
                 termination=term.value, initial_cost=costs[0], final_cost=costs[1], residuals0=res[:nres.value].copy())
        if extra_counts:
            d["corner_num"], d["surf_num"] = int(counts[0]), int(counts[1])
        return d

    def lo_solve(self, outer):
        return self._solve_dict(self.L.orc_get_lo_solve, outer)

    def map_pose(self):
        qw, tw, qm, tm = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        self.L.orc_get_map_pose(self.h, _p(qw, D), _p(tw, D), _p(qm, D), _p(tm, D))
        return qw, tw, qm, tm

    def map_published_pose(self):
        """(q, t) that LaserMapping::publish reports: the optimised pose, or the high-frequency pose on a skipped frame."""
        q, t = np.zeros(4), np.zeros(3)
        self.L.orc_get_map_published_pose(self.h, _p(q, D), _p(t, D))
        return q, t

    def map_num_outer(self):
        return self.L.orc_map_num_outer(self.h)

    def map_solve(self, outer):
        return self._solve_dict(self.L.orc_get_map_solve, outer, extra_counts=True)

    def map_factors(self, outer):
        s = self.map_solve(outer)
        nc, ns = s["corner_num"], s["surf_num"]
        ci, cab = np.zeros(max(nc, 1), dtype=np.int32), np.zeros((max(nc, 1), 6))
        si, spl = np.zeros(max(ns, 1), dtype=np.int32), np.zeros((max(ns, 1), 4))
        self.L.orc_get_map_factors(self.h, outer, _p(ci, I), _p(cab, D), nc, _p(si, I), _p(spl, D), ns)
        return ci[:nc], cab[:nc], si[:ns], spl[:ns]

    def map_info(self):
        cen = np.zeros(3, dtype=np.int32)
        tot = np.zeros(2, dtype=np.int64)
        valid = np.zeros(125, dtype=np.int32)
        nv = I(0)
        self.L.orc_get_map_info(self.h, _p(cen, I), _p(tot, C.c_longlong), _p(valid, I), 125, C.byref(nv))
        return dict(cen=cen, total_corner=int(tot[0]), total_surf=int(tot[1]), valid=valid[:nv.value].copy())

    def map_cube(self, which, cube):
        n = self.L.orc_get_map_cube(self.h, which, cube, None, 0)
        buf = np.zeros((max(n, 1), 4), dtype=np.float32)
        self.L.orc_get_map_cube(self.h, which, cube, _p(buf, F), n)
        return buf[:n]


# ---------------------------------------------------------------- standalone pieces
def voxel_grid(pts, leaf, variant="liborc.so"):
    L = lib(variant)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.zeros((max(pts.shape[0], 1), 4), dtype=np.float32)
    n = L.orc_voxel_grid(_p(pts, F), pts.shape[0], F(leaf), _p(out, F), out.shape[0])
    return out[:n]


def knn(pts, queries, k, use_tree=True):
    L = lib()
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    idx = np.zeros((q.shape[0], k), dtype=np.int32)
    d2 = np.zeros((q.shape[0], k), dtype=np.float32)
    L.orc_knn(_p(pts, F), pts.shape[0], _p(q, F), q.shape[0], k, _p(idx, I), _p(d2, F), int(use_tree))
    return idx, d2


def eval_lidar_factor(ftype, curr, geom, q, t, s=1.0):
    """Residual + tangent-space Jacobian of one LiDAR factor; s = the functors' interpolation ratio (lidarFactor.hpp:26-33)."""
    L = lib()
    curr, geom, q, t = [np.ascontiguousarray(a, dtype=np.float64) for a in (curr, geom, q, t)]
    r, J = np.zeros(3), np.zeros((3, 6))
    n = L.orc_eval_lidar_factor_s(ftype, _p(curr, D), _p(geom, D), _p(q, D), _p(t, D), D(s), _p(r, D), _p(J, D))
    return r[:n], J[:n]


def solve(factors, p0, p1, quaternion=True, huber_a=0.1, max_iters=4):
    """Run the Ceres-LM restatement on a list of factor rows (see orc_capi.cpp: orc_solve)."""
    L = lib()
    f = np.zeros((len(factors), 16))
    for i, row in enumerate(factors):
        f[i, :len(row)] = row
    p0 = np.array(p0, dtype=np.float64)
    p1 = np.array(p1, dtype=np.float64)
    trace = np.zeros((256, 8))
    ni, term = I(0), I(0)
    H0, g0, costs = np.zeros((6, 6)), np.zeros(6), np.zeros(2)
    rc = L.orc_solve(_p(f, D), f.shape[0], int(quaternion), D(huber_a), max_iters, _p(p0, D), _p(p1, D), _p(trace, D), 256,
                     C.byref(ni), _p(H0, D), _p(g0, D), C.byref(term), _p(costs, D))
    assert rc == 0
    return dict(p0=p0, p1=p1, trace=trace[:ni.value], H0=H0, g0=g0, termination=term.value, initial_cost=costs[0],
                final_cost=costs[1])


# ---- image front-end (orc_img.h): cv::goodFeaturesToTrack / cv::calcOpticalFlowPyrLK restated, reference parameters as defaults
U8 = C.c_ubyte


def good_features(img, max_corners=1024, quality=0.03, min_distance=7.5, block_size=5, want_eig=False):
    """image_util.cpp:13-36 — corners [n, 2] f32 (x, y) in acceptance order (and the min-eigenvalue map)."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = max(max_corners, 1) if max_corners > 0 else w * h
    xy = np.zeros((cap, 2), dtype=np.float32)
    eig = np.zeros((h, w), dtype=np.float32) if want_eig else None
    n = L.orc_good_features(_p(img, U8), w, h, max_corners, D(quality), D(min_distance), block_size, _p(xy, F), cap,
                            _p(eig, F) if want_eig else None)
    return (xy[:n], eig) if want_eig else xy[:n]


def pyramid_levels(img, win=15, max_level=2):
    """[(level image u8 [h, w], Scharr derivative int16 [h, w, 2])] of cv::buildOpticalFlowPyramid + calcSharrDeriv."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    nl = L.orc_pyramid_level(_p(img, U8), w, h, win, max_level, -1, None, None, None)
    out = []
    for lv in range(nl):
        im = np.zeros(h * w, dtype=np.uint8)
        de = np.zeros(h * w * 2, dtype=np.int16)
        wh = np.zeros(2, dtype=np.int32)
        L.orc_pyramid_level(_p(img, U8), w, h, win, max_level, lv, _p(im, U8), _p(de, C.c_short), _p(wh, I))
        lw, lh = int(wh[0]), int(wh[1])
        out.append((im[:lw * lh].reshape(lh, lw).copy(), de[:2 * lw * lh].reshape(lh, lw, 2).copy()))
    return out


def pyr_lk(prev, nxt, pts, win=15, max_level=2, max_count=10, epsilon=0.03):
    """image_util.cpp:351-372 — (next_pts [n, 2] f32, status [n] u8)."""
    L = lib()
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    assert prev.shape == nxt.shape
    h, w = prev.shape
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    out = np.zeros_like(pts)
    st = np.zeros(max(pts.shape[0], 1), dtype=np.uint8)
    L.orc_pyr_lk(_p(prev, U8), _p(nxt, U8), w, h, _p(pts, F), pts.shape[0], _p(out, F), _p(st, U8), win, max_level, max_count, D(epsilon))
    return out, st[:pts.shape[0]]


def clahe(img, clip_limit=2.0, tiles=8):
    """cv::createCLAHE(2.0)->apply (visual_odometry.cpp:31,97-100): uint8 [h, w] -> uint8 [h, w]."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros_like(img)
    L.orc_clahe(_p(img, U8), img.shape[1], img.shape[0], D(clip_limit), int(tiles), _p(out, U8))
    return out


def gaussian_blur(img):
    """cv::GaussianBlur(img, (7, 7), 2, 2, BORDER_REFLECT_101) in OpenCV's fixed-point form (orc_img.h)."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros_like(img)
    L.orc_gaussian_blur(_p(img, U8), img.shape[1], img.shape[0], _p(out, U8))
    return out


def orb_descriptors(img, kps, pattern):
    """image_util.cpp:162-212 with ORB on goodFeaturesToTrack keypoints: (kept indices into kps, descriptors [m, 32] u8)."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    k = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 2)
    pat = np.ascontiguousarray(pattern, dtype=np.int8).reshape(256, 4)
    cap = max(k.shape[0], 1)
    kept = np.zeros(cap, dtype=np.int32)
    desc = np.zeros((cap, 32), dtype=np.uint8)
    n = L.orc_orb_descriptors(_p(img, U8), img.shape[1], img.shape[0], _p(k, F), k.shape[0], pat.ctypes.data_as(C.c_void_p), _p(kept, I), _p(desc, U8), cap)
    return kept[:n], desc[:n]


def orb_matches(kps_prev, kept_prev, desc_prev, kps_curr, kept_curr, desc_curr):
    """visual_odometry.cpp:113-116,296-303 without optical flow: matchDescriptors(prev, curr) (BF, Hamming, 2-NN + ratio 0.8) -> the match loop's
    integer pixel pairs (prev_uv, curr_uv): keypoints[1 - i][queryIdx].pt / keypoints[i][trainIdx].pt truncated to int."""
    q, t = bf_match_hamming(desc_prev, desc_curr, knn=True)
    kp = np.asarray(kps_prev, np.float32)[np.asarray(kept_prev)]
    kc = np.asarray(kps_curr, np.float32)[np.asarray(kept_curr)]
    return kp[q].astype(np.int32), kc[t].astype(np.int32)


def bf_match_hamming(desc0, desc1, knn=True):
    """image_util.cpp:221-296 (BF, NORM_HAMMING): (queryIdx, trainIdx) int32 arrays; knn: 2-NN + ratio 0.8, else NN + cross check."""
    L = lib()
    a = np.ascontiguousarray(desc0, dtype=np.uint8)
    b = np.ascontiguousarray(desc1, dtype=np.uint8)
    assert a.ndim == 2 and b.ndim == 2 and a.shape[1] == b.shape[1]
    q = np.zeros(max(a.shape[0], 1), dtype=np.int32)
    t = np.zeros(max(a.shape[0], 1), dtype=np.int32)
    n = L.orc_bf_match_hamming(_p(a, U8), a.shape[0], _p(b, U8), b.shape[0], a.shape[1], int(knn), _p(q, I), _p(t, I), q.shape[0])
    return q[:n], t[:n]


def flow_matches(corners, tracked, status):
    """visual_odometry.cpp:296-308 with optical_flow_match: (prev_uv, curr_uv) int32 [m, 2] of the tracked corners, float -> int
    truncation; prev = the corner (detected in the current image, used as a point of the previous one), curr = where it went."""
    ok = np.asarray(status) == 1
    return np.asarray(corners)[ok].astype(np.int32), np.asarray(tracked)[ok].astype(np.int32)


class VOOracle:
    def __init__(self, cam_T_velo, rect0_T_cam, P_rect0, remove_outlier=100):
        self.L = lib()
        a = np.ascontiguousarray(cam_T_velo, dtype=np.float32)
        b = np.ascontiguousarray(rect0_T_cam, dtype=np.float32)
        c = np.ascontiguousarray(P_rect0, dtype=np.float32)
        self.h = C.c_void_p(self.L.orc_vo_create(_p(a, F), _p(b, F), _p(c, F), remove_outlier))

    def __del__(self):
        try:
            self.L.orc_vo_destroy(self.h)
        except Exception:
            pass

    def reset(self):
        self.L.orc_vo_reset(self.h)

    def process_point_cloud(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        self.L.orc_vo_process_point_cloud(self.h, _p(cloud, F), cloud.shape[0])

    def buckets(self, which_map=0):
        nb = 249 * 75
        bx, by, bd = [np.zeros(nb, dtype=np.float32) for _ in range(3)]
        bc = np.zeros(nb, dtype=np.int32)
        n = self.L.orc_vo_get_buckets(self.h, which_map, _p(bx, F), _p(by, F), _p(bd, F), _p(bc, I), nb)
        return bx[:n], by[:n], bd[:n], bc[:n]

    def points2d(self, which_map=0):
        n = self.L.orc_vo_get_points2d(self.h, which_map, None, 0)
        buf = np.zeros((max(n, 1), 3), dtype=np.float32)
        self.L.orc_vo_get_points2d(self.h, which_map, _p(buf, F), n)
        return buf[:n]

    def query_depth(self, which_map, x, y):
        return float(self.L.orc_vo_query_depth(self.h, which_map, F(x), F(y)))

    def solve(self, prev_uv, curr_uv, init_angles=None, init_t=None):
        pu = np.ascontiguousarray(prev_uv, dtype=np.int32)
        cu = np.ascontiguousarray(curr_uv, dtype=np.int32)
        ang, t = np.zeros(3), np.zeros(3)
        cnt = np.zeros(2, dtype=np.int32)
        trace = np.zeros((128, 8))
        ni, term = I(0), I(0)
        H0, g0, costs = np.zeros((6, 6)), np.zeros(6), np.zeros(2)
        ia_arr = np.ascontiguousarray(init_angles, dtype=np.float64) if init_angles is not None else None
        it_arr = np.ascontiguousarray(init_t, dtype=np.float64) if init_t is not None else None
        ia = _p(ia_arr, D) if ia_arr is not None else None
        it = _p(it_arr, D) if it_arr is not None else None
        self.L.orc_vo_solve(self.h, _p(pu, I), _p(cu, I), pu.shape[0], ia, it, _p(ang, D), _p(t, D), _p(cnt, I), _p(trace, D), 128,
                            C.byref(ni), _p(H0, D), _p(g0, D), C.byref(term), _p(costs, D))
        rows = np.zeros((pu.shape[0], 7))
        self.L.orc_vo_get_match_debug(self.h, _p(rows, D), rows.shape[0])
        return dict(angles=ang, t=t, counter32=int(cnt[0]), counter22=int(cnt[1]), trace=trace[:ni.value], H0=H0, g0=g0,
                    termination=term.value, initial_cost=costs[0], final_cost=costs[1], match_debug=rows)
