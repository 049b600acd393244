"""vloam-cmu-16833_amd — MI355X-native (gfx950) per-scan LiDAR(-visual) odometry hot path of VLOAM.

Thin ctypes binding over ``libvloam_hip.so`` (hand-written HIP kernels + C ABI, see
``include/vloam_hip/c_api.h``) plus a Python mirror of the reference's pull-style class surface
(``ScanRegistration`` / ``LaserOdometry`` / ``LaserMapping`` / ``LidarOdometryMapping``;
reference: src/lidar_odometry_mapping/include/lidar_odometry_mapping/*.h) used by the tests and the bench.

There is no CPU path in this package: importing it requires the built shared library and creating a
handle requires a HIP device.  (The CPU oracle lives in ``oracle/`` and is test infrastructure only.)
"""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # the stage streams of a handle must not share hardware queues — with each other or with the host's own streams (before the HIP runtime loads)
import ctypes as C
import os
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
# VLOAM_HIP_LIB: another build of the SAME library (A/B runs of kernel variants, stamp builds); never a different implementation
LIB_PATH = os.environ.get("VLOAM_HIP_LIB") or os.path.join(_DIR, "libvloam_hip.so")

VLOAM_OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_EMPTY, ERR_NO_DEVICE, ERR_ORDER = 0, -1, -2, -3, -4, -5, -6

K_MAX_RINGS, K_SECTORS = 64, 6
K_MAX_SHARP, K_MAX_LESS_SHARP, K_MAX_FLAT = 768, 7680, 1536
K_MAX_LO_FACTORS = K_MAX_SHARP + K_MAX_FLAT
K_IMG_MAX_CORNERS = 1024   # image_util.cpp:23 maxCorners
K_LM_MAX_TRACE = 104
K_STACK_CAP_CORNER, K_STACK_CAP_SURF = 8192, 24576
K_MAP_FACTOR_CAP = K_STACK_CAP_CORNER + K_STACK_CAP_SURF


class VloamError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("vloam status %d: %s" % (status, msg))
        self.status = status


class Config(C.Structure):
    _fields_ = [("scan_line", C.c_int), ("minimum_range", C.c_double), ("mapping_skip_frame", C.c_int),
                ("mapping_line_resolution", C.c_float), ("mapping_plane_resolution", C.c_float), ("detach_VO_LO", C.c_int),
                ("reset_VO_to_identity", C.c_int), ("remove_VO_outlier", C.c_int), ("with_mapping", C.c_int),
                ("max_points", C.c_int), ("max_frames", C.c_int), ("map_capacity_log2", C.c_int), ("debug", C.c_int),
                ("timing", C.c_int), ("image_width", C.c_int), ("image_height", C.c_int), ("CLAHE", C.c_int)]


class Calib(C.Structure):
    _fields_ = [("cam_T_velo", C.c_float * 16), ("rect0_T_cam", C.c_float * 16), ("P_rect0", C.c_float * 12)]


class LMRecord(C.Structure):
    _fields_ = [("x_in", C.c_double * 7), ("x_out", C.c_double * 7), ("H0", C.c_double * 36), ("g0", C.c_double * 6),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("n_iterations", C.c_double),
                ("termination", C.c_double), ("n_factors", C.c_double), ("n_evals", C.c_double),
                ("cyc", C.c_double * 4), ("trace", (C.c_double * 8) * K_LM_MAX_TRACE)]

    def to_dict(self):
        n = int(self.n_iterations)
        tr = np.ctypeslib.as_array(self.trace).reshape(K_LM_MAX_TRACE, 8)[:n].copy()
        return dict(x_in=np.array(self.x_in), x_out=np.array(self.x_out), H0=np.array(self.H0).reshape(6, 6),
                    g0=np.array(self.g0), initial_cost=self.initial_cost, final_cost=self.final_cost, trace=tr,
                    termination=int(self.termination), n_factors=int(self.n_factors), n_evals=int(self.n_evals),
                    cyc=np.array(self.cyc))


_lib = None


def lib():
    """Load libvloam_hip.so.  Raises (loudly) if it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libvloam_hip.so is not built (%s). Run __graft_entry__.build() or "
                              "`make -C vloam-cmu-16833_amd/csrc`. This package has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.vloam_last_error.restype = C.c_char_p
        L.vloam_version.restype = C.c_char_p
        _lib = L
    return _lib


def default_config(**kw):
    c = Config()
    lib().vloam_default_config(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError("vloam_config has no field %r" % k)
        setattr(c, k, v)
    return c


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class Handle:
    """One sequence on one GPU (``vloam_handle``)."""

    def __init__(self, device=0, n_sessions=1, **cfg):
        """n_sessions > 1: a batched handle — that many independent sequences advanced in lock step by batch_process_scan*;
        select(b) chooses the session the getters read."""
        self.L = lib()
        self.cfg = default_config(**cfg)
        self.h = C.c_void_p()
        self.n_sessions = int(n_sessions)
        self._chk(self.L.vloam_create_batch(C.byref(self.cfg), int(device), self.n_sessions, C.byref(self.h)))

    def _chk(self, st):
        if st != VLOAM_OK:
            raise VloamError(st, self.L.vloam_last_error().decode())

    def close(self):
        if self.h:
            self.L.vloam_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- façade stages (LidarOdometryMapping::reset / scanRegistrationIO / laserOdometryIO / laserMappingIO)
    def reset_frame(self):
        self._chk(self.L.vloam_reset_frame(self.h))

    def scan_registration(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        assert cloud.ndim == 2 and cloud.shape[1] == 4
        self._chk(self.L.vloam_scan_registration(self.h, _fp(cloud), cloud.shape[0]))

    def scan_registration_device(self, dptr, n):
        self._chk(self.L.vloam_scan_registration_device(self.h, C.c_void_p(dptr), int(n)))

    def features(self, which):
        n = C.c_int(0)
        self._chk(self.L.vloam_get_features(self.h, which, None, 0, C.byref(n)))
        buf = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self._chk(self.L.vloam_get_features(self.h, which, _fp(buf), n.value, C.byref(n)))
        return buf[:n.value]

    def set_lo_prior(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        self._chk(self.L.vloam_set_lo_prior(self.h, _fp(q), _fp(t)))

    @staticmethod
    def _cloud_arg(c):
        if c is None:
            return None, None, 0
        a = np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 4)
        if a.shape[0] == 0:
            a = np.zeros((1, 4), np.float32)   # a non-null address for an empty substituted cloud
            return a, _fp(a), 0
        return a, _fp(a), a.shape[0]

    def set_odometry_input(self, laserCloud=None, cornerPointsSharp=None, cornerPointsLessSharp=None, surfPointsFlat=None, surfPointsLessFlat=None):
        """LaserOdometry::input with clouds that are not scan registration's (laser_odometry.cpp:135-146); None keeps the device's."""
        a = [self._cloud_arg(c) for c in (laserCloud, cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat)]
        self._chk(self.L.vloam_set_odometry_input(self.h, a[0][1], a[0][2], a[1][1], a[1][2], a[2][1], a[2][2], a[3][1], a[3][2], a[4][1], a[4][2]))

    def odometry_pose(self):
        q, t = np.zeros(4), np.zeros(3)
        self._chk(self.L.vloam_get_odometry_pose(self.h, _fp(q), _fp(t)))
        return q, t

    def set_mapping_input(self, laserCloudCornerLast=None, laserCloudSurfLast=None, laserCloudFullRes=None, q_wodom_curr=None, t_wodom_curr=None):
        """LaserMapping::input with clouds / an odometry pose that are not LaserOdometry::output's (laser_mapping.cpp:167-196)."""
        a = [self._cloud_arg(c) for c in (laserCloudCornerLast, laserCloudSurfLast, laserCloudFullRes)]
        q = None if q_wodom_curr is None else np.ascontiguousarray(q_wodom_curr, dtype=np.float64)
        t = None if t_wodom_curr is None else np.ascontiguousarray(t_wodom_curr, dtype=np.float64)
        self._chk(self.L.vloam_set_mapping_input(self.h, a[0][1], a[0][2], a[1][1], a[1][2], a[2][1], a[2][2], None if q is None else _fp(q),
                                                 None if t is None else _fp(t)))

    def laser_odometry(self):
        qw, tw, ql, tl = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        self._chk(self.L.vloam_laser_odometry(self.h, _fp(qw), _fp(tw), _fp(ql), _fp(tl)))
        return qw, tw, ql, tl

    def laser_mapping(self):
        q, t = np.zeros(4), np.zeros(3)
        self._chk(self.L.vloam_laser_mapping(self.h, _fp(q), _fp(t)))
        return q, t

    # ---- asynchronous whole-sweep path
    def process_scan(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        self._chk(self.L.vloam_process_scan(self.h, _fp(cloud), cloud.shape[0]))

    def process_scan_host_ptr(self, hptr, n):
        """vloam_process_scan on a raw HOST address (e.g. a pinned torch tensor's data_ptr()): n packed float4."""
        self._chk(self.L.vloam_process_scan(self.h, C.c_void_p(hptr), int(n)))

    def process_scan_device(self, dptr, n):
        self._chk(self.L.vloam_process_scan_device(self.h, C.c_void_p(dptr), int(n)))

    # ---- batched execution (n_sessions sequences per launch chain)
    def batch_process_scan(self, clouds):
        clouds = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        assert len(clouds) == self.n_sessions
        ptrs = (C.c_void_p * self.n_sessions)(*[c.ctypes.data for c in clouds])
        ns = (C.c_int * self.n_sessions)(*[c.shape[0] for c in clouds])
        self._chk(self.L.vloam_batch_process_scan(self.h, ptrs, ns))

    def batch_process_scan_device(self, dptrs, ns):
        assert len(dptrs) == self.n_sessions and len(ns) == self.n_sessions
        ptrs = (C.c_void_p * self.n_sessions)(*[int(p) for p in dptrs])
        nn = (C.c_int * self.n_sessions)(*[int(n) for n in ns])
        self._chk(self.L.vloam_batch_process_scan_device(self.h, ptrs, nn))

    def select(self, session):
        self._chk(self.L.vloam_select_session(self.h, int(session)))
        return self

    def sync(self):
        self._chk(self.L.vloam_sync(self.h))

    def frame_count(self):
        n = C.c_int(0)
        self._chk(self.L.vloam_frame_count(self.h, C.byref(n)))
        return n.value

    def trajectory(self, first=0, count=None):
        if count is None:
            count = self.frame_count() - first
        out = np.zeros((max(count, 1), 14))
        self._chk(self.L.vloam_get_trajectory(self.h, first, count, _fp(out)))
        return out[:count]

    def trajectory_device_ptr(self):
        p, b = C.c_void_p(), C.c_longlong(0)
        self._chk(self.L.vloam_trajectory_device_ptr(self.h, C.byref(p), C.byref(b)))
        return p.value, b.value

    def profile_kernel(self, name, max_launches=4096):
        """HIP-event pair around every launch of the kernel named `name` (its __global__ symbol) on the handle's stream."""
        self._chk(self.L.vloam_profile_kernel(self.h, name.encode(), int(max_launches)))

    def profile_read(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.L.vloam_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_table(self):
        """{kernel symbol: (total ms, launches)} of the launches recorded since profile_kernel("*", n)."""
        self.L.vloam_profile_kernel_name.restype = C.c_char_p
        nk = self.L.vloam_profile_kernel_count()
        ms = np.zeros(nk)
        cnt = np.zeros(nk, dtype=np.int32)
        self._chk(self.L.vloam_profile_read_table(self.h, nk, _fp(ms), _fp(cnt)))
        return {self.L.vloam_profile_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(1, nk) if cnt[k] > 0}

    def stage_ms(self):
        ms = np.zeros(4)
        n = C.c_int(0)
        self._chk(self.L.vloam_get_stage_ms(self.h, _fp(ms), C.byref(n)))
        return ms, n.value

    def counts(self):
        c = np.zeros(16, dtype=np.int64)
        self._chk(self.L.vloam_get_counts(self.h, _fp(c)))
        names = ["N_in", "N2", "n_sharp", "n_lessSharp", "n_flat", "n_lessFlat", "C", "S", "F_corner", "F_plane", "E_o", "n_c",
                 "n_s", "K_m", "E_m", "M"]
        return dict(zip(names, [int(v) for v in c]))

    def health(self):
        """vloam_get_health: cooperative solves that degraded to one workgroup, whether the handle switched to one-workgroup solves, table rebuilds."""
        v = np.zeros(8, dtype=np.int64)
        self._chk(self.L.vloam_get_health(self.h, _fp(v)))
        return dict(fallback_solves=int(v[0]), one_workgroup_solves=bool(v[1]), rebuilds=int(v[2]))

    # ---- VO
    def vo_set_calib(self, cam_T_velo, rect0_T_cam, P_rect0):
        c = Calib()
        c.cam_T_velo[:] = [float(v) for v in np.asarray(cam_T_velo, dtype=np.float32).ravel()]
        c.rect0_T_cam[:] = [float(v) for v in np.asarray(rect0_T_cam, dtype=np.float32).ravel()]
        c.P_rect0[:] = [float(v) for v in np.asarray(P_rect0, dtype=np.float32).ravel()]
        self._chk(self.L.vloam_vo_set_calib(self.h, C.byref(c)))

    def vo_process_point_cloud(self, cloud):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        self._chk(self.L.vloam_vo_process_point_cloud(self.h, _fp(cloud), cloud.shape[0]))

    def vo_solve(self, prev_uv, curr_uv, angle_axis, t):
        pu = np.ascontiguousarray(prev_uv, dtype=np.int32)
        cu = np.ascontiguousarray(curr_uv, dtype=np.int32)
        aa = np.array(angle_axis, dtype=np.float64)
        tt = np.array(t, dtype=np.float64)
        cnt = np.zeros(2, dtype=np.int32)
        self._chk(self.L.vloam_vo_solve(self.h, _fp(pu), _fp(cu), pu.shape[0], _fp(aa), _fp(tt), _fp(cnt)))
        return aa, tt, int(cnt[0]), int(cnt[1])

    # ---- coupled VLOAM frame loop (configs[3])
    def set_extrinsics(self, base_T_cam0, velo_T_cam0):
        a = np.ascontiguousarray(base_T_cam0, dtype=np.float64).reshape(16)
        b = np.ascontiguousarray(velo_T_cam0, dtype=np.float64).reshape(16)
        self._chk(self.L.vloam_set_extrinsics(self.h, _fp(a), _fp(b)))

    def _matches(self, prev_uv, curr_uv):
        if prev_uv is None or curr_uv is None:
            return None, None, None, None, 0
        pu = np.ascontiguousarray(prev_uv, dtype=np.int32)
        cu = np.ascontiguousarray(curr_uv, dtype=np.int32)
        return pu, cu, _fp(pu), _fp(cu), pu.shape[0]

    def process_frame(self, cloud, prev_uv=None, curr_uv=None):
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        pu, cu, pp, cp, n = self._matches(prev_uv, curr_uv)
        self._chk(self.L.vloam_process_frame(self.h, _fp(cloud), cloud.shape[0], pp, cp, n))

    def process_frame_device(self, dptr, n_pts, prev_uv=None, curr_uv=None):
        pu, cu, pp, cp, n = self._matches(prev_uv, curr_uv)
        self._chk(self.L.vloam_process_frame_device(self.h, C.c_void_p(dptr), int(n_pts), pp, cp, n))

    def batch_process_frame_device(self, dptrs, ns, matches):
        """One coupled VLOAM frame for every session: matches[b] = (prev_uv, curr_uv) int32 [m, 2] (or (None, None))."""
        B = self.n_sessions
        assert len(dptrs) == B and len(ns) == B and len(matches) == B
        keep = [self._matches(m[0], m[1]) for m in matches]
        ptrs = (C.c_void_p * B)(*[int(p) for p in dptrs])
        nn = (C.c_int * B)(*[int(n) for n in ns])
        IP = C.POINTER(C.c_int)
        pp = (IP * B)(*[C.cast(k[2], IP) for k in keep])
        cp = (IP * B)(*[C.cast(k[3], IP) for k in keep])
        nm = (C.c_int * B)(*[int(k[4]) for k in keep])
        self._chk(self.L.vloam_batch_process_frame_device(self.h, ptrs, nn, pp, cp, nm))

    def batch_process_frame(self, clouds, matches):
        """Host-memory variant of batch_process_frame_device: clouds[b] float32 [n, 4]."""
        B = self.n_sessions
        assert len(clouds) == B and len(matches) == B
        cl = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        keep = [self._matches(m[0], m[1]) for m in matches]
        FP, IP = C.POINTER(C.c_float), C.POINTER(C.c_int)
        ptrs = (FP * B)(*[C.cast(_fp(c), FP) for c in cl])
        nn = (C.c_int * B)(*[int(c.shape[0]) for c in cl])
        pp = (IP * B)(*[C.cast(k[2], IP) for k in keep])
        cp = (IP * B)(*[C.cast(k[3], IP) for k in keep])
        nm = (C.c_int * B)(*[int(k[4]) for k in keep])
        self._chk(self.L.vloam_batch_process_frame(self.h, ptrs, nn, pp, cp, nm))

    def vo_trajectory(self, first=0, count=None):
        if count is None:
            count = self.frame_count() - first
        out = np.zeros((max(count, 1), 7))
        self._chk(self.L.vloam_get_vo_trajectory(self.h, first, count, _fp(out)))
        return out[:count]

    def vo_result(self):
        aa, t, cnt, pq, pt = np.zeros(3), np.zeros(3), np.zeros(2, dtype=np.int32), np.zeros(4), np.zeros(3)
        self._chk(self.L.vloam_get_vo_result(self.h, _fp(aa), _fp(t), _fp(cnt), _fp(pq), _fp(pt)))
        return dict(angles=aa, t=t, counter32=int(cnt[0]), counter22=int(cnt[1]), prior_q=pq, prior_t=pt)

    # ---- image front-end (optical-flow configuration; needs image_width / image_height in the config)
    def vo_process_image(self, gray):
        """VisualOdometry::processImage (visual_odometry.cpp:91-132, optical_flow_match = true): uint8 [h, w]."""
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        self._chk(self.L.vloam_vo_process_image(self.h, _fp(g), g.shape[1], g.shape[0], g.shape[1]))

    def vo_process_image_device(self, dptr, width, height, stride=None):
        self._chk(self.L.vloam_vo_process_image_device(self.h, C.c_void_p(dptr), int(width), int(height), int(stride or width)))

    def vo_keypoints(self):
        n = C.c_int(0)
        out = np.zeros((K_IMG_MAX_CORNERS, 2), dtype=np.float32)
        self._chk(self.L.vloam_vo_get_keypoints(self.h, _fp(out), K_IMG_MAX_CORNERS, C.byref(n)))
        return out[:n.value]

    def vo_flow(self):
        """(corner in the previous image, tracked position in the new image, status) per corner of the last image."""
        n = C.c_int(0)
        a = np.zeros((K_IMG_MAX_CORNERS, 2), dtype=np.float32)
        b = np.zeros((K_IMG_MAX_CORNERS, 2), dtype=np.float32)
        st = np.zeros(K_IMG_MAX_CORNERS, dtype=np.uint8)
        self._chk(self.L.vloam_vo_get_flow(self.h, _fp(a), _fp(b), _fp(st), K_IMG_MAX_CORNERS, C.byref(n)))
        return a[:n.value], b[:n.value], st[:n.value]

    def vo_flow_matches(self):
        n = C.c_int(0)
        a = np.zeros((K_IMG_MAX_CORNERS, 2), dtype=np.int32)
        b = np.zeros((K_IMG_MAX_CORNERS, 2), dtype=np.int32)
        self._chk(self.L.vloam_vo_get_flow_matches(self.h, _fp(a), _fp(b), K_IMG_MAX_CORNERS, C.byref(n)))
        return a[:n.value], b[:n.value]

    def vo_set_orb_pattern(self, pattern):
        """ORB + brute-force configuration (optical_flow_match = false): OpenCV's bit_pattern_31_ as int8 [256, 4]; None = optical flow."""
        if pattern is None:
            self._chk(self.L.vloam_vo_set_orb_pattern(self.h, None))
            return
        p = np.ascontiguousarray(pattern, dtype=np.int8).reshape(256, 4)
        self._chk(self.L.vloam_vo_set_orb_pattern(self.h, _fp(p)))

    def vo_descriptors(self):
        """(keypoints [n, 2] f32 after ORB's border filter, descriptors [n, 32] u8) of the latest image."""
        n = C.c_int(0)
        self._chk(self.L.vloam_vo_get_descriptors(self.h, None, None, 0, C.byref(n)))
        xy = np.zeros((max(n.value, 1), 2), np.float32)
        d = np.zeros((max(n.value, 1), 32), np.uint8)
        self._chk(self.L.vloam_vo_get_descriptors(self.h, _fp(xy), _fp(d), n.value, C.byref(n)))
        return xy[:n.value], d[:n.value]

    def vo_match_descriptors(self, desc_prev, desc_curr, knn=True):
        """ImageUtil::matchDescriptors (BF, NORM_HAMMING): (queryIdx, trainIdx) int32 arrays in query order."""
        a = np.ascontiguousarray(desc_prev, dtype=np.uint8)
        b = np.ascontiguousarray(desc_curr, dtype=np.uint8)
        q = np.zeros(max(a.shape[0], 1), dtype=np.int32)
        t = np.zeros(max(a.shape[0], 1), dtype=np.int32)
        n = C.c_int(0)
        self._chk(self.L.vloam_vo_match_descriptors(self.h, _fp(a), a.shape[0], _fp(b), b.shape[0], a.shape[1] if a.ndim == 2 else 0, int(knn), _fp(q), _fp(t),
                                                    q.shape[0], C.byref(n)))
        return q[:n.value], t[:n.value]

    def process_frame_image(self, cloud, gray):
        """One frame of the coupled loop from raw inputs: sweep + grey image (matches come from the image front-end)."""
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        self._chk(self.L.vloam_process_frame_image(self.h, _fp(cloud), cloud.shape[0], _fp(g), g.shape[1], g.shape[0], g.shape[1]))

    def batch_process_frame_image(self, clouds, grays):
        """One coupled frame of every session of a batched handle from raw inputs: clouds[b] float32 [n, 4], grays[b] uint8 [h, w] (one size)."""
        B = self.n_sessions
        assert len(clouds) == B and len(grays) == B
        cl = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        gs = [np.ascontiguousarray(g, dtype=np.uint8) for g in grays]
        assert all(g.shape == gs[0].shape for g in gs), "the images of one call share their size"
        FP, BP = C.POINTER(C.c_float), C.POINTER(C.c_ubyte)
        ptrs = (FP * B)(*[C.cast(_fp(c), FP) for c in cl])
        nn = (C.c_int * B)(*[int(c.shape[0]) for c in cl])
        gp = (BP * B)(*[C.cast(_fp(g), BP) for g in gs])
        self._chk(self.L.vloam_batch_process_frame_image(self.h, ptrs, nn, gp, gs[0].shape[1], gs[0].shape[0], gs[0].shape[1]))

    def batch_process_frame_image_device(self, dptrs, n_pts, gptrs, width, height, stride=None):
        B = self.n_sessions
        assert len(dptrs) == B and len(n_pts) == B and len(gptrs) == B
        ptrs = (C.c_void_p * B)(*[C.c_void_p(int(p)) for p in dptrs])
        nn = (C.c_int * B)(*[int(v) for v in n_pts])
        gp = (C.c_void_p * B)(*[C.c_void_p(int(p)) for p in gptrs])
        self._chk(self.L.vloam_batch_process_frame_image_device(self.h, ptrs, nn, gp, int(width), int(height), int(stride or width)))

    def process_frame_image_device(self, dptr, n_pts, gptr, width, height, stride=None):
        self._chk(self.L.vloam_process_frame_image_device(self.h, C.c_void_p(dptr), int(n_pts), C.c_void_p(gptr), int(width), int(height),
                                                          int(stride or width)))

    def img_debug(self, width, height):
        """eig map f32 [h, w] and the pyramid of the last image: [(u8 [h_l, w_l], int16 [h_l, w_l, 2])]."""
        eig = self.debug_raw(4, 0, np.float32).reshape(height, width)
        lv = []
        w, h = width, height
        for l in range(3):
            try:
                im = self.debug_raw(4, 1 + l, np.uint8)
            except VloamError:
                break
            lv.append((im.reshape(h, w).copy(), self.debug_raw(4, 4 + l, np.int16).reshape(h, w, 2).copy()))
            w, h = (w + 1) // 2, (h + 1) // 2
        return eig, lv

    # ---- parity hooks
    def debug_raw(self, stage, item, dtype, max_bytes=1 << 26):
        n = C.c_longlong(0)
        self._chk(self.L.vloam_debug_get(self.h, stage, item, None, C.c_longlong(0), C.byref(n)))
        nb = min(n.value, max_bytes)
        buf = np.zeros(max(nb, 8), dtype=np.uint8)
        self._chk(self.L.vloam_debug_get(self.h, stage, item, _fp(buf), C.c_longlong(nb), C.byref(n)))
        return buf[:nb].view(dtype)

    def debug_lm_record(self, stage, item):
        raw = self.debug_raw(stage, item, np.uint8)
        rec = LMRecord.from_buffer_copy(raw.tobytes())
        return rec.to_dict()

    def sr_debug(self):
        sc = self.debug_raw(0, 9, np.float32)
        return dict(curvature=self.debug_raw(0, 0, np.float32), sort=self.debug_raw(0, 1, np.int32),
                    picked=self.debug_raw(0, 2, np.int32), label=self.debug_raw(0, 3, np.int32),
                    scanStartInd=self.debug_raw(0, 4, np.int32), scanEndInd=self.debug_raw(0, 5, np.int32),
                    sharpInd=self.debug_raw(0, 6, np.int32), lessSharpInd=self.debug_raw(0, 7, np.int32),
                    flatInd=self.debug_raw(0, 8, np.int32), startOri=sc[0], endOri=sc[1], istar=int(sc[2]),
                    n_after_s1=int(sc[3]), N2=int(sc[4]))

    def lo_debug(self, outer):
        c = self.debug_raw(1, outer * 16 + 0, np.int32).reshape(-1, 4)
        p = self.debug_raw(1, outer * 16 + 1, np.int32).reshape(-1, 4)
        rec = self.debug_lm_record(1, outer * 16 + 2)
        resid = self.debug_raw(1, outer * 16 + 3, np.float64).reshape(3, K_MAX_LO_FACTORS)
        return dict(corner=c[c[:, 0] >= 0][:, :3], plane=p[p[:, 0] >= 0], corner_slots=np.nonzero(c[:, 0] >= 0)[0],
                    plane_slots=np.nonzero(p[:, 0] >= 0)[0] + K_MAX_SHARP, rec=rec, resid=resid)


    def map_debug(self, outer):
        cap = K_MAP_FACTOR_CAP
        types = self.debug_raw(2, outer * 16 + 0, np.int32)
        A = self.debug_raw(2, outer * 16 + 1, np.float64).reshape(3, cap).T
        B = self.debug_raw(2, outer * 16 + 2, np.float64).reshape(3, cap).T
        resid = self.debug_raw(2, outer * 16 + 4, np.float64).reshape(3, cap)
        rec = self.debug_lm_record(2, outer * 16 + 3)
        cs = np.nonzero(types[:K_STACK_CAP_CORNER] == 1)[0]
        ss = np.nonzero(types[K_STACK_CAP_CORNER:] == 3)[0]
        return dict(corner_idx=cs, corner_ab=np.hstack([A[cs], B[cs]]), surf_idx=ss,
                    surf_plane=np.hstack([A[ss + K_STACK_CAP_CORNER], B[ss + K_STACK_CAP_CORNER, :1]]), rec=rec, resid=resid,
                    corner_slots=cs, surf_slots=ss + K_STACK_CAP_CORNER)

    def vo_debug(self, n_match):
        nb = 249 * 75
        cur = [self.debug_raw(3, k, np.float32 if k < 3 else np.int32)[:nb] for k in range(4)]
        prev = [self.debug_raw(3, 4 + k, np.float32 if k < 3 else np.int32)[:nb] for k in range(4)]
        rows = self.debug_raw(3, 8, np.float64).reshape(-1, 7)[:n_match]
        return dict(cur=cur, prev=prev, match_rows=rows, rec=self.debug_lm_record(3, 9))

    def map_state(self):
        raw = self.debug_raw(2, 64, np.uint8)
        d = raw[:21 * 8].view(np.float64)
        i = raw[21 * 8:21 * 8 + 13 * 4].view(np.int32)
        return dict(parameters=d[:7].copy(), q_wmap_wodom=d[7:11].copy(), t_wmap_wodom=d[11:14].copy(), cen=i[0:3].copy(),
                    centerCube=i[3:6].copy(), n_corner_stack=int(i[6]), n_surf_stack=int(i[7]), do_optimize=int(i[8]),
                    n_map_corner=int(i[9]), n_map_surf=int(i[10]), deferred=int(i[11]))

    def get_map(self):
        """/laser_cloud_map (laser_mapping.cpp:778-793) as float32 [n, 4]: per cube the corner cloud then the surf cloud."""
        n = C.c_longlong(0)
        self._chk(self.L.vloam_get_map(self.h, None, C.c_longlong(0), C.byref(n)))
        buf = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self._chk(self.L.vloam_get_map(self.h, _fp(buf), C.c_longlong(n.value), C.byref(n)))
        return buf[:n.value]

    def map_health(self):
        v = self.debug_raw(2, 69, np.int32)
        return dict(keys=(int(v[0]), int(v[4])), purged=(int(v[1]), int(v[5])), block_keys=(int(v[2]), int(v[6])), rebuilds=int(v[8]),
                    max_candidates=int(v[9]), deferred=(int(v[10]), int(v[11])))

    def map_force_rebuild(self):
        n = C.c_longlong(0)
        self._chk(self.L.vloam_debug_get(self.h, 2, 70, None, C.c_longlong(0), C.byref(n)))

    def map_dump(self, kind):
        """Live voxels of the corner (0) / surf (1) map as (count int32[n], xyzi float32[n, 4])."""
        rows = self.debug_raw(2, 67 + kind, np.uint32, max_bytes=1 << 30).reshape(-1, 7)
        return rows[:, 2].astype(np.int32), rows[:, 3:7].copy().view(np.float32)


# ------------------------------------------------------------------------------------------------
# Python mirror of the reference's class surface (same method names / call order / error behaviour).
class ScanRegistration:
    """vloam::ScanRegistration (scan_registration.h:71-77): init / reset / input / output."""

    def __init__(self, handle):
        self.hd = handle

    def init(self):
        pass  # parameters were bound at vloam_create (the reference reads them from the ROS parameter server here)

    def reset(self):
        self.hd.reset_frame()

    def input(self, laserCloudIn):
        self.hd.scan_registration(laserCloudIn)

    def output(self):
        return tuple(self.hd.features(k) for k in range(5))


class LaserOdometry:
    """vloam::LaserOdometry (laser_odometry.h:70-84): init / input / solveLO / output."""

    def __init__(self, handle):
        self.hd = handle
        self._pose = None

    def init(self):
        pass

    def input(self, *clouds):
        pass  # clouds stay resident in HBM; the reference deep-copies them here (laser_odometry.cpp:141-145)

    def solveLO(self):
        self._pose = self.hd.laser_odometry()

    # the public pose members of the reference class (laser_odometry.h:86-98), x y z w
    q_w_curr = property(lambda self: self._pose[0])
    t_w_curr = property(lambda self: self._pose[1])
    q_last_curr = property(lambda self: self._pose[2])
    t_last_curr = property(lambda self: self._pose[3])

    def output(self):
        qw, tw, _, _ = self._pose
        skip = (self.hd.frame_count() + (0 if self.hd.cfg.with_mapping else 0)) % self.hd.cfg.mapping_skip_frame != 0
        return qw, tw, self.hd.features(5), self.hd.features(6), self.hd.features(0), skip


class LaserMapping:
    """vloam::LaserMapping (laser_mapping.h:85-94): init / reset / input / solveMapping."""

    def __init__(self, handle):
        self.hd = handle
        self.pose = None

    def init(self):
        pass

    def reset(self):
        pass

    def input(self, *args):
        pass

    def solveMapping(self):
        self.pose = self.hd.laser_mapping()

    q_w_curr = property(lambda self: self.pose[0])   # laser_mapping.h:129-130
    t_w_curr = property(lambda self: self.pose[1])


class LidarOdometryMapping:
    """vloam::LidarOdometryMapping façade (lidar_odometry_mapping.cpp:65-154)."""

    def __init__(self, device=0, **cfg):
        self.hd = Handle(device, **cfg)
        self.scan_registration = ScanRegistration(self.hd)
        self.laser_odometry = LaserOdometry(self.hd)
        self.laser_mapping = LaserMapping(self.hd)

    def reset(self):
        self.scan_registration.reset()
        self.laser_mapping.reset()

    def scanRegistrationIO(self, laserCloudIn):
        self.scan_registration.input(laserCloudIn)

    def laserOdometryIO(self):
        self.laser_odometry.solveLO()
        return self.laser_odometry._pose

    def laserMappingIO(self):
        self.laser_mapping.solveMapping()
        return self.laser_mapping.pose
