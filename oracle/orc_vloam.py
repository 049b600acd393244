"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/orc_math.h).

The coupled per-frame VLOAM loop of the reference (configs[3]) on the CPU oracle:

    MAIN/src/vloam_main_node.cpp:125-180   callback(): VO->reset, LOAM->reset, processPointCloud, solveNlsAll (count > 0),
                                           vloam_tf->VO2VeloAndBase, scanRegistrationIO, laserOdometryIO, laserMappingIO
    TF/src/vloam_tf.cpp:55-75              static extrinsics, VO2VeloAndBase (VO -> LiDAR odometry prior)
    LOM/src/laser_odometry.cpp:223-236     combined mode: para_q / para_t overwritten by velo_last_VOT_velo_curr in BOTH outer rounds
    LOM/src/laser_odometry.cpp:563-567     LaserOdometry::publish: cam0_curr_LOT_cam0_prev (LiDAR odometry -> VO prior)
    VO/src/visual_odometry.cpp:258-281     solveNlsAll initial guess from cam0_curr_LOT_cam0_prev (unless reset_VO_to_identity)
    VO/src/visual_odometry.cpp:425-430     angle-axis -> tf2 quaternion

tf2 semantics restated here (tf2/LinearMath, double precision): a Transform is (3x3 basis, origin); setRotation(q) builds the
basis from the quaternion (Matrix3x3::setRotation, s = 2 / |q|^2), getRotation() converts the basis back
(Matrix3x3::getRotation, trace / largest-diagonal branches), Quaternion::setRotation(axis, angle) = (axis * sin(angle/2) /
|axis|, cos(angle/2)), getAngle() = 2 acos(w), getAxis() = xyz / sqrt(1 - w^2) or (1, 0, 0) when 1 - w^2 < 10 eps.

The image front-end is out of scope: matched pixel pairs (prev_uv, curr_uv) are inputs.
"""
import numpy as np

import orc

EPS = np.finfo(np.float64).eps


# ---------------------------------------------------------------------------------------------- tf2 restated
def basis_from_quat(q):
    x, y, z, w = [float(v) for v in q]
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1.0 - (xx + yy)]])


def quat_from_basis(m):
    """Matrix3x3::getRotation."""
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if tr > 0.0:
        s = np.sqrt(tr + 1.0)
        q[3] = s * 0.5
        s = 0.5 / s
        q[0] = (m[2, 1] - m[1, 2]) * s
        q[1] = (m[0, 2] - m[2, 0]) * s
        q[2] = (m[1, 0] - m[0, 1]) * s
    else:
        i = (2 if m[1, 1] < m[2, 2] else 1) if m[0, 0] < m[1, 1] else (2 if m[0, 0] < m[2, 2] else 0)
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = s * 0.5
        s = 0.5 / s
        q[3] = (m[k, j] - m[j, k]) * s
        q[j] = (m[j, i] + m[i, j]) * s
        q[k] = (m[k, i] + m[i, k]) * s
    return q


class TF:
    """tf2::Transform."""

    def __init__(self, basis=None, origin=None):
        self.m = np.eye(3) if basis is None else np.array(basis, dtype=np.float64)
        self.o = np.zeros(3) if origin is None else np.array(origin, dtype=np.float64)

    @staticmethod
    def from_qt(q, t):
        return TF(basis_from_quat(q), t)

    @staticmethod
    def from_matrix4(T):
        T = np.asarray(T, dtype=np.float64)
        return TF(T[:3, :3], T[:3, 3])

    def __mul__(self, other):   # Transform::operator*: (m1 m2, m1 o2 + o1)
        return TF(self.m @ other.m, self.m @ other.o + self.o)

    def inverse(self):          # (m^T, m^T * -o)
        inv = self.m.T
        return TF(inv, inv @ (-self.o))

    def rotation(self):
        return quat_from_basis(self.m)

    def has_nan(self):
        return bool(np.any(np.isnan(self.o)) or np.any(np.isnan(self.rotation())))


def quat_axis_angle(axis, angle):
    """Quaternion::setRotation(axis, angle)."""
    axis = np.asarray(axis, dtype=np.float64)
    d = np.sqrt(float(axis @ axis))
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.sin(angle * 0.5) / d
        return np.array([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(angle * 0.5)])


def quat_get_angle(q):
    return 2.0 * np.arccos(min(1.0, max(-1.0, float(q[3]))))   # tf2Acos clamps its argument


def quat_get_axis(q):
    s_squared = 1.0 - float(q[3]) * float(q[3])
    if s_squared < 10.0 * EPS:
        return np.array([1.0, 0.0, 0.0])
    s = np.sqrt(s_squared)
    return np.array([q[0] / s, q[1] / s, q[2] / s])


# ---------------------------------------------------------------------------------------------- the frame loop
class VloamOracle:
    """One VLOAM session: VisualOdometry + LidarOdometryMapping + VloamTF, driven exactly like callback()."""

    def __init__(self, cam_T_velo, rect0_T_cam, P_rect0, base_T_cam0, velo_T_cam0, detach_VO_LO=False, reset_VO_to_identity=False,
                 remove_VO_outlier=100, scan_line=64, with_mapping=True, mapping_skip_frame=1):
        self.vo = orc.VOOracle(cam_T_velo, rect0_T_cam, P_rect0, remove_outlier=remove_VO_outlier)
        self.lidar = orc.Oracle(scan_line=scan_line, detach_vo_lo=detach_VO_LO, with_mapping=with_mapping, mapping_skip_frame=mapping_skip_frame)
        self.base_T_cam0 = TF.from_matrix4(base_T_cam0)   # vloam_tf.cpp:55
        self.velo_T_cam0 = TF.from_matrix4(velo_T_cam0)   # vloam_tf.cpp:56
        self.reset_VO_to_identity = reset_VO_to_identity
        self.count = 0                                     # vloam_main_node.cpp:113
        self.cam0_curr_T_cam0_last = TF()                  # visual_odometry.cpp:73-74
        self.world_VOT_base_last = TF()                    # vloam_tf.cpp:10-11
        self.cam0_curr_LOT_cam0_prev = TF()                # first read at count == 1, after LaserOdometry::publish of count 0 set it
        self.velo_last_VOT_velo_curr = TF()
        self.vo_result = None

    def process(self, cloud, prev_uv=None, curr_uv=None):
        """One callback().  prev_uv / curr_uv: integer pixel pairs (previous frame -> this frame), ignored for the first frame."""
        self.vo.reset()                                    # VO->reset(); LOAM->reset() is inside Oracle.process
        self.vo.process_point_cloud(cloud)                 # Section 3
        self.vo_result = None
        if self.count > 0:                                 # Section 4: VO->solveNlsAll()
            if self.reset_VO_to_identity:
                a0, t0 = np.zeros(3), np.zeros(3)
            else:                                          # visual_odometry.cpp:270-281
                q = self.cam0_curr_LOT_cam0_prev.rotation()
                a0 = quat_get_axis(q) * quat_get_angle(q)
                t0 = self.cam0_curr_LOT_cam0_prev.o.copy()
            r = self.vo.solve(prev_uv, curr_uv, a0, t0)
            r["init_angles"], r["init_t"] = a0, t0
            self.vo_result = r
            angle = float(np.sqrt(r["angles"][0] ** 2 + r["angles"][1] ** 2 + r["angles"][2] ** 2))   # visual_odometry.cpp:427
            with np.errstate(invalid="ignore", divide="ignore"):
                q = quat_axis_angle(r["angles"] / angle, angle)                                       # NaN when angle == 0, like the reference
            self.cam0_curr_T_cam0_last = TF(basis_from_quat(q), r["t"])
        self.VO2VeloAndBase(self.cam0_curr_T_cam0_last)
        self.lidar.set_vo_prior(self.velo_last_VOT_velo_curr.rotation(), self.velo_last_VOT_velo_curr.o)  # laser_odometry.cpp:225-232
        rc = self.lidar.process(cloud)                     # Section 5
        # LaserOdometry::publish, laser_odometry.cpp:563-567
        _, _, q_last_curr, t_last_curr = self.lidar.lo_pose()
        base_prev_LOT_base_curr = TF.from_qt(q_last_curr, t_last_curr)
        self.cam0_curr_LOT_cam0_prev = self.base_T_cam0.inverse() * base_prev_LOT_base_curr.inverse() * self.base_T_cam0
        self.count += 1
        return rc

    def process_image(self, cloud, gray):
        """One callback() from raw inputs: VisualOdometry::processImage (visual_odometry.cpp:91-132) — corners of the new image, then either
        (optical_flow_match = true) tracked from the previous image into the new one, or (self.orb_pattern set: optical_flow_match = false)
        described by ORB and matched by brute force against the previous image's — then the match loop's integer pairs."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        if getattr(self, "CLAHE", False):                                  # visual_odometry.cpp:97-98
            gray = orc.clahe(gray)
        corners = orc.good_features(gray)                                  # image_util.cpp:13-36
        prev_uv = curr_uv = None
        self.flow = None
        pattern = getattr(self, "orb_pattern", None)
        if pattern is not None:
            # optical_flow_match = false (the launch default, vloam_main.launch:10): descKeypoints edits the keypoint vector (border filter),
            # matchDescriptors(previous, this), and the match loop reads the FILTERED keypoints through the DMatch indices (visual_odometry.cpp:106-116,296-303)
            kept, desc = orc.orb_descriptors(gray, corners, pattern)
            self.orb = (corners[kept], desc)
            if self.count > 0:
                pk, pd = self.prev_orb
                q, t = orc.bf_match_hamming(pd, desc, knn=True)
                prev_uv, curr_uv = pk[q].astype(np.int32), corners[kept][t].astype(np.int32)
            self.prev_orb = self.orb
            self.prev_image = gray
            self.keypoints = corners
            return self.process(cloud, prev_uv, curr_uv)
        if self.count > 0:
            tracked, status = orc.pyr_lk(self.prev_image, gray, corners)   # image_util.cpp:351-372 (prev image, new image, NEW corners)
            prev_uv, curr_uv = orc.flow_matches(corners, tracked, status)   # visual_odometry.cpp:296-308
            self.flow = (corners, tracked, status)
        self.prev_image = gray
        self.keypoints = corners
        return self.process(cloud, prev_uv, curr_uv)

    def VO2VeloAndBase(self, cam0_curr_VOT_cam0_last):     # vloam_tf.cpp:59-75
        inv = cam0_curr_VOT_cam0_last.inverse()
        self.velo_last_VOT_velo_curr = self.velo_T_cam0 * inv * self.velo_T_cam0.inverse()
        self.base_last_VOT_base_curr = self.base_T_cam0 * inv * self.base_T_cam0.inverse()
        if not self.base_last_VOT_base_curr.has_nan():     # "avoid nan at the first couple steps"
            self.world_VOT_base_last = self.world_VOT_base_last * self.base_last_VOT_base_curr

    # ---- what the product's trajectory log holds
    def vo_world_pose(self):
        return self.world_VOT_base_last.rotation(), self.world_VOT_base_last.o.copy()

    def lo_prior(self):
        return self.velo_last_VOT_velo_curr.rotation(), self.velo_last_VOT_velo_curr.o.copy()
