"""§8f "next" rows: on-disk formats and the VloamTF frame algebra, host side (pure numpy).

* KITTI raw sweep loader — ``PointCloudUtil::loadPointCloud`` (src/visual_odometry/src/point_cloud_util.cpp:118-146):
  ``float32[4]`` (x, y, z, reflectance) per point; the reference keeps x, y, z only.
* calibration parser — ``PointCloudUtil::loadTransformations`` (point_cloud_util.cpp:5-116): ``R:`` / ``T:`` of
  ``calib_velo_to_cam.txt`` -> cam_T_velo, ``R_rect_00:`` / ``P_rect_00:`` of ``calib_cam_to_cam.txt``.
* VO <-> LO coupling — ``VloamTF::VO2VeloAndBase`` (src/vloam_tf/src/vloam_tf.cpp:59-75) and the LO -> VO prior
  (laser_odometry.cpp:563-567).
* trajectory files — ``VloamTF::{VO,LO,MO}2Cam0StartFrame`` (vloam_tf.cpp:77-153): pose of cam0 at frame k in the cam0
  frame of the start frame, 12 ``%f`` numbers per row (row-major 3x4 [R|t]), as in src/vloam_main/results/*/{VO,LO,MO}*.txt.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------- sweeps
def load_kitti_bin(path):
    """KITTI velodyne .bin -> float32 [n, 4] (x, y, z, 0): the packed float4 layout the C ABI takes."""
    raw = np.fromfile(path, dtype=np.float32)
    raw = raw[: (raw.size // 4) * 4].reshape(-1, 4).copy()
    raw[:, 3] = 0.0  # reflectance is not used (the node converts to pcl::PointXYZ, vloam_main_node.cpp:148)
    return raw


def save_kitti_bin(path, cloud, reflectance=None):
    out = np.zeros((cloud.shape[0], 4), dtype=np.float32)
    out[:, :3] = cloud[:, :3]
    if reflectance is not None:
        out[:, 3] = reflectance
    out.tofile(path)


# ---------------------------------------------------------------------------------------------- images
def load_png_gray(path):
    """8-bit greyscale PNG (KITTI raw image_00 / image_01: colour type 0, bit depth 8, not interlaced) -> uint8 [h, w].
    A dependency-free reader for exactly that format (zlib + the five PNG row filters); anything else raises."""
    import struct
    import zlib
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG file" % path)
    pos, idat, w, h = 8, [], None, None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or ctype != 0 or interlace != 0:
                raise ValueError("%s: only 8-bit greyscale, non-interlaced PNGs are read here (got depth %d, colour type %d, interlace %d)" %
                                 (path, depth, ctype, interlace))
        elif typ == b"IDAT":
            if w is None:
                raise ValueError("%s: IDAT before IHDR" % path)
            idat.append(body)
        elif typ == b"IEND":
            break
    if w is None or not idat:
        raise ValueError("%s: no IHDR / IDAT chunk" % path)
    try:   # a library decoder when one is around (neither is part of this image); the reader below is the dependency-free path
        import io
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(data)).convert("L"), dtype=np.uint8).reshape(h, w)
    except ImportError:
        pass
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), dtype=np.uint8)
    prev = np.zeros(w, dtype=np.uint8)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:]
        if f == 0:
            cur = line
        elif f == 1:     # Sub: a running sum modulo 256
            cur = np.cumsum(line, dtype=np.uint8)
        elif f == 2:     # Up
            cur = line + prev            # uint8 arithmetic wraps modulo 256
        elif f in (3, 4):   # Average / Paeth depend on the decoded pixel to the left: one pass over plain Python ints (no numpy scalars)
            src, up_row = line.tolist(), prev.tolist()
            dst = [0] * w
            left = up_left = 0
            if f == 3:
                for x in range(w):
                    left = (src[x] + ((left + up_row[x]) >> 1)) & 255
                    dst[x] = left
            else:
                for x in range(w):
                    up = up_row[x]
                    pa, pb, pc = abs(up - up_left), abs(left - up_left), abs(left + up - 2 * up_left)
                    pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else up_left)
                    left = (src[x] + pred) & 255
                    dst[x] = left
                    up_left = up
            cur = np.array(dst, dtype=np.uint8)
        else:
            raise ValueError("%s: bad PNG filter type %d" % (path, f))
        out[y] = cur
        prev = out[y]
    return out


def save_png_gray(path, img):
    """uint8 [h, w] -> 8-bit greyscale PNG (filter 0 on every row)."""
    import struct
    import zlib
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


# ---------------------------------------------------------------------------------------------- calibration
def _numbers_after(line, key):
    return [np.float32(v) for v in line[len(key):].split()]


def load_transformations(calib_cam_to_cam_path, calib_velo_to_cam_path):
    """-> (cam_T_velo 4x4, rect0_T_cam 4x4, P_rect0 3x4), float32, exactly the members loadTransformations fills
    (file-based path: rect0_T_cam(3,3) = 1; the ROS path of visual_odometry.cpp:140-144 leaves it 0)."""
    cam_T_velo = np.zeros((4, 4), dtype=np.float32)
    rect0_T_cam = np.zeros((4, 4), dtype=np.float32)
    P_rect0 = np.zeros((3, 4), dtype=np.float32)
    with open(calib_velo_to_cam_path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("R: "):
                cam_T_velo[:3, :3] = np.array(_numbers_after(line, "R: "), dtype=np.float32).reshape(3, 3)
            elif line.startswith("T: "):
                cam_T_velo[:3, 3] = _numbers_after(line, "T: ")
    cam_T_velo[3, 3] = 1
    with open(calib_cam_to_cam_path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("R_rect_00: "):
                rect0_T_cam[:3, :3] = np.array(_numbers_after(line, "R_rect_00: "), dtype=np.float32).reshape(3, 3)
                rect0_T_cam[3, 3] = 1
            elif line.startswith("P_rect_00: "):
                P_rect0[:, :] = np.array(_numbers_after(line, "P_rect_00: "), dtype=np.float32).reshape(3, 4)
    return cam_T_velo, rect0_T_cam, P_rect0


# ---------------------------------------------------------------------------------------------- SE3 helpers (tf2 semantics)
def make_T(q_xyzw, t):
    x, y, z, w = [float(v) for v in q_xyzw]
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n if n > 0 else 0.0
    R = np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                  [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                  [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def inv_T(T):
    R, t = T[:3, :3], T[:3, 3]
    out = np.eye(4)
    out[:3, :3] = R.T
    out[:3, 3] = -R.T @ t
    return out


def T_to_qt(T):
    R = T[:3, :3]
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q), T[:3, 3].copy()


def angle_axis_to_T(aa, t):
    """cam0_curr_T_cam0_last as solveNlsAll builds it (visual_odometry.cpp:425-430): axis = aa / |aa|, angle = |aa|.
    |aa| == 0 gives NaN in the reference (no zero guard); identity is returned here."""
    ang = float(np.linalg.norm(aa))
    if ang == 0.0:
        q = np.array([0.0, 0.0, 0.0, 1.0])
    else:
        q = np.concatenate([np.sin(ang / 2) * np.asarray(aa, dtype=np.float64) / ang, [np.cos(ang / 2)]])
    return make_T(q, t)


class VloamTF:
    """The transform blackboard of src/vloam_tf (static extrinsics + the VO / LO / MO chains), ROS-free."""

    def __init__(self, imu_T_velo, imu_T_cam0, base_T_imu=None):
        self.imu_T_velo = np.asarray(imu_T_velo, dtype=np.float64)
        self.imu_T_cam0 = np.asarray(imu_T_cam0, dtype=np.float64)
        self.base_T_imu = np.eye(4) if base_T_imu is None else np.asarray(base_T_imu, dtype=np.float64)
        self.base_T_cam0 = self.base_T_imu @ self.imu_T_cam0            # vloam_tf.cpp:55
        self.velo_T_cam0 = inv_T(self.imu_T_velo) @ self.imu_T_cam0     # vloam_tf.cpp:56
        self.world_VOT_base_last = np.eye(4)
        self.world_LOT_base_last = np.eye(4)
        self.world_MOT_base_last = np.eye(4)
        self._start = {}

    # VO -> LO prior (vloam_tf.cpp:59-75)
    def VO2VeloAndBase(self, cam0_curr_VOT_cam0_last):
        inv = inv_T(cam0_curr_VOT_cam0_last)
        self.velo_last_VOT_velo_curr = self.velo_T_cam0 @ inv @ inv_T(self.velo_T_cam0)
        self.base_last_VOT_base_curr = self.base_T_cam0 @ inv @ inv_T(self.base_T_cam0)
        if not np.any(np.isnan(self.base_last_VOT_base_curr)):  # "avoid nan at the first couple steps"
            self.world_VOT_base_last = self.world_VOT_base_last @ self.base_last_VOT_base_curr
        return T_to_qt(self.velo_last_VOT_velo_curr)             # -> vloam_set_lo_prior(q, t)

    # LO -> VO prior (laser_odometry.cpp:563-567)
    def LO2CamPrior(self, q_last_curr, t_last_curr):
        self.base_prev_LOT_base_curr = make_T(q_last_curr, t_last_curr)
        self.cam0_curr_LOT_cam0_prev = inv_T(self.base_T_cam0) @ inv_T(self.base_prev_LOT_base_curr) @ self.base_T_cam0
        return self.cam0_curr_LOT_cam0_prev

    # trajectory rows (vloam_tf.cpp:77-153)
    def _to_cam0_start(self, key, world_T_base_last, count):
        cam0_init_T_cam0_last = inv_T(self.base_T_cam0) @ world_T_base_last @ self.base_T_cam0
        if count == 0:
            self._start[key] = cam0_init_T_cam0_last
        return (inv_T(self._start[key]) @ cam0_init_T_cam0_last).astype(np.float32)  # .cast<float>() before printing

    def VO2Cam0StartFrame(self, count):
        return self._to_cam0_start("VO", self.world_VOT_base_last, count)

    def LO2Cam0StartFrame(self, q_w, t_w, count):
        self.world_LOT_base_last = make_T(q_w, t_w)              # laser_odometry.cpp:570-571
        return self._to_cam0_start("LO", self.world_LOT_base_last, count)

    def MO2Cam0StartFrame(self, q_w, t_w, count):
        self.world_MOT_base_last = make_T(q_w, t_w)              # laser_mapping.cpp:728-729
        return self._to_cam0_start("MO", self.world_MOT_base_last, count)


def format_pose_row(T):
    """fprintf(f, "%f %f ... %f\\n", T(0,0) ... T(2,3)) — 12 numbers, row-major 3x4."""
    return " ".join("%f" % float(T[r, c]) for r in range(3) for c in range(4)) + "\n"


def write_trajectory(path, poses):
    with open(path, "w") as f:
        for T in poses:
            f.write(format_pose_row(T))


def read_trajectory(path):
    rows = np.loadtxt(path, ndmin=2)
    out = np.tile(np.eye(4), (rows.shape[0], 1, 1))
    out[:, :3, :] = rows.reshape(-1, 3, 4)
    return out
