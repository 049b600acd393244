"""Seeded synthetic HDL-64E / VLP-16 sweep generator.

Test / bench input only: analytic ray casting of a ground plane, axis-aligned boxes and vertical
cylinders placed along a smooth SE3 trajectory.  Emits ring-major ``[n_rings * n_azimuth, 4]``
float32 clouds (x, y, z, 0) in the sensor frame; misses are NaN (exercises the reference's
``removeNaNFromPointCloud``, scan_registration.cpp:157), returns closer than ``minimum_range`` are
kept (exercises ``removeClosedPointCloud``, scan_registration.cpp:100-129).

Beam table: ring r -> elevation 1.9 - r/3 deg (r <= 31), -8.93 - (r-32)/2 deg (r >= 32), i.e. the
HDL-64E table shifted 0.1 deg down so the reference's ``int((2-angle)*3+0.5)`` /
``32+int((-8.83-angle)*2+0.5)`` mapping (scan_registration.cpp:213-226) lands mid-bin.
"""
import numpy as np

GROUND_Z = -1.73
MAX_RANGE = 80.0


def beam_elevations_deg(n_rings):
    r = np.arange(n_rings, dtype=np.float64)
    if n_rings == 64:
        return np.where(r <= 31, 1.9 - r / 3.0, -8.93 - (r - 32) / 2.0)
    if n_rings == 16:  # VLP-16: scanID = int((angle + 15) / 2 + 0.5)  (scan_registration.cpp:195-203)
        return -15.0 + 2.0 * r
    if n_rings == 32:  # HDL-32: scanID = int((angle + 92/3) * 3/4)      (scan_registration.cpp:204-212)
        return -92.0 / 3.0 + (r + 0.5) * 4.0 / 3.0
    raise ValueError("n_rings must be 16, 32 or 64")


def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def rot_to_quat_xyzw(R):
    """Rotation matrix -> unit quaternion (x, y, z, w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
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
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class SynthSequence:
    def __init__(self, n_rings=64, n_azimuth=2048, n_sweeps=200, seed_scene=1234, seed_traj=42, seed_noise=5678,
                 noise_sigma=0.02, speed=10.0, dt=0.1, pitch_amp=0.01, roll_amp=0.01, heave_amp=0.05):
        self.n_rings, self.n_azimuth, self.n_sweeps = n_rings, n_azimuth, n_sweeps
        self.seed_noise, self.noise_sigma = seed_noise, noise_sigma
        el = np.deg2rad(beam_elevations_deg(n_rings))
        # clockwise so ori = -atan2(y, x) increases.  A fixed per-column jitter keeps the firing azimuths off a perfect grid: on a
        # perfect grid whole columns sit EXACTLY on the reference's +-pi/2 unwrap thresholds (scan_registration.cpp:237-261),
        # where a 1-ulp difference between two atan2f implementations flips relTime — and with it int(intensity) — by a turn.
        jit = np.random.default_rng(seed_scene + 7).uniform(-0.35, 0.35, n_azimuth)
        az = -2.0 * np.pi * (np.arange(n_azimuth) + jit) / n_azimuth
        ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
        d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (n_rings, n_azimuth))], -1)
        self.dirs = d.reshape(-1, 3)  # ring-major
        # ---- trajectory: smooth speed / yaw-rate / small roll, pitch, heave
        rt = np.random.default_rng(seed_traj)
        ph = rt.uniform(0, 2 * np.pi, 6)
        k = np.arange(n_sweeps)
        t = k * dt
        v = speed * (1.0 + 0.1 * np.sin(2 * np.pi * t / 7.3 + ph[0]))
        yaw_rate = 0.2 * np.sin(2 * np.pi * t / 11.0 + ph[1])
        yaw = np.concatenate([[0.0], np.cumsum(yaw_rate[:-1] * dt)])
        x = np.concatenate([[0.0], np.cumsum(v[:-1] * np.cos(yaw[:-1]) * dt)])
        y = np.concatenate([[0.0], np.cumsum(v[:-1] * np.sin(yaw[:-1]) * dt)])
        z = heave_amp * np.sin(2 * np.pi * t / 5.0 + ph[2])        # (defaults: the benchmark sequence; larger amplitudes: steep / banked test drives)
        pitch = pitch_amp * np.sin(2 * np.pi * t / 3.7 + ph[3])
        roll = roll_amp * np.sin(2 * np.pi * t / 4.3 + ph[4])
        self.R = [_rot_zyx(yaw[i], pitch[i], roll[i]) for i in range(n_sweeps)]
        self.t = np.stack([x, y, z], -1)
        # ---- scene placed along the path corridor
        rs = np.random.default_rng(seed_scene)
        path_len = float(np.sum(v) * dt) + 160.0
        boxes, cyls = [], []
        yaw0, yawN = yaw[0], yaw[-1]

        def path_point(s):  # arc length -> (pos, left normal); extrapolates straight before/after the run
            s0 = s - 80.0
            arc = np.concatenate([[0.0], np.cumsum(v[:-1] * dt)])
            if s0 <= 0:
                p = np.array([x[0], y[0]]) + s0 * np.array([np.cos(yaw0), np.sin(yaw0)])
                h = yaw0
            elif s0 >= arc[-1]:
                p = np.array([x[-1], y[-1]]) + (s0 - arc[-1]) * np.array([np.cos(yawN), np.sin(yawN)])
                h = yawN
            else:
                i = int(np.searchsorted(arc, s0)) - 1
                f = (s0 - arc[i]) / max(arc[i + 1] - arc[i], 1e-9)
                p = np.array([x[i] + f * (x[i + 1] - x[i]), y[i] + f * (y[i + 1] - y[i])])
                h = yaw[i]
            return p, np.array([-np.sin(h), np.cos(h)])

        s = 0.0
        side = 1.0
        while s < path_len:
            p, nrm = path_point(s)
            sx, sy, sz = rs.uniform(2, 10), rs.uniform(2, 10), rs.uniform(2.5, 8)
            off = rs.uniform(7.0, 22.0) + 0.5 * max(sx, sy)
            c = p + side * off * nrm
            boxes.append([c[0] - sx / 2, c[1] - sy / 2, GROUND_Z, c[0] + sx / 2, c[1] + sy / 2, GROUND_Z + sz])
            side = -side
            s += rs.uniform(6.0, 10.0)
        s = 2.0
        side = -1.0
        while s < path_len:
            p, nrm = path_point(s)
            off = rs.uniform(4.0, 15.0)
            c = p + side * off * nrm
            cyls.append([c[0], c[1], 0.15, GROUND_Z, GROUND_Z + rs.uniform(3.0, 6.0)])
            side = -side
            s += rs.uniform(3.0, 5.0)
        self.boxes = np.array(boxes)
        self.cyls = np.array(cyls)

    # ------------------------------------------------------------------ poses
    def pose(self, k):
        return self.R[k], self.t[k]

    def gt_relative(self, k):
        """(q_xyzw, t) of T_{k-1}^{-1} T_k, i.e. p_last = q * p_curr + t — what LaserOdometry estimates."""
        R0, t0 = self.pose(k - 1)
        R1, t1 = self.pose(k)
        R = R0.T @ R1
        return rot_to_quat_xyzw(R), R0.T @ (t1 - t0)

    def gt_world(self, k):
        """Pose of sweep k in sweep 0's frame."""
        R0, t0 = self.pose(0)
        R1, t1 = self.pose(k)
        return rot_to_quat_xyzw(R0.T @ R1), R0.T @ (t1 - t0)

    # ------------------------------------------------------------------ ray casting
    def _az_window(self, pts_world, R, o, margin_cols=3):
        """Column range [j0, j1) (mod n_azimuth) whose rays can reach the object with the given world corner points."""
        n_az = self.n_azimuth
        ps = (pts_world - o[None, :]) @ R  # sensor frame
        if np.min(np.hypot(ps[:, 0], ps[:, 1])) < 1.0:
            return 0, n_az
        az = np.arctan2(ps[:, 1], ps[:, 0])
        ref = az[0]
        d = (az - ref + np.pi) % (2 * np.pi) - np.pi  # spread around the first corner, wrap-safe for spans < pi
        lo, hi = ref + d.min(), ref + d.max()
        if hi - lo > np.pi * 0.9:
            return 0, n_az
        # column j has azimuth -2*pi*j/n_az (roll/pitch <= 0.01 rad shift it by < 1 column at these elevations)
        j_hi = int(np.ceil(-lo / (2 * np.pi) * n_az)) + margin_cols
        j_lo = int(np.floor(-hi / (2 * np.pi) * n_az)) - margin_cols
        return j_lo, j_hi + 1

    def ranges(self, k):
        R, o = self.pose(k)
        nr, na = self.n_rings, self.n_azimuth
        d = (self.dirs @ R.T).reshape(nr, na, 3)  # world directions, [ring][column]
        best = np.full((nr, na), np.inf)
        # ground
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (GROUND_Z - o[2]) / d[:, :, 2]
        tg = np.where((d[:, :, 2] < 0) & (tg > 0), tg, np.inf)
        best = np.minimum(best, tg)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d

        def cols(j0, j1):
            if j1 - j0 >= na:
                return [slice(0, na)]
            a, b = j0 % na, j1 % na
            return [slice(a, b)] if a < b else [slice(a, na), slice(0, b)]

        # boxes (slab test) — only the columns that can see the box
        bx = self.boxes
        cen = 0.5 * (bx[:, :2] + bx[:, 3:5])
        near = np.linalg.norm(cen - o[None, :2], axis=1) < MAX_RANGE + 10.0
        for b in bx[near]:
            corners = np.array([[b[i], b[j], b[kk]] for i in (0, 3) for j in (1, 4) for kk in (2, 5)])
            j0, j1 = self._az_window(corners, R, o)
            for sl in cols(j0, j1):
                t1 = (b[None, None, 0:3] - o[None, None, :]) * inv[:, sl]
                t2 = (b[None, None, 3:6] - o[None, None, :]) * inv[:, sl]
                tmin = np.nanmax(np.minimum(t1, t2), axis=2)
                tmax = np.nanmin(np.maximum(t1, t2), axis=2)
                hit = (tmax >= tmin) & (tmin > 0)
                bs = best[:, sl]
                best[:, sl] = np.where(hit & (tmin < bs), tmin, bs)
        # vertical cylinders
        cy = self.cyls
        near = np.linalg.norm(cy[:, :2] - o[None, :2], axis=1) < MAX_RANGE + 2.0
        for c in cy[near]:
            corners = np.array([[c[0] + sx * c[2], c[1] + sy * c[2], z] for sx in (-1, 1) for sy in (-1, 1) for z in (c[3], c[4])])
            j0, j1 = self._az_window(corners, R, o)
            ox, oy = o[0] - c[0], o[1] - c[1]
            for sl in cols(j0, j1):
                dd = d[:, sl]
                a = dd[:, :, 0] ** 2 + dd[:, :, 1] ** 2
                bq = 2 * (ox * dd[:, :, 0] + oy * dd[:, :, 1])
                cq = ox * ox + oy * oy - c[2] ** 2
                disc = bq * bq - 4 * a * cq
                with np.errstate(invalid="ignore", divide="ignore"):
                    tc = (-bq - np.sqrt(disc)) / (2 * a)
                zc = o[2] + tc * dd[:, :, 2]
                hit = (disc > 0) & (tc > 0) & (zc >= c[3]) & (zc <= c[4])
                bs = best[:, sl]
                best[:, sl] = np.where(hit & (tc < bs), tc, bs)
        return best.reshape(-1)

    def sweep(self, k):
        """float32 [n_rings*n_azimuth, 4] ring-major cloud of sweep k in the sensor frame."""
        rng = np.random.default_rng(self.seed_noise + k)
        r = self.ranges(k)
        r = r + rng.normal(0.0, self.noise_sigma, r.shape[0]) if self.noise_sigma > 0 else r
        miss = ~np.isfinite(r) | (r > MAX_RANGE) | (r <= 0.05)
        with np.errstate(invalid="ignore"):
            pts = self.dirs * r[:, None]
        out = np.zeros((pts.shape[0], 4), dtype=np.float32)
        out[:, :3] = pts.astype(np.float32)
        out[miss, :3] = np.nan
        return out


# ---------------------------------------------------------------------------------------------------- camera (config 4)
def kitti_like_calib():
    """(cam_T_velo 4x4, rect0_T_cam 4x4 with only the 3x3 block filled — the ROS path leaves (3,3) = 0,
    visual_odometry.cpp:140-144 — and P_rect0 3x4), f32, KITTI-like pinhole 1242 x 375."""
    cam_T_velo = np.array([[0, -1, 0, -0.004], [0, 0, -1, -0.076], [1, 0, 0, -0.272], [0, 0, 0, 1]], dtype=np.float32)
    rect0_T_cam = np.zeros((4, 4), dtype=np.float32)
    rect0_T_cam[:3, :3] = np.eye(3, dtype=np.float32)
    P = np.array([[718.856, 0, 607.1928, 0], [0, 718.856, 185.2157, 0], [0, 0, 1, 0]], dtype=np.float32)
    return cam_T_velo, rect0_T_cam, P


def kitti_like_extrinsics():
    """(base_T_cam0, velo_T_cam0) 4x4 f64 as VloamTF::processStaticTransform derives them (vloam_tf.cpp:55-56):
    base_T_cam0 = base_T_imu * imu_T_cam0, velo_T_cam0 = imu_T_velo^-1 * imu_T_cam0, with KITTI-like statics (kitti2bag's
    base_link -> imu_link offset, the raw-data imu -> velodyne mount) and the camera of kitti_like_calib()."""
    cam_T_velo = kitti_like_calib()[0].astype(np.float64)
    velo_T_cam0 = np.linalg.inv(cam_T_velo)
    imu_T_velo = np.eye(4)
    imu_T_velo[:3, 3] = [0.81, -0.32, 0.80]
    imu_T_velo[:3, :3] = _rot_zyx(0.0148, -0.0021, 0.0009)
    base_T_imu = np.eye(4)
    base_T_imu[:3, 3] = [-1.405, 0.32, 0.93]
    imu_T_cam0 = imu_T_velo @ velo_T_cam0
    return base_T_imu @ imu_T_cam0, np.linalg.inv(imu_T_velo) @ imu_T_cam0


def _hash01(ix, iy, seed):
    """Integer lattice -> [0, 1): a fixed 64-bit mix (the same values on every machine)."""
    h = (ix.astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (iy.astype(np.int64).astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) ^ \
        np.uint64((seed * 0x165667B19E3779F9) & 0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(29); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _value_noise(u, v, cell, seed):
    fu, fv = u / cell, v / cell
    iu, iv = np.floor(fu), np.floor(fv)
    tu, tv = fu - iu, fv - iv
    tu, tv = tu * tu * (3 - 2 * tu), tv * tv * (3 - 2 * tv)
    iu, iv = iu.astype(np.int64), iv.astype(np.int64)
    a, b = _hash01(iu, iv, seed), _hash01(iu + 1, iv, seed)
    c, d = _hash01(iu, iv + 1, seed), _hash01(iu + 1, iv + 1, seed)
    return (a * (1 - tu) + b * tu) * (1 - tv) + (c * (1 - tu) + d * tu) * tv


def render_image(seq, k, width=1242, height=375):
    """uint8 [height, width] grey image of sweep k's scene through the camera of kitti_like_calib(): the same ground / boxes / cylinders the
    LiDAR sees, with a procedural surface texture (value noise in surface coordinates, octaves faded out below the pixel footprint) —
    so that corners tracked between two frames move the way the geometry says."""
    cam_T_velo, _, P = kitti_like_calib()
    K = P[:, :3].astype(np.float64)
    Tcv = cam_T_velo.astype(np.float64)
    Rcv, tcv = Tcv[:3, :3], Tcv[:3, 3]
    R, o = seq.pose(k)
    us, vs = np.meshgrid(np.arange(width, dtype=np.float64) * (1242.0 / width) + 0.5 * (1242.0 / width - 1), np.arange(height, dtype=np.float64) * (375.0 / height) + 0.5 * (375.0 / height - 1))
    dc = np.stack([(us - K[0, 2]) / K[0, 0], (vs - K[1, 2]) / K[1, 1], np.ones_like(us)], -1).reshape(-1, 3)
    dc /= np.linalg.norm(dc, axis=1, keepdims=True)
    dw = (dc @ Rcv) @ R.T                      # cam -> velo (Rcv^T d) -> world (R d)
    ow = o + R @ (-Rcv.T @ tcv)
    n = dw.shape[0]
    best = np.full(n, np.inf)
    tu, tv = np.zeros(n), np.zeros(n)          # texture coordinates of the hit
    shade = np.zeros(n)
    sid = np.zeros(n, dtype=np.int64)          # texture seed per surface
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dw
        tg = (GROUND_Z - ow[2]) / dw[:, 2]
    hit = (dw[:, 2] < 0) & (tg > 0) & (tg < 2.0 * MAX_RANGE)
    best = np.where(hit, tg, best)
    pg = ow[None, :] + tg[:, None] * dw
    tu, tv = np.where(hit, pg[:, 0], tu), np.where(hit, pg[:, 1], tv)
    shade = np.where(hit, 0.55, shade)
    bx = seq.boxes
    cen = 0.5 * (bx[:, :2] + bx[:, 3:5])
    fwd = (R @ np.array([1.0, 0, 0]))[:2]
    rel = cen - ow[None, :2]
    near = (np.linalg.norm(rel, axis=1) < MAX_RANGE + 10.0) & (rel @ fwd > -12.0)
    for bi in np.nonzero(near)[0]:
        b = bx[bi]
        t1 = (b[None, 0:3] - ow[None, :]) * inv
        t2 = (b[None, 3:6] - ow[None, :]) * inv
        tlo = np.minimum(t1, t2)
        tmin = np.nanmax(tlo, axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        h = (tmax >= tmin) & (tmin > 0) & (tmin < best)
        if not h.any():
            continue
        ax = np.nanargmax(tlo[h], axis=1)       # entry face
        ph = ow[None, :] + tmin[h, None] * dw[h]
        best[h] = tmin[h]
        tu[h] = np.where(ax == 0, ph[:, 1], ph[:, 0])
        tv[h] = np.where(ax == 2, ph[:, 1], ph[:, 2])
        shade[h] = np.where(ax == 0, 0.85, np.where(ax == 1, 0.7, 1.0))
        sid[h] = 1 + 3 * bi + ax
    cy = seq.cyls
    rel = cy[:, :2] - ow[None, :2]
    near = (np.linalg.norm(rel, axis=1) < 60.0) & (rel @ fwd > -3.0)
    for ci in np.nonzero(near)[0]:
        c = cy[ci]
        ox, oy = ow[0] - c[0], ow[1] - c[1]
        a = dw[:, 0] ** 2 + dw[:, 1] ** 2
        bq = 2 * (ox * dw[:, 0] + oy * dw[:, 1])
        cq = ox * ox + oy * oy - c[2] ** 2
        disc = bq * bq - 4 * a * cq
        with np.errstate(invalid="ignore", divide="ignore"):
            tc = (-bq - np.sqrt(disc)) / (2 * a)
        zc = ow[2] + tc * dw[:, 2]
        h = (disc > 0) & (tc > 0) & (zc >= c[3]) & (zc <= c[4]) & (tc < best)
        if not h.any():
            continue
        ph = ow[None, :] + tc[h, None] * dw[h]
        best[h] = tc[h]
        tu[h] = np.arctan2(ph[:, 1] - c[1], ph[:, 0] - c[0]) * c[2]
        tv[h] = ph[:, 2]
        shade[h] = 0.9
        sid[h] = 100000 + ci
    ok = np.isfinite(best)
    foot = np.where(ok, best, 1.0) / K[0, 0] * (1242.0 / width)     # metres per pixel at the hit
    val = np.zeros(n)
    wsum = np.zeros(n)
    amp = 1.0
    for cell in (3.0, 1.1, 0.4, 0.15):
        wgt = amp * np.clip((cell / np.maximum(foot, 1e-9) - 3.0) / 3.0, 0.0, 1.0)   # an octave needs >= 3 pixels per cell, full weight from 6
        if wgt.max() > 0:
            val += wgt * _value_noise(tu, tv, cell, 17)
            wsum += wgt
        amp *= 0.7
    tex = np.where(wsum > 0, val / np.maximum(wsum, 1e-9), 0.5)
    # hard-edged tiles (constant per cell): their junctions are the corners Shi-Tomasi finds; faded out like the noise octaves
    for cell, a_t in ((1.3, 0.5), (0.45, 0.35)):
        wgt = np.clip((cell / np.maximum(foot, 1e-9) - 4.0) / 4.0, 0.0, 1.0)
        tile = _hash01(np.floor(tu / cell), np.floor(tv / cell), 23)
        tex = tex + a_t * wgt * (tile - 0.5)
    # per-surface brightness offset: edges between faces / objects are corners too
    base = 0.25 + 0.5 * _hash01(sid, sid * 0 + 3, 5)
    g = np.where(ok, shade * (0.45 * base + 0.75 * tex), 0.0)
    sky = 0.75 + 0.2 * (vs.reshape(-1) / 375.0)
    g = np.where(ok, g, sky)
    return np.clip(np.rint(255.0 * g / 1.3), 0, 255).astype(np.uint8).reshape(height, width)


def synth_matches(seq, k, n_match=1400, pixel_noise=0.5, seed=99):
    """Pixel pairs (prev frame k-1 -> current frame k) of scene points visible in both images, standing in for the
    OpenCV front-end (out of scope): integer (truncated) pixel coordinates like visual_odometry.cpp:283-294."""
    cam_T_velo, _, P = kitti_like_calib()
    K = P[:, :3].astype(np.float64)
    Tcv = cam_T_velo.astype(np.float64)
    rng = np.random.default_rng(seed + k)
    prev = seq.sweep(k - 1)[:, :3].astype(np.float64)
    prev = prev[np.isfinite(prev[:, 0])]
    q, t = seq.gt_relative(k)          # p_prev = R p_curr + t
    R = quat_to_rot(q)
    curr = (prev - t[None, :]) @ R      # R^T (p_prev - t)
    def proj(pts):
        pc = pts @ Tcv[:3, :3].T + Tcv[:3, 3]
        uv = pc @ K.T
        return uv[:, :2] / uv[:, 2:3], pc[:, 2]
    uv0, z0 = proj(prev)
    uv1, z1 = proj(curr)
    ok = (z0 > 1.0) & (z1 > 1.0) & (uv0[:, 0] > 2) & (uv0[:, 0] < 1239) & (uv0[:, 1] > 2) & (uv0[:, 1] < 372) & \
         (uv1[:, 0] > 2) & (uv1[:, 0] < 1239) & (uv1[:, 1] > 2) & (uv1[:, 1] < 372)
    idx = np.nonzero(ok)[0]
    idx = rng.choice(idx, size=min(n_match, idx.size), replace=False)
    a = uv0[idx] + rng.normal(0, pixel_noise, (idx.size, 2))
    b = uv1[idx] + rng.normal(0, pixel_noise, (idx.size, 2))
    return a.astype(np.float32).astype(np.int32), b.astype(np.float32).astype(np.int32)


# ---------------------------------------------------------------------------------------------------- images (config 4 front-end)
def synth_texture(w, h, seed=7, octaves=(96, 48, 24, 12, 6)):
    """Smooth random texture, float64 [h, w] in [0, 255]: multi-octave value noise (bilinear) plus a few hard-edged rectangles —
    smooth enough for Lucas-Kanade, cornered enough for Shi-Tomasi."""
    rng = np.random.default_rng(seed)
    ys, xs = np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64)
    img = np.zeros((h, w))
    amp = 1.0
    for cs in octaves:
        g = rng.random((h // cs + 3, w // cs + 3))
        fy, fx = ys / cs, xs / cs
        iy, ix = fy.astype(int), fx.astype(int)
        ty, tx = (fy - iy)[:, None], (fx - ix)[None, :]
        a = g[iy][:, ix]; b = g[iy][:, ix + 1]; c = g[iy + 1][:, ix]; d = g[iy + 1][:, ix + 1]
        img += amp * ((a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty)
        amp *= 0.6
    for _ in range(max(8, w * h // 700)):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        ww, hh = int(rng.integers(5, 28)), int(rng.integers(5, 22))
        img[y0:y0 + hh, x0:x0 + ww] += rng.uniform(0.25, 0.5) * (1 if rng.random() < 0.5 else -1)
    img -= img.min()
    return img * (255.0 / img.max())


def warp_image(canvas, w, h, A, t):
    """uint8 [h, w] view of a float canvas: pixel (x, y) shows canvas at A @ (x, y) + t (bilinear), i.e. content moves by the inverse."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = A[0, 0] * xs + A[0, 1] * ys + t[0]
    sy = A[1, 0] * xs + A[1, 1] * ys + t[1]
    sx = np.clip(sx, 0, canvas.shape[1] - 1.001); sy = np.clip(sy, 0, canvas.shape[0] - 1.001)
    ix, iy = sx.astype(int), sy.astype(int)
    tx, ty = sx - ix, sy - iy
    v = (canvas[iy, ix] * (1 - tx) + canvas[iy, ix + 1] * tx) * (1 - ty) + (canvas[iy + 1, ix] * (1 - tx) + canvas[iy + 1, ix + 1] * tx) * ty
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def synth_image_pair(w=1242, h=375, seed=7, shift=(3.6, -1.3), rot=0.004, scale=1.003):
    """(prev, next, flow function): two uint8 views of one texture; a point at (x, y) in prev appears at flow(x, y) in next."""
    m = 64
    canvas = synth_texture(w + 2 * m, h + 2 * m, seed)
    I2 = np.eye(2)
    prev = warp_image(canvas, w, h, I2, (m, m))
    c, s = np.cos(rot) * scale, np.sin(rot) * scale
    B = np.array([[c, -s], [s, c]])                     # next(x) = canvas(B x + tb)  ->  prev point p shows at B^-1 (p + m - tb)
    tb = np.array([m - shift[0], m - shift[1]]) - (B - I2) @ np.array([w / 2, h / 2])
    nxt = warp_image(canvas, w, h, B, tb)
    Binv = np.linalg.inv(B)
    flow = lambda p: (np.asarray(p, dtype=np.float64) + m - tb) @ Binv.T   # noqa: E731
    return prev, nxt, flow


def orb_test_pattern(seed=31):
    """A seeded stand-in for OpenCV's learned ORB sampling table (orb.cpp: bit_pattern_31_ — library data that is not in the reference tree, handed
    in by the caller: vloam_vo_set_orb_pattern): int8 [256, 4] = (x0, y0, x1, y1) per test, Gaussian around the patch centre like the original's
    distribution, inside the 31 x 31 patch, the two points of a test distinct."""
    rng = np.random.default_rng(seed)
    pat = np.zeros((256, 4), np.int8)
    for k in range(256):
        while True:
            v = np.clip(np.rint(rng.normal(0.0, 5.5, 4)), -13, 13).astype(np.int8)
            if (v[0], v[1]) != (v[2], v[3]):
                pat[k] = v
                break
    return pat
