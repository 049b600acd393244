"""CPU: §8f rows — KITTI .bin / calibration formats, VloamTF algebra, trajectory rows (the format of the reference's
committed src/vloam_main/results/*/LO0.txt, of which tests/golden/ref_results_LO0_head.txt holds the first rows)."""
import importlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def kio(vl):
    return importlib.import_module("vloam_amd.kitti_io")


def test_bin_roundtrip(vl, tmp_path, sweeps):
    k = kio(vl)
    cloud = sweeps(64, 256, 0)
    fin = cloud[np.isfinite(cloud[:, 0])]
    p = tmp_path / "0000000000.bin"
    k.save_kitti_bin(p, fin, reflectance=np.linspace(0, 1, fin.shape[0], dtype=np.float32))
    assert os.path.getsize(p) == fin.shape[0] * 16               # float32[4] per point (point_cloud_util.cpp:118-146)
    back = k.load_kitti_bin(p)
    assert back.dtype == np.float32 and back.shape == fin.shape
    assert np.array_equal(back[:, :3], fin[:, :3]) and np.all(back[:, 3] == 0)


def test_calibration_parser(vl, tmp_path):
    k = kio(vl)
    (tmp_path / "calib_velo_to_cam.txt").write_text(
        "calib_time: 15-Mar-2012 11:37:16\n"
        "R: 7.533745e-03 -9.999714e-01 -6.166020e-04 1.480249e-02 7.280733e-04 -9.998902e-01 9.998621e-01 7.523790e-03 1.480755e-02\n"
        "T: -4.069766e-03 -7.631618e-02 -2.717806e-01\ndelta_f: 0.000000e+00 0.000000e+00\n")
    (tmp_path / "calib_cam_to_cam.txt").write_text(
        "calib_time: 09-Jan-2012 13:57:47\ncorner_dist: 9.950000e-02\n"
        "R_rect_00: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01\n"
        "P_rect_00: 7.215377e+02 0.000000e+00 6.095593e+02 0.000000e+00 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00\n"
        "R_rect_01: 1 0 0 0 1 0 0 0 1\n")
    cam_T_velo, rect0_T_cam, P = k.load_transformations(tmp_path / "calib_cam_to_cam.txt", tmp_path / "calib_velo_to_cam.txt")
    assert cam_T_velo.dtype == np.float32 and cam_T_velo[3, 3] == 1 and np.all(cam_T_velo[3, :3] == 0)
    assert cam_T_velo[0, 1] == np.float32(-9.999714e-01) and cam_T_velo[2, 3] == np.float32(-2.717806e-01)
    assert rect0_T_cam[3, 3] == 1 and rect0_T_cam[1, 1] == np.float32(9.999421e-01)
    assert P.shape == (3, 4) and P[0, 0] == np.float32(7.215377e+02) and P[1, 2] == np.float32(1.728540e+02) and P[2, 2] == 1


def test_tf_algebra_and_rows(vl, tmp_path):
    k = kio(vl)
    imu_T_velo = k.make_T([0.0, 0.0, 0.0087, 0.99996], [0.81, -0.32, 0.80])
    imu_T_cam0 = k.make_T([0.5, -0.5, 0.5, -0.5], [1.08, -0.31, 0.73])
    tf = k.VloamTF(imu_T_velo, imu_T_cam0)
    assert np.allclose(tf.velo_T_cam0, k.inv_T(imu_T_velo) @ imu_T_cam0)
    # VO -> LO prior: a camera-frame motion maps to the conjugate velodyne-frame motion, inverted (vloam_tf.cpp:61-63)
    cam_T = k.angle_axis_to_T([0.01, -0.02, 0.005], [0.02, -0.01, -1.0])
    q, t = tf.VO2VeloAndBase(cam_T)
    back = k.inv_T(tf.velo_T_cam0) @ k.make_T(q, t) @ tf.velo_T_cam0
    assert np.allclose(back, k.inv_T(cam_T), atol=1e-12)
    # LO -> VO prior (laser_odometry.cpp:563-567): feeding the VO-derived velodyne motion back gives the VO motion again
    # when base == velo; in general it is a rigid transform
    prior = tf.LO2CamPrior(q, t)
    assert np.allclose(prior[:3, :3] @ prior[:3, :3].T, np.eye(3), atol=1e-12)
    tf_same = k.VloamTF(np.eye(4), imu_T_cam0)      # velo == imu == base
    q2, t2 = tf_same.VO2VeloAndBase(cam_T)
    assert np.allclose(tf_same.LO2CamPrior(q2, t2), cam_T, atol=1e-12)
    # trajectory rows: frame 0 is the identity row; later rows are relative to it
    r0 = tf.LO2Cam0StartFrame([0, 0, 0, 1], [0, 0, 0], 0)
    assert k.format_pose_row(r0) == "1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000\n"
    r1 = tf.LO2Cam0StartFrame([0, 0, 0.01, 0.99995], [1.3, 0.0, 0.0], 1)
    assert r1.dtype == np.float32 and abs(np.linalg.det(r1[:3, :3].astype(np.float64)) - 1) < 1e-5
    assert abs(np.linalg.norm(r1[:3, 3]) - 1.3) < 0.05
    k.write_trajectory(tmp_path / "LO0.txt", [r0, r1])
    assert k.read_trajectory(tmp_path / "LO0.txt").shape == (2, 4, 4)
    # the reference's own result files parse with the same reader and re-print identically
    gold = os.path.join(HERE, "golden", "ref_results_LO0_head.txt")
    ref = k.read_trajectory(gold)
    assert ref.shape == (4, 4, 4) and np.allclose(ref[0], np.eye(4))
    with open(gold) as f:
        for line, T in zip(f, ref):
            assert k.format_pose_row(T) == line


def test_png_gray_reader_all_row_filters(tmp_path):
    """load_png_gray against a PNG whose rows use all five filter types (None, Sub, Up, Average, Paeth), encoded here."""
    import importlib
    import struct
    import zlib
    import conftest
    conftest.load_pkg()
    kio = importlib.import_module("vloam_amd.kitti_io")
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(23, 41), dtype=np.uint8)
    rows = []
    prev = np.zeros(41, dtype=np.int64)
    for y in range(23):
        cur = img[y].astype(np.int64)
        left = np.concatenate([[0], cur[:-1]])
        ul = np.concatenate([[0], prev[:-1]])
        f = y % 5
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - left
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        rows.append(bytes([f]) + (enc & 255).astype(np.uint8).tobytes())
        prev = cur
    ch = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)   # noqa: E731
    raw = zlib.compress(b"".join(rows))
    path = tmp_path / "f.png"
    path.write_bytes(b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", 41, 23, 8, 0, 0, 0, 0)) + ch(b"IDAT", raw[:50]) + ch(b"IDAT", raw[50:]) + ch(b"IEND", b""))
    assert np.array_equal(kio.load_png_gray(path), img)
    kio.save_png_gray(tmp_path / "g.png", img)
    assert np.array_equal(kio.load_png_gray(tmp_path / "g.png"), img)
    # a file without an IHDR chunk is refused with a ValueError (not a NameError further down)
    good = open(tmp_path / "g.png", "rb").read()
    n_ihdr = 12 + 13
    (tmp_path / "bad.png").write_bytes(good[:8] + good[8 + n_ihdr:])
    with pytest.raises(ValueError):
        kio.load_png_gray(tmp_path / "bad.png")
