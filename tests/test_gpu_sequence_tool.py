"""GPU: the §8f harness end to end — sweeps from KITTI-format .bin files through the façade mirrors, LO/MO rows written in the
reference's results format, compared with rows derived from the oracle's poses through the same VloamTF algebra."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_run_sequence_rows_match_oracle(vl, orc, sweeps, tmp_path):
    kio = importlib.import_module("vloam_amd.kitti_io")
    vel = tmp_path / "velodyne_points" / "data"
    os.makedirs(vel)
    n = 5
    clouds = []
    for k in range(n):
        c = sweeps(64, 512, k)
        c = c[np.isfinite(c[:, 0])]
        kio.save_kitti_bin(vel / ("%010d.bin" % k), c)
        clouds.append(kio.load_kitti_bin(vel / ("%010d.bin" % k)))
    out = tmp_path / "res"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sequence.py"), "--velodyne", str(vel), "--out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lo = kio.read_trajectory(out / "LO0.txt")
    mo = kio.read_trajectory(out / "MO0.txt")
    assert lo.shape == (n, 4, 4) and mo.shape == (n, 4, 4)

    o = orc.Oracle(scan_line=64, mapping_skip_frame=2)
    tf = kio.VloamTF(kio.make_T([0, 0, 0.0074, 0.99997], [0.81, -0.32, 0.80]), kio.make_T([0.5, -0.5, 0.5, -0.5], [1.08, -0.32, 0.72]))
    for k, c in enumerate(clouds):
        o.process(c)
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        row_lo = tf.LO2Cam0StartFrame(qw, tw, k)
        row_mo = tf.MO2Cam0StartFrame(qm, tm, k)
        # "%f" keeps 6 decimals: rows agree to the printed precision (poses agree to ~1e-9)
        assert np.allclose(lo[k, :3], row_lo[:3], atol=2e-6), k
        assert np.allclose(mo[k, :3], row_mo[:3], atol=2e-6), k


@pytest.mark.gpu
def test_run_sequence_coupled_frames_and_metrics(tmp_path):
    """--vloam: the coupled VO + LiDAR frame loop over a synthetic sequence, VO0 / LO0 / MO0 rows + one JSON line of metrics per frame."""
    import json
    out = tmp_path / "res"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sequence.py"), "--synthetic", "6", "--azimuth", "512", "--vloam",
                        "--metrics", str(tmp_path / "frames.jsonl"), "--out", str(out), "--mapping-skip-frame", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    kio = importlib.import_module("vloam_amd.kitti_io")
    vo, lo, mo = [kio.read_trajectory(out / ("%s0.txt" % k)) for k in ("VO", "LO", "MO")]
    assert vo.shape == lo.shape == mo.shape == (6, 4, 4)
    assert np.allclose(vo[0], np.eye(4)) and np.allclose(lo[0], np.eye(4))          # rows are relative to the start frame
    d = np.linalg.norm(vo[5, :3, 3] - lo[5, :3, 3])
    assert np.linalg.norm(lo[5, :3, 3]) > 3.0 and d < 0.5, (vo[5, :3, 3], lo[5, :3, 3])  # 5 m of travel, VO and LO chains agree to decimetres
    rows = [json.loads(l) for l in open(tmp_path / "frames.jsonl")]
    assert len(rows) == 6 and rows[3]["vo"]["counter32"] > 100 and rows[3]["lo_round1"]["iterations"] >= 2
    assert rows[3]["counts"]["K_m"] > 100 and rows[3]["map_round1"]["residual_blocks"] == rows[3]["counts"]["K_m"]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sequence.py"), "--synthetic", "3", "--azimuth", "512",
                         "--metrics", str(tmp_path / "f2.jsonl"), "--out", str(tmp_path / "res2")], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    rows2 = [json.loads(l) for l in open(tmp_path / "f2.jsonl")]
    assert len(rows2) == 3 and rows2[2]["stage_ms"]["laserOdometry"] > 0


@pytest.mark.gpu
def test_run_sequence_from_rendered_images(tmp_path):
    """--vloam --images: every frame's pixel matches come from a rendered grey image through the device image front-end."""
    import json
    out = tmp_path / "res"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sequence.py"), "--synthetic", "4", "--azimuth", "512", "--vloam", "--images",
                        "--metrics", str(tmp_path / "frames.jsonl"), "--out", str(out), "--mapping-skip-frame", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    kio = importlib.import_module("vloam_amd.kitti_io")
    vo, lo = [kio.read_trajectory(out / ("%s0.txt" % k)) for k in ("VO", "LO")]
    assert vo.shape == lo.shape == (4, 4, 4)
    rows = [json.loads(l) for l in open(tmp_path / "frames.jsonl")]
    assert rows[0]["image"]["keypoints"] > 100 and rows[0]["image"]["tracked"] == 0
    assert rows[3]["image"]["tracked"] > 0.8 * rows[3]["image"]["keypoints"] and rows[3]["vo"]["counter32"] + rows[3]["vo"]["counter22"] > 100
    assert np.linalg.norm(lo[3, :3, 3]) > 2.0 and np.linalg.norm(vo[3, :3, 3] - lo[3, :3, 3]) < 0.5   # 3 m of travel; VO (from image flow) follows LO


@pytest.mark.gpu
def test_run_sequence_on_a_kitti_raw_layout_with_images(tmp_path, synth):
    """configs[3] from files: velodyne .bin sweeps, image_00 PNGs and the two KITTI calibration text files (what
    PointCloudUtil::loadTransformations parses, point_cloud_util.cpp:5-116) go through run_sequence.py --vloam --images and give
    the same VO / LO / MO rows as the in-memory synthetic run of the same drive."""
    kio = importlib.import_module("vloam_amd.kitti_io")
    n = 3
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1)
    vel, img = tmp_path / "velodyne_points" / "data", tmp_path / "image_00" / "data"
    os.makedirs(vel); os.makedirs(img)
    for k in range(n):
        c = seq.sweep(k)
        kio.save_kitti_bin(vel / ("%010d.bin" % k), c[np.isfinite(c[:, 0])])
        kio.save_png_gray(img / ("%010d.png" % k), synth.render_image(seq, k))
    cam_T_velo, _, P = synth.kitti_like_calib()
    fmt = lambda a: " ".join("%.9e" % float(v) for v in np.asarray(a).reshape(-1))   # noqa: E731
    (tmp_path / "calib_velo_to_cam.txt").write_text("calib_time: 15-Mar-2012 11:37:16\nR: %s\nT: %s\n" % (fmt(cam_T_velo[:3, :3]), fmt(cam_T_velo[:3, 3])))
    (tmp_path / "calib_cam_to_cam.txt").write_text("calib_time: 09-Jan-2012 13:57:47\nR_rect_00: %s\nP_rect_00: %s\n" % (fmt(np.eye(3)), fmt(P)))
    common = ["--vloam", "--images", "--mapping-skip-frame", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sequence.py"), "--velodyne", str(vel), "--image-dir", str(img),
                        "--calib-cam-to-cam", str(tmp_path / "calib_cam_to_cam.txt"), "--calib-velo-to-cam", str(tmp_path / "calib_velo_to_cam.txt"),
                        "--out", str(tmp_path / "a")] + common, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    # the in-memory run drops nothing either: NaN misses are removed by scan registration itself, the .bin writer removed them before
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sequence.py"), "--synthetic", str(n), "--azimuth", "512", "--out", str(tmp_path / "b")] + common,
                        capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    for name in ("VO0.txt", "LO0.txt", "MO0.txt"):
        a, b = kio.read_trajectory(tmp_path / "a" / name), kio.read_trajectory(tmp_path / "b" / name)
        assert a.shape == (n, 4, 4) and np.allclose(a, b, atol=2e-6), name
    assert np.array_equal(kio.load_png_gray(img / "0000000001.png"), synth.render_image(seq, 1))
    # an RGB / 16-bit PNG is refused with a message, not mis-read
    import struct
    import zlib
    ch = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)   # noqa: E731
    (tmp_path / "rgb.png").write_bytes(b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 0)) + ch(b"IDAT", zlib.compress(b"\0" * 14)) + ch(b"IEND", b""))
    with pytest.raises(ValueError):
        kio.load_png_gray(tmp_path / "rgb.png")
