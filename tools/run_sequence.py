#!/usr/bin/env python3
"""Run the LiDAR odometry + mapping hot path over a directory of KITTI raw sweeps (or a synthetic sequence) and write
LO0.txt / MO0.txt in the reference's results format — the part of vloam_main_node's callback
(src/vloam_main/src/vloam_main_node.cpp:134-176, 215-222) that sits either side of the HIP path, without ROS.

  python tools/run_sequence.py --velodyne /data/2011_09_26_drive_0001_sync/velodyne_points/data --out results/
  python tools/run_sequence.py --synthetic 50 --out /tmp/res        # needs an MI355X either way
  python tools/run_sequence.py --synthetic 50 --vloam --metrics /tmp/res/frames.jsonl --out /tmp/res
      --vloam:   the coupled per-frame loop (vloam_process_frame: VO solve -> LO prior -> SR -> LO -> VO prior -> mapping, combined mode)
                 on synthetic pixel matches, also writes VO0.txt
      --images:  with --vloam: the matches come from grey camera images instead (vloam_process_frame_image: Shi-Tomasi corners +
                 pyramidal Lucas-Kanade on the device, the reference's optical_flow_match = true); the synthetic sequence's images are
                 rendered from the same scene (synth.render_image).  Real data: --velodyne DIR --image-dir .../image_00/data
                 --calib-cam-to-cam calib_cam_to_cam.txt --calib-velo-to-cam calib_velo_to_cam.txt (KITTI raw layout, 8-bit grey PNGs)
  python tools/run_sequence.py --vloam --images --velodyne DRIVE/velodyne_points/data --image-dir DRIVE/image_00/data \
      --calib-cam-to-cam CALIB/calib_cam_to_cam.txt --calib-velo-to-cam CALIB/calib_velo_to_cam.txt --out results/
      --metrics: one JSON object per frame — the reference prints these through ROS_INFO / TicToc: feature counts
                 (scan_registration.cpp), correspondences and solver iterations (laser_odometry.cpp:453-465, laser_mapping.cpp:606-618),
                 per-stage milliseconds
"""
import argparse
import glob
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    path = os.path.join(ROOT, "vloam-cmu-16833_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("vloam_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vloam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def unwrap_boundary_points(cloud, band=4e-6):
    """Points whose azimuth lies within `band` rad (about 2 float ulp at pi) of one of scanRegistration's unwrap thresholds
    (scan_registration.cpp:236-262).  The azimuth is an f32 atan2 whose last bit differs between math libraries (the device's, glibc's),
    so these are the points whose relTime could land one revolution apart between this library and a CPU run of the reference: the count
    bounds the per-frame 'threshold flips'.  Computed from the input cloud with numpy; no device state involved."""
    xyz = np.asarray(cloud, np.float32)[:, :3]
    ok = np.isfinite(xyz).all(axis=1) & ((xyz.astype(np.float64) ** 2).sum(axis=1) >= 0.1 ** 2)
    if not ok.any():
        return 0
    x, y = xyz[ok, 0], xyz[ok, 1]
    ori = -np.arctan2(y, x).astype(np.float64)
    start = float(ori[0])
    end = float(ori[-1]) + 2 * np.pi
    if end - start > 3 * np.pi:
        end -= 2 * np.pi
    elif end - start < np.pi:
        end += 2 * np.pi
    two_pi = 2 * np.pi
    thresholds = (start - np.pi / 2, start + 1.5 * np.pi, start + np.pi, start - np.pi, end - 1.5 * np.pi - two_pi, end + np.pi / 2 - two_pi,
                  end - 1.5 * np.pi, end + np.pi / 2)
    near = np.zeros(ori.shape, bool)
    for t in thresholds:
        d = np.abs(((ori - t) + np.pi) % two_pi - np.pi)
        near |= d < band
    return int(near.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--velodyne", help="directory of KITTI raw velodyne .bin sweeps")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic 64 x --azimuth sweeps instead")
    ap.add_argument("--azimuth", type=int, default=2048)
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--mapping-skip-frame", type=int, default=2)
    ap.add_argument("--vloam", action="store_true", help="coupled VO + LiDAR frames (synthetic sequences only: synthetic pixel matches)")
    ap.add_argument("--images", action="store_true", help="with --vloam: take the pixel matches from grey images (device image front-end)")
    ap.add_argument("--clahe", action="store_true", help="with --images: cv::createCLAHE(2.0) on every image first (the launch file's CLAHE parameter)")
    ap.add_argument("--image-dir", help="directory of 8-bit grey PNGs, one per sweep (KITTI raw image_00/data)")
    ap.add_argument("--calib-cam-to-cam", help="KITTI calib_cam_to_cam.txt (R_rect_00, P_rect_00)")
    ap.add_argument("--calib-velo-to-cam", help="KITTI calib_velo_to_cam.txt (R, T)")
    ap.add_argument("--metrics", help="write per-frame metrics as JSON lines to this file")
    ap.add_argument("--imu-T-velo", help="16 numbers, row major (default: KITTI 2011_09_26 extrinsics, approx.)")
    ap.add_argument("--imu-T-cam0", help="16 numbers, row major")
    a = ap.parse_args()
    vl = load_pkg()
    import importlib
    kio = importlib.import_module("vloam_amd.kitti_io")

    if a.velodyne:
        files = sorted(glob.glob(os.path.join(a.velodyne, "*.bin")))
        clouds = (kio.load_kitti_bin(f) for f in files)
        n = len(files)
    else:
        synth = importlib.import_module("vloam_amd.synth")
        seq = synth.SynthSequence(n_rings=64, n_azimuth=a.azimuth, n_sweeps=max(a.synthetic, 2) + 1)
        clouds = (seq.sweep(k) for k in range(a.synthetic))
        n = a.synthetic
    if n == 0:
        sys.exit("no sweeps")

    def mat(arg, default):
        return np.array([float(v) for v in arg.split()]).reshape(4, 4) if arg else default
    # static extrinsics of the KITTI raw rig (imu -> velodyne, imu -> cam0); identity base_T_imu
    imu_T_velo = mat(a.imu_T_velo, kio.make_T([0, 0, 0.0074, 0.99997], [0.81, -0.32, 0.80]))
    imu_T_cam0 = mat(a.imu_T_cam0, kio.make_T([0.5, -0.5, 0.5, -0.5], [1.08, -0.32, 0.72]))
    tf = kio.VloamTF(imu_T_velo, imu_T_cam0)

    real_images = None
    if a.vloam and a.velodyne:
        if not (a.images and a.image_dir and a.calib_cam_to_cam and a.calib_velo_to_cam):
            sys.exit("--vloam on recorded sweeps needs --images --image-dir DIR --calib-cam-to-cam FILE --calib-velo-to-cam FILE "
                     "(the pixel matches come from the images; ORB matching is not provided)")
        real_images = sorted(glob.glob(os.path.join(a.image_dir, "*.png")))
        if len(real_images) < n:
            sys.exit("%d sweeps but only %d images in %s" % (n, len(real_images), a.image_dir))
    if a.images and not a.vloam:
        sys.exit("--images belongs to the coupled loop: add --vloam")
    img_cfg = {}
    if a.images:
        ih, iw = (kio.load_png_gray(real_images[0]).shape if real_images else (375, 1242))
        img_cfg = dict(image_width=int(iw), image_height=int(ih), CLAHE=int(a.clahe))
    loam = vl.LidarOdometryMapping(device=a.device, mapping_skip_frame=a.mapping_skip_frame, detach_VO_LO=0 if a.vloam else 1,
                                   timing=1 if (a.metrics and not a.vloam) else 0, **img_cfg)
    hd = loam.hd
    if a.vloam:
        if real_images:   # PointCloudUtil::loadTransformations (point_cloud_util.cpp:5-116)
            hd.vo_set_calib(*kio.load_transformations(a.calib_cam_to_cam, a.calib_velo_to_cam))
        else:
            hd.vo_set_calib(*synth.kitti_like_calib())
        hd.set_extrinsics(tf.base_T_cam0, tf.velo_T_cam0)
    os.makedirs(a.out, exist_ok=True)
    lo_rows, mo_rows, vo_rows = [], [], []
    mf = open(a.metrics, "w") if a.metrics else None
    ms_prev = np.zeros(4)
    for count, cloud in enumerate(clouds):
        if a.vloam:
            if a.images:
                hd.process_frame_image(cloud, kio.load_png_gray(real_images[count]) if real_images else synth.render_image(seq, count))
            else:
                m = synth.synth_matches(seq, count) if count > 0 else (None, None)
                hd.process_frame(cloud, m[0], m[1])
            row = hd.trajectory(count, 1)[0]
            q_lo, t_lo, q_mo, t_mo = row[0:4], row[4:7], row[7:11], row[11:14]
            v = hd.vo_trajectory(count, 1)[0]
            tf.world_VOT_base_last = kio.make_T(v[0:4], v[4:7])
            vo_rows.append(tf.VO2Cam0StartFrame(count))
        else:
            loam.reset()
            loam.scanRegistrationIO(cloud)
            loam.laserOdometryIO()
            loam.laserMappingIO()
            lo, lm = loam.laser_odometry, loam.laser_mapping
            tf.LO2CamPrior(lo.q_last_curr, lo.t_last_curr)
            q_lo, t_lo, q_mo, t_mo = lo.q_w_curr, lo.t_w_curr, lm.q_w_curr, lm.t_w_curr
        lo_rows.append(tf.LO2Cam0StartFrame(q_lo, t_lo, count))
        mo_rows.append(tf.MO2Cam0StartFrame(q_mo, t_mo, count))
        if mf:
            c = hd.counts()
            rec = {"frame": count, "points_in": int(cloud.shape[0]), "counts": c, "unwrap_boundary_points": unwrap_boundary_points(cloud),
                   "lo_pose": [float(x) for x in list(q_lo) + list(t_lo)],
                   "map_pose": [float(x) for x in list(q_mo) + list(t_mo)]}
            if count > 0:
                for name, st, item in (("lo_round0", 1, 2), ("lo_round1", 1, 18), ("map_round0", 2, 3), ("map_round1", 2, 19)):
                    r = hd.debug_lm_record(st, item)
                    rec[name] = {"residual_blocks": r["n_factors"], "iterations": int(r["trace"].shape[0]), "evaluations": r["n_evals"],
                                 "initial_cost": r["initial_cost"], "final_cost": r["final_cost"], "termination": r["termination"]}
            if a.vloam and count > 0:
                rv = hd.vo_result()
                rec["vo"] = {"counter32": rv["counter32"], "counter22": rv["counter22"], "angles_0to1": [float(x) for x in rv["angles"]],
                             "t_0to1": [float(x) for x in rv["t"]]}
            if a.images:
                rec["image"] = {"keypoints": int(hd.vo_keypoints().shape[0]), "tracked": int(hd.vo_flow_matches()[0].shape[0])}
            if not a.vloam:
                ms, _ = hd.stage_ms()
                rec["stage_ms"] = {"scanRegistration": float(ms[0] - ms_prev[0]), "laserOdometry": float(ms[1] - ms_prev[1]),
                                   "laserMapping": float(ms[2] - ms_prev[2])}
                ms_prev = ms.copy()
            mf.write(json.dumps(rec) + "\n")
    if mf:
        mf.close()
    kio.write_trajectory(os.path.join(a.out, "LO0.txt"), lo_rows)
    kio.write_trajectory(os.path.join(a.out, "MO0.txt"), mo_rows)
    if vo_rows:
        kio.write_trajectory(os.path.join(a.out, "VO0.txt"), vo_rows)
    print("wrote %d rows to %s/{LO0,MO0%s}.txt" % (n, a.out, ",VO0" if vo_rows else ""))


if __name__ == "__main__":
    main()
