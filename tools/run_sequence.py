#!/usr/bin/env python3
"""Run the LiDAR odometry + mapping hot path over a directory of KITTI raw sweeps (or a synthetic sequence) and write
LO0.txt / MO0.txt in the reference's results format — the part of vloam_main_node's callback
(src/vloam_main/src/vloam_main_node.cpp:134-176, 215-222) that sits either side of the HIP path, without ROS.

  python tools/run_sequence.py --velodyne /data/2011_09_26_drive_0001_sync/velodyne_points/data --out results/
  python tools/run_sequence.py --synthetic 50 --out /tmp/res        # needs an MI355X either way
"""
import argparse
import glob
import importlib.util
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--velodyne", help="directory of KITTI raw velodyne .bin sweeps")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic 64x2048 sweeps instead")
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--mapping-skip-frame", type=int, default=2)
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
        seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=max(a.synthetic, 2))
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

    loam = vl.LidarOdometryMapping(device=a.device, mapping_skip_frame=a.mapping_skip_frame, detach_VO_LO=1)
    os.makedirs(a.out, exist_ok=True)
    lo_rows, mo_rows = [], []
    for count, cloud in enumerate(clouds):
        loam.reset()
        loam.scanRegistrationIO(cloud)
        loam.laserOdometryIO()
        loam.laserMappingIO()
        lo, lm = loam.laser_odometry, loam.laser_mapping
        tf.LO2CamPrior(lo.q_last_curr, lo.t_last_curr)
        lo_rows.append(tf.LO2Cam0StartFrame(lo.q_w_curr, lo.t_w_curr, count))
        mo_rows.append(tf.MO2Cam0StartFrame(lm.q_w_curr, lm.t_w_curr, count))
    kio.write_trajectory(os.path.join(a.out, "LO0.txt"), lo_rows)
    kio.write_trajectory(os.path.join(a.out, "MO0.txt"), mo_rows)
    print("wrote %d rows to %s/{LO0,MO0}.txt" % (n, a.out))


if __name__ == "__main__":
    main()
