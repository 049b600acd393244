#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run in the dev container: `python tests/golden/make_golden.py`).

The reference has no tests and no golden vectors and cannot be built or imported here, so these
fixtures pin the ORACLE's outputs (regression vectors): the CPU tests check the oracle still reproduces them,
the GPU tests check the HIP path reproduces them.  Inputs are stored too (the small case), so nothing depends
on numpy's random stream staying stable.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import conftest  # noqa: E402
import orc  # noqa: E402


def run_case(n_rings, n_az, n_frames, store_inputs):
    synth = conftest.load_synth()
    seq = synth.SynthSequence(n_rings=n_rings, n_azimuth=n_az, n_sweeps=n_frames)
    o = orc.Oracle(scan_line=n_rings, with_mapping=True)
    out = {}
    for k in range(n_frames):
        cloud = seq.sweep(k)
        if store_inputs:
            out["in_%d" % k] = cloud[:, :3].copy()
        assert o.process(cloud) == 0
        pre = "f%d_" % k
        out[pre + "N2"] = np.int32(o.cloud(0).shape[0])
        out[pre + "ring_start"] = o.sr_ints(3)
        out[pre + "ring_end"] = o.sr_ints(4)
        out[pre + "sharpInd"] = o.sr_ints(5)
        out[pre + "lessSharpInd"] = o.sr_ints(6)
        out[pre + "flatInd"] = o.sr_ints(7)
        out[pre + "n_lessFlat"] = np.int32(o.cloud(4).shape[0])
        out[pre + "lessFlat_xyz_sum"] = o.cloud(4)[:, :3].astype(np.float64).sum(axis=0)
        qw, tw, ql, tl = o.lo_pose()
        out[pre + "lo_pose"] = np.concatenate([qw, tw, ql, tl])
        qm, tm, qmo, tmo = o.map_pose()
        out[pre + "map_pose"] = np.concatenate([qm, tm, qmo, tmo])
        info = o.map_info()
        out[pre + "map_totals"] = np.array([info["total_corner"], info["total_surf"]], dtype=np.int64)
        for outer in range(o.lo_num_outer()):
            c, p = o.lo_corr(outer)
            s = o.lo_solve(outer)
            out[pre + "lo%d_corner" % outer] = c
            out[pre + "lo%d_plane" % outer] = p
            out[pre + "lo%d_H0" % outer] = s["H0"]
            out[pre + "lo%d_g0" % outer] = s["g0"]
            out[pre + "lo%d_trace" % outer] = s["trace"]
            out[pre + "lo%d_resid0_head" % outer] = s["residuals0"][:64]
        for outer in range(o.map_num_outer()):
            s = o.map_solve(outer)
            out[pre + "map%d_counts" % outer] = np.array([s["corner_num"], s["surf_num"]], dtype=np.int32)
            out[pre + "map%d_H0" % outer] = s["H0"]
            out[pre + "map%d_trace" % outer] = s["trace"]
    return out


def run_vloam_case(n_rings, n_az, n_frames):
    """The coupled per-frame loop (MAIN/src/vloam_main_node.cpp:125-180) in combined mode: inputs (sweeps + pixel matches) and the
    oracle's per-frame VO estimate, VO -> LO prior, LO / mapping / VO world poses."""
    import orc_vloam
    synth = conftest.load_synth()
    seq = synth.SynthSequence(n_rings=n_rings, n_azimuth=n_az, n_sweeps=n_frames + 1)
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
    o = orc_vloam.VloamOracle(cam_T_velo, rect0_T_cam, P, base_T_cam0, velo_T_cam0, detach_VO_LO=False, scan_line=n_rings, with_mapping=True)
    out = {"cam_T_velo": cam_T_velo, "rect0_T_cam": rect0_T_cam, "P_rect0": P, "base_T_cam0": base_T_cam0, "velo_T_cam0": velo_T_cam0}
    for k in range(n_frames):
        cloud = seq.sweep(k)
        out["in_%d" % k] = cloud[:, :3].copy()
        m = synth.synth_matches(seq, k) if k > 0 else (np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32))
        out["prev_uv_%d" % k], out["curr_uv_%d" % k] = m
        assert o.process(cloud, m[0] if k else None, m[1] if k else None) == 0
        pre = "f%d_" % k
        if o.vo_result is not None:
            out[pre + "vo"] = np.concatenate([o.vo_result["angles"], o.vo_result["t"], [o.vo_result["counter32"], o.vo_result["counter22"]]])
        pq, pt = o.lo_prior()
        out[pre + "prior"] = np.concatenate([pq, pt])
        qw, tw, ql, tl = o.lidar.lo_pose()
        qm, tm = o.lidar.map_published_pose()
        vq, vt = o.vo_world_pose()
        out[pre + "poses"] = np.concatenate([qw, tw, qm, tm, vq, vt])
        for outer in range(o.lidar.lo_num_outer()):
            c, p = o.lidar.lo_corr(outer)
            out[pre + "lo%d_corner" % outer] = c
            out[pre + "lo%d_plane" % outer] = p
    return out


def run_image_case(w, h, n_images, seed):
    """Image front-end (optical-flow configuration): n_images views of one texture, corners of every image and the flow of image k's
    corners from image k - 1 into image k (visual_odometry.cpp:91-132)."""
    synth = conftest.load_synth()
    canvas = synth.synth_texture(w + 128, h + 128, seed)
    out = {}
    prev = None
    for k in range(n_images):
        c, s = np.cos(0.004 * k) * (1 + 0.002 * k), np.sin(0.004 * k) * (1 + 0.002 * k)
        B = np.array([[c, -s], [s, c]])
        img = synth.warp_image(canvas, w, h, B, np.array([64 - 3.1 * k, 64 + 1.2 * k]) - (B - np.eye(2)) @ np.array([w / 2, h / 2]))
        out["img_%d" % k] = img
        corners = orc.good_features(img)
        out["corners_%d" % k] = corners
        if prev is not None:
            tracked, status = orc.pyr_lk(prev, img, corners)
            out["tracked_%d" % k], out["status_%d" % k] = tracked, status
        prev = img
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "image_320x96_3frames.npz"), **run_image_case(320, 96, 3, seed=21))
    vl4 = run_vloam_case(64, 256, 5)
    np.savez_compressed(os.path.join(HERE, "vloam_64x256_5frames.npz"), **vl4)
    small = run_case(64, 256, 3, store_inputs=True)
    np.savez_compressed(os.path.join(HERE, "loam_64x256_3frames.npz"), **small)
    big = run_case(64, 2048, 3, store_inputs=False)  # inputs regenerated from the seeds
    np.savez_compressed(os.path.join(HERE, "loam_64x2048_3frames.npz"), **big)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
