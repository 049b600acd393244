"""Replay of a coupled-frame case of tests/test_gpu_fuzz.py (test_coupled_frames_on_random_inputs) with the debug hooks on: per frame the VO
estimate, the VO -> LO prior, and for both outer rounds of the odometry the correspondence sets, the start point and the trust-region traces of
device and oracle — where do they part?

  python tools/fuzz_diag_coupled.py <columns> <seed> <detach 0|1> [frames]      (on the GPU box; e.g. 2040 625534 0)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
np.set_printoptions(linewidth=220, precision=12)
import conftest
vl = conftest.load_pkg(); synth = conftest.load_synth()
import orc; orc.build()
import orc_vloam
import test_gpu_fuzz as fz
from test_gpu_laser_odometry import qdist

n_az, seed, detach = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rng = np.random.default_rng(seed)
cam_T_velo, rect0_T_cam, P = fz._perturbed_calib(synth, rng)
base_T_cam0, velo_T_cam0 = synth.kitti_like_extrinsics()
velo_T_cam0 = np.linalg.inv(cam_T_velo.astype(np.float64))
base, fin, clouds, poses = fz.moving_clouds(synth, 64, n_az, seed, n, step=0.3, min_turn=1.0)
K, T = P[:, :3].astype(np.float64), cam_T_velo.astype(np.float64)


def pixels(k):
    R, t = poses[k]
    pc = (base[fin, :3].astype(np.float64) @ R.T + t) @ T[:3, :3].T + T[:3, 3]
    uv = pc @ K.T
    with np.errstate(divide="ignore", invalid="ignore"):
        return uv[:, :2] / uv[:, 2:3], pc[:, 2]


h = vl.Handle(0, detach_VO_LO=int(detach), with_mapping=1, debug=1, max_points=max(base.shape[0], 1024))
h.vo_set_calib(cam_T_velo, rect0_T_cam, P)
h.set_extrinsics(base_T_cam0, velo_T_cam0)
o = orc_vloam.VloamOracle(cam_T_velo, rect0_T_cam, P, base_T_cam0, velo_T_cam0, detach_VO_LO=detach, with_mapping=True)
for k in range(n):
    m = (None, None)
    if k > 0:
        (u0, z0), (u1, z1) = pixels(k - 1), pixels(k)
        ok = (z0 > 0.5) & (z1 > 0.5) & (u0[:, 0] >= 0) & (u0[:, 0] < 1242) & (u0[:, 1] >= 0) & (u0[:, 1] < 375) & (u1[:, 0] >= 0) & (u1[:, 0] < 1242) & (u1[:, 1] >= 0) & (u1[:, 1] < 375)
        idx = rng.choice(np.nonzero(ok)[0], size=min(900, int(ok.sum())), replace=False)
        pu = np.concatenate([u0[idx].astype(np.float32).astype(np.int32), np.stack([rng.integers(0, 1242, 300), rng.integers(0, 375, 300)], axis=1).astype(np.int32)])
        cu = np.concatenate([u1[idx].astype(np.float32).astype(np.int32), np.stack([rng.integers(0, 1242, 300), rng.integers(0, 375, 300)], axis=1).astype(np.int32)])
        m = (np.ascontiguousarray(pu), np.ascontiguousarray(cu))
    h.process_frame(clouds[k], m[0], m[1])
    assert o.process(clouds[k], m[0], m[1]) == 0
    h.sync()
    r = h.vo_result()
    if k > 0:
        v = o.vo_result
        oq, ot = o.lo_prior()
        print("frame %d: VO counters %s vs %s  d angles %.2e d t %.2e | prior dq %.2e dt %.2e" % (k, (r["counter32"], r["counter22"]), (v["counter32"], v["counter22"]),
              np.linalg.norm(r["angles"] - v["angles"]), np.linalg.norm(r["t"] - v["t"]), qdist(r["prior_q"], oq), np.linalg.norm(r["prior_t"] - ot)))
    tj = h.trajectory()[k]
    qw, tw, ql, tl = o.lidar.lo_pose()
    qm, tm = o.lidar.map_published_pose()
    print("frame %d: LO world dq %.2e dt %.2e | map dq %.2e dt %.2e" % (k, qdist(tj[0:4], qw), np.linalg.norm(tj[4:7] - tw), qdist(tj[7:11], qm), np.linalg.norm(tj[11:14] - tm)))
    if k > 0:
        for outer in range(2):
            d = h.lo_debug(outer)
            oc, op = o.lidar.lo_corr(outer)
            same = np.array_equal(d["corner"], oc) and np.array_equal(d["plane"], op)
            s = o.lidar.lo_solve(outer); rec = d["rec"]
            print("   LO outer %d: corr same %s  n %d/%d  x_in dq %.2e dt %.2e  trace %s vs %s  x_out dq %.2e dt %.2e  term %s vs %s" % (
                outer, same, rec["n_factors"], oc.shape[0] + op.shape[0], qdist(rec["x_in"][:4], s["q_in"]), np.linalg.norm(rec["x_in"][4:] - s["t_in"]),
                rec["trace"].shape, s["trace"].shape, qdist(rec["x_out"][:4], s["q_out"]), np.linalg.norm(rec["x_out"][4:] - s["t_out"]), rec["termination"], s["termination"]))
            if not same:
                dc = set(map(tuple, d["corner"])) ^ set(map(tuple, oc)); dp = set(map(tuple, d["plane"])) ^ set(map(tuple, op))
                print("      corner rows in one only:", sorted(dc)[:8], " plane rows in one only:", sorted(dp)[:8])
            if rec["trace"].shape != s["trace"].shape or not np.array_equal(rec["trace"][:, 6:8], s["trace"][:, 6:8]) or qdist(rec["x_out"][:4], s["q_out"]) > 1e-9:
                print("      dev trace\n", rec["trace"]); print("      orc trace\n", s["trace"])
                print("      dev x_in", rec["x_in"], "\n      orc x_in", s["q_in"], s["t_in"])
