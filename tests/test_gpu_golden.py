"""-m gpu: the HIP path against the committed golden fixtures (tests/golden/*.npz, written by make_golden.py from
the CPU oracle; the reference itself has no golden vectors).  Nothing here touches oracle/."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def qdist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def run(vl, g, clouds):
    h = vl.Handle(0, debug=1, with_mapping=1)
    for k, cloud in enumerate(clouds):
        h.reset_frame()
        h.scan_registration(cloud)
        qw, tw, ql, tl = h.laser_odometry()
        qm, tm = h.laser_mapping()
        pre = "f%d_" % k
        d = h.sr_debug()
        assert d["N2"] == int(g[pre + "N2"])
        assert np.array_equal(d["scanStartInd"], g[pre + "ring_start"]) and np.array_equal(d["scanEndInd"], g[pre + "ring_end"])
        for name in ("sharpInd", "lessSharpInd", "flatInd"):
            assert np.array_equal(d[name], g[pre + name]), (k, name)
        lf = h.features(4)
        assert lf.shape[0] == int(g[pre + "n_lessFlat"])
        assert np.allclose(lf[:, :3].astype(np.float64).sum(axis=0), g[pre + "lessFlat_xyz_sum"], rtol=0, atol=1e-9)
        gp = g[pre + "lo_pose"]
        assert qdist(qw, gp[0:4]) < 1e-8 and np.linalg.norm(tw - gp[4:7]) < 1e-8
        assert qdist(ql, gp[7:11]) < 1e-8 and np.linalg.norm(tl - gp[11:14]) < 1e-8
        gm = g[pre + "map_pose"]
        assert qdist(qm, gm[0:4]) < 1e-8 and np.linalg.norm(tm - gm[4:7]) < 1e-8
        if k > 0:
            for outer in range(2):
                lo = h.lo_debug(outer)
                assert np.array_equal(lo["corner"], g[pre + "lo%d_corner" % outer])
                assert np.array_equal(lo["plane"], g[pre + "lo%d_plane" % outer])
                assert np.allclose(lo["rec"]["H0"], g[pre + "lo%d_H0" % outer], rtol=1e-8, atol=1e-9)
                assert np.allclose(lo["rec"]["trace"][:, 0], g[pre + "lo%d_trace" % outer][:, 0], rtol=1e-8, atol=1e-12)
                md = h.map_debug(outer)
                assert np.array_equal([md["corner_idx"].size, md["surf_idx"].size], g[pre + "map%d_counts" % outer])
                assert np.allclose(md["rec"]["trace"][:, 0], g[pre + "map%d_trace" % outer][:, 0], rtol=1e-8, atol=1e-12)
        tot = [h.map_dump(0)[0].size, h.map_dump(1)[0].size]
        assert np.array_equal(tot, g[pre + "map_totals"])


def test_golden_small(vl):
    g = np.load(os.path.join(HERE, "golden", "loam_64x256_3frames.npz"))
    clouds = []
    for k in range(3):
        c = np.zeros((g["in_%d" % k].shape[0], 4), dtype=np.float32)
        c[:, :3] = g["in_%d" % k]
        clouds.append(c)
    run(vl, g, clouds)


def test_golden_full_size(vl, sweeps):
    g = np.load(os.path.join(HERE, "golden", "loam_64x2048_3frames.npz"))
    run(vl, g, [sweeps(64, 2048, k, n_sweeps=3) for k in range(3)])


def test_golden_coupled_vloam_frames(vl):
    """The coupled VO + LiDAR frame loop (vloam_process_frame, detach_VO_LO = 0) against tests/golden/vloam_64x256_5frames.npz."""
    g = np.load(os.path.join(HERE, "golden", "vloam_64x256_5frames.npz"))
    h = vl.Handle(0, debug=1, with_mapping=1, detach_VO_LO=0)
    h.vo_set_calib(g["cam_T_velo"], g["rect0_T_cam"], g["P_rect0"])
    h.set_extrinsics(g["base_T_cam0"], g["velo_T_cam0"])
    for k in range(5):
        c = np.zeros((g["in_%d" % k].shape[0], 4), dtype=np.float32)
        c[:, :3] = g["in_%d" % k]
        if k:
            h.process_frame(c, g["prev_uv_%d" % k], g["curr_uv_%d" % k])
        else:
            h.process_frame(c)
        pre = "f%d_" % k
        tol = 1e-6 if k == 1 else 1e-8   # frame 1: the VO starts from 2 acos(1 - ulp) (tests/test_oracle_vloam.py)
        r = h.vo_result()
        if k:
            gv = g[pre + "vo"]
            assert (r["counter32"], r["counter22"]) == (int(gv[6]), int(gv[7]))
            assert np.linalg.norm(r["angles"] - gv[0:3]) < tol and np.linalg.norm(r["t"] - gv[3:6]) < tol
            if k > 1:
                for outer in range(2):
                    lo = h.lo_debug(outer)
                    assert np.array_equal(lo["corner"], g[pre + "lo%d_corner" % outer]) and np.array_equal(lo["plane"], g[pre + "lo%d_plane" % outer])
                    assert np.array_equal(lo["rec"]["x_in"][:4], r["prior_q"]) and np.array_equal(lo["rec"]["x_in"][4:], r["prior_t"])
        gp = g[pre + "prior"]
        assert qdist(r["prior_q"], gp[0:4]) < tol and np.linalg.norm(r["prior_t"] - gp[4:7]) < tol
        tj, vj, gq = h.trajectory()[k], h.vo_trajectory()[k], g[pre + "poses"]
        tolp = 1e-6 if k >= 1 else 1e-8
        assert qdist(tj[0:4], gq[0:4]) < tolp and np.linalg.norm(tj[4:7] - gq[4:7]) < tolp
        assert qdist(tj[7:11], gq[7:11]) < tolp and np.linalg.norm(tj[11:14] - gq[11:14]) < tolp
        assert qdist(vj[0:4], gq[14:18]) < tolp and np.linalg.norm(vj[4:7] - gq[18:21]) < tolp


def test_golden_image_front_end(vl):
    """The HIP image front-end reproduces the committed corners / flow of tests/golden/image_320x96_3frames.npz bit for bit."""
    g = np.load(os.path.join(HERE, "golden", "image_320x96_3frames.npz"))
    h = vl.Handle(0, with_mapping=0, image_width=320, image_height=96)
    for k in range(3):
        h.vo_process_image(g["img_%d" % k])
        assert np.array_equal(h.vo_keypoints(), g["corners_%d" % k])
        a, b, st = h.vo_flow()
        if k == 0:
            assert a.shape[0] == 0
        else:
            assert np.array_equal(a, g["corners_%d" % k]) and np.array_equal(b, g["tracked_%d" % k]) and np.array_equal(st, g["status_%d" % k])
    h.close()
