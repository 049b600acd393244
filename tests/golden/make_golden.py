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


if __name__ == "__main__":
    small = run_case(64, 256, 3, store_inputs=True)
    np.savez_compressed(os.path.join(HERE, "loam_64x256_3frames.npz"), **small)
    big = run_case(64, 2048, 3, store_inputs=False)  # inputs regenerated from the seeds
    np.savez_compressed(os.path.join(HERE, "loam_64x2048_3frames.npz"), **big)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
