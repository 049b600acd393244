"""Inputs that drive the reference's degenerate-frame branches (shared by the CPU and the -m gpu tests).

    laser_odometry.cpp:272,359      d^2 < DISTANCE_SQ_THRESHOLD (25) rejects EVERY correspondence -> a Ceres problem without residual blocks
    laser_odometry.cpp:452-455      fewer than 10 correspondences: "less correspondence" is printed and the solve PROCEEDS
    laser_mapping.cpp:448,631-635   the map holds <= 10 corner or <= 50 surf points for several frames AFTER frame 0: no optimisation,
                                    transformUpdate and the map insert still run
    visual_odometry.cpp:309-314     every match beyond remove_VO_outlier; :345,393 every depth0 <= 0 (CostFunctor22 only); no match at all

Every builder returns plain numpy inputs; what they trigger is asserted on the oracle (tests/test_oracle_degenerate.py) and on the
device against the oracle (tests/test_gpu_degenerate.py)."""
import numpy as np

FAR = 20.0   # every return of a synthetic sweep lies within 5 .. 80 m: scaled by 20 nothing comes within 20 m of the sweep before


def far_sweep(cloud, scale=FAR):
    """The same sweep seen `scale` times farther away: ring ids and azimuths (angles) unchanged, no point within 5 m of any point of a
    normal sweep -> zero laser-odometry correspondences against the sweep before AND for the sweep after."""
    out = np.array(cloud, dtype=np.float32, copy=True)
    out[:, :3] *= np.float32(scale)
    return out


def wedge_sweep(cloud, n_rings, n_azimuth, rings=(20, 21), col0=100, width=24):
    """far_sweep except a small azimuth wedge of a few scan lines, which stays where it was: the flat picks of that wedge find their
    three neighbours in the previous sweep -> a handful (< 10) of correspondences."""
    near = np.asarray(cloud, dtype=np.float32).reshape(n_rings, n_azimuth, 4)
    out = far_sweep(cloud).reshape(n_rings, n_azimuth, 4)
    for r in rings:
        out[r, col0:col0 + width] = near[r, col0:col0 + width]
    return out.reshape(-1, 4)


def lo_sequence(synth, n=7, shape=(64, 512), far_at=(3,), wedge_at=(), **seeds):
    """n sweeps of a synthetic drive with the sweeps in far_at / wedge_at replaced by their degenerate variants."""
    seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=n + 1, **seeds)
    out = []
    for k in range(n):
        c = seq.sweep(k)
        if k in far_at:
            c = far_sweep(c)
        elif k in wedge_at:
            c = wedge_sweep(c, shape[0], shape[1])
        out.append(np.ascontiguousarray(c, dtype=np.float32))
    return out


def sparse_map_sequence(synth, n=7):
    """A 16-line sensor with 128 columns: the surf map stays at <= 50 points for the first three frames (0, 38, 50 points at gather time)
    and crosses the gate at frame 3 (59) — laser_mapping.cpp:448 with a NON-empty map."""
    seq = synth.SynthSequence(n_rings=16, n_azimuth=128, n_sweeps=14, seed_scene=1234)   # (the drive's shape depends on its length: fixed)
    return [np.ascontiguousarray(seq.sweep(k), dtype=np.float32) for k in range(n)]


def vo_cases(synth, seq, k):
    """(name, prev_uv, curr_uv) for frame k of `seq`: all matches without LiDAR depth (image rows above every projected return),
    no match at all, every match displaced by more than remove_VO_outlier = 100 px."""
    pu, cu = synth.synth_matches(seq, k)
    top_p, top_c = pu.copy(), cu.copy()
    top_p[:, 1] = top_p[:, 1] % 40
    top_c[:, 1] = top_p[:, 1] + (cu[:, 1] - pu[:, 1])
    empty = np.zeros((0, 2), dtype=np.int32)
    return [("no_depth", top_p, top_c), ("no_match", empty, empty), ("all_outliers", pu, cu + 300)]
