"""Inputs that drive branches of the RESTATED THIRD-PARTY code (oracle/orc_pcl.cpp, oracle/orc_ceres.cpp) that ordinary sweeps never take —
shared by the CPU tests (tests/test_oracle_branches.py: the oracle really takes the branch, and the second Python transcriptions agree) and the
-m gpu tests (tests/test_gpu_branches.py: the HIP path against the oracle on the same input).  The table "branch -> oracle test -> GPU test" is in
DESIGN.md §2.

Two levers: whole sweeps (what the reference's callback would see), and — where a branch needs clouds no scene produces on demand — clouds
handed to the stages directly (LaserOdometry::input / LaserMapping::input deep-copy whatever they are given, laser_odometry.cpp:141-145,
laser_mapping.cpp:172-181: vloam_set_odometry_input / vloam_set_mapping_input, orc_stage_*)."""
import numpy as np


def ground_only_sequence(synth, n=6, shape=(64, 512)):
    """An open field: the sweeps see the ground plane and nothing else.  Every plane normal is (0, 0, 1) up to noise, no edge features worth the
    name: J^T J of both solves is (nearly) rank 3 — x, y and yaw are unobservable —, the columns' LM diagonal falls under min_lm_diagonal (1e-6:
    the clamp of LevenbergMarquardtStrategy::ComputeStep) and the damped 6 x 6 system is solved with pivots of ~1e-10."""
    seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=n + 1, speed=2.0)
    seq.boxes = np.zeros((0, 6))
    seq.cyls = np.zeros((0, 5))
    return [np.ascontiguousarray(seq.sweep(k), dtype=np.float32) for k in range(n)]


def repeated_sweep_sequence(synth, n=5, shape=(64, 512), repeat_at=(2, 3)):
    """A sensor that stands still and a source that replays a sweep (a paused bag): sweep k == sweep k - 1 bit for bit.  Scan-to-scan every
    feature finds ITSELF: all residuals are exactly 0, the gradient is 0, and Ceres stops in FinalizeIterationAndCheckIfMinimizerCanContinue at
    iteration 0 on gradient_tolerance — before any trust-region step."""
    seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=n + 1)
    out = []
    for k in range(n):
        out.append(out[-1].copy() if k in repeat_at else np.ascontiguousarray(seq.sweep(k), dtype=np.float32))
    return out


def lattice(nx, ny, nz, step, origin, ring=0.0):
    """Points on a regular lattice, float32-exact coordinates (step and origin are multiples of 2^-4), x fastest."""
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pts = np.zeros((g.shape[0], 4), np.float32)
    pts[:, :3] = g * np.float32(step) + np.asarray(origin, np.float32)
    pts[:, 3] = np.float32(ring)
    return pts


def tie_clouds():
    """(seed_corner, seed_surf, query_corner, query_surf) for an identity pose; every coordinate a multiple of 2^-7 (exact in f32), one seed point per
    map voxel (so the map holds the seed points themselves), the structures straddle the cube face at x = 25 m.

    corner: three "poles lying down" — two rows of ten points 1/64 m apart in y (either side of a 0.4 m voxel face), 7/16 m apart in x.  A query
    on the axis between the rows at a node's x has 2 neighbours at one distance and then FOUR at the next: the 5-NN takes three of those four.
    surf: a 10 x 9 lattice (7/8 m) on a plane whose columns alternate +-1/32 m in z.  A query half-way between two nodes of a row has 2
    neighbours at one distance and then FOUR at the next (d^2 = 0.958 < 1 m^2, the gate of laser_mapping.cpp:547): again three of four.
    Which three changes the fitted line / plane, so the tie rule (lowest index of the gathered map cloud) is visible in the factors."""
    sc, qc = [], []
    for line, (Y, Z) in enumerate(((0.0, -0.5), (2.0, 0.25), (-2.0, 1.0))):
        for i in range(10):
            x = 23.0 + 0.4375 * i
            sc.append([x, Y + 0.390625, Z, 0.0])
            sc.append([x, Y + 0.40625, Z, 0.0])
            if 2 <= i <= 7:
                qc.append([x, Y + 0.3984375, Z + 0.0078125 * (line + 1), 0.0])
    ss, qs = [], []
    for i in range(10):
        for j in range(9):
            ss.append([20.125 + 0.875 * i, -3.5 + 0.875 * j, -1.5 + 0.03125 * (1 if i % 2 == 0 else -1), 0.0])
            if i < 9 and 1 <= j <= 7:
                qs.append([20.125 + 0.875 * i + 0.4375, -3.5 + 0.875 * j, -1.5, 0.0])
    return tuple(np.array(a, np.float32) for a in (sc, ss, qc, qs))


def six_way_walk(step=55.0):
    """Offsets (metres, one per sweep) added to the odometry translation handed to LaserMapping::input so that the 21 x 21 x 11 cube window
    (50 m cubes, centre index kept inside [3, size - 3), laser_mapping.cpp:218-402) rolls along every axis in BOTH directions: out to -440 m in
    x, +165 m / -165 m in z, -440 m / +440 m in y, then back across the origin to +440 m in x.  Consecutive sweeps stay within sensor range of
    each other, so the valid 5 x 5 x 3 block around a new position holds earlier sweeps' points and the scan-to-map gate opens along the way."""
    way = [(0, 0, 0), (-440, 0, 0), (-440, 0, 165), (-440, 0, -165), (-440, -440, -165), (-440, 440, 0), (440, 440, 0)]
    out = [np.zeros(3)]
    for a, b in zip(way[:-1], way[1:]):
        a, b = np.array(a, float), np.array(b, float)
        n = int(np.ceil(np.linalg.norm(b - a) / step))
        out += [a + (b - a) * (i / n) for i in range(1, n + 1)]
    return out
