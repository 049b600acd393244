"""-m gpu: the branch cases of tests/branch_cases.py on the device against the oracle (table branch -> oracle test -> GPU test: DESIGN.md §2).

Sweeps where a scene produces the branch; clouds handed to the stages directly where it takes clouds no scene produces on demand
(vloam_set_odometry_input / vloam_set_mapping_input — the reference's stages copy whatever they are handed)."""
import numpy as np

import branch_cases as B

POSE_TOL = 1e-8
EMPTY = np.zeros((0, 4), np.float32)


def qdist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


class OracleStages:
    def __init__(self, o):
        self.o = o

    def sr(self, cloud):
        assert self.o.stage_sr(cloud) == 0

    def set_sr(self, clouds5):
        for w, c in enumerate(clouds5):
            if c is not None:
                self.o.set_sr_cloud(w, c)

    def lo(self):
        self.o.stage_lo()

    def map(self, **kw):
        assert self.o.stage_map(**kw) == 0


class DeviceStages:
    def __init__(self, h):
        self.h = h

    def sr(self, cloud):
        self.h.reset_frame()
        self.h.scan_registration(cloud)

    def set_sr(self, clouds5):
        self.h.set_odometry_input(*clouds5)

    def lo(self):
        return self.h.laser_odometry()

    def map(self, corner=None, surf=None, full=None, q=None, t=None):
        if any(v is not None for v in (corner, surf, full, q, t)):
            self.h.set_mapping_input(corner, surf, full, q, t)
        return self.h.laser_mapping()


def plane_cloud(z, dx=0.0):
    """A patch of the plane z = const as eight scan lines of twelve points (intensity = scan line id)."""
    pts = []
    for r in range(10, 18):
        for i in range(12):
            pts.append([6.0 + 0.5 * i + dx, -4.0 + 0.75 * (r - 10), z, float(r)])
    return np.array(pts, np.float32)


def run_plane_only(sr, set_cloud, lo, mapping, seq, after_b=None):
    """Three sweeps whose odometry clouds are replaced by patches of ONE plane (LidarPlaneFactors with the common normal (0, 0, 1)).
    B lies exactly in A's plane and the first solve of a sequence starts from the exact identity: every residual is 0, the gradient is 0, and
    the minimiser stops at iteration 0 on gradient_tolerance (after_b() is called there).  C lies 1/16 m above: three exactly zero Jacobian
    columns (x, y, yaw) -> min_lm_diagonal clamp, rank-3 normal equations, a step in z / roll / pitch only."""
    a, b, c = plane_cloud(-1.75), plane_cloud(-1.75, dx=0.125), plane_cloud(-1.75 + 0.0625, dx=0.25)
    for k, pc in enumerate((a, b, c)):
        sr(seq.sweep(k))
        for w, cl in ((1, EMPTY), (2, EMPTY), (3, pc), (4, pc)):
            set_cloud(w, cl)
        lo()
        if k == 1 and after_b:
            after_b()
        mapping()
    return a, b, c


def map_seed_and_queries():
    return B.tie_clouds()


def run_map_ties(x, seq, oracle=False):
    """Sweep 0 seeds the map with one point per voxel on a lattice (identity pose); sweep 1's scan features sit in the middle of lattice cells:
    eight map points at ONE f32 distance per query, across the cube face at x = 25 m."""
    st = OracleStages(x) if oracle else DeviceStages(x)
    seed_c, seed_s, qc, qs = map_seed_and_queries()
    ident_q, ident_t = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
    st.sr(seq.sweep(0)); st.lo(); st.map(corner=seed_c, surf=seed_s, q=ident_q, t=ident_t)
    st.sr(seq.sweep(1)); st.lo()
    return st.map(corner=qc, surf=qs, q=ident_q, t=ident_t)


def run_map_empty_and_single_cell(x, seq, oracle=False):
    """Next sweep: LaserMapping::input gets an EMPTY corner cloud (VoxelGrid of nothing) and a surf cloud whose 40 points share one 0.8 m cell."""
    st = OracleStages(x) if oracle else DeviceStages(x)
    rng = np.random.default_rng(3)
    cell = np.zeros((40, 4), np.float32)
    cell[:, :3] = (np.array([24.05, 0.05, -1.55]) + rng.uniform(0, 0.7, (40, 3))).astype(np.float32)
    st.sr(seq.sweep(2)); st.lo()
    return st.map(corner=EMPTY, surf=cell, q=np.array([0.0, 0.0, 0.0, 1.0]), t=np.zeros(3))


# ------------------------------------------------------------------------------------------------------------------ the GPU tests
import pytest  # noqa: E402

gpu = pytest.mark.gpu


@gpu
def test_rank_deficient_solve_and_gradient_tolerance_on_the_device(vl, orc, synth):
    from test_gpu_laser_odometry import compare_outer
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=5)
    h = vl.Handle(0, with_mapping=1, debug=1)
    o = orc.Oracle(with_mapping=True)
    d, s = DeviceStages(h), OracleStages(o)
    at_b = {}

    def dev_after_b():
        at_b["dev"] = [h.lo_debug(r)["rec"] for r in range(2)]

    def orc_after_b():
        at_b["orc"] = [o.lo_solve(r) for r in range(2)]

    run_plane_only(d.sr, lambda w, c: h.set_odometry_input(*[c if i == w else None for i in range(5)]), d.lo, lambda: d.map(), seq, dev_after_b)
    run_plane_only(s.sr, lambda w, c: o.set_sr_cloud(w, c), s.lo, lambda: s.map(), seq, orc_after_b)
    for r in range(2):   # sweep B: gradient tolerance at iteration 0, on both sides
        assert at_b["orc"][r]["trace"].shape[0] == 1 and at_b["orc"][r]["termination"] == 1 and at_b["orc"][r]["trace"][0, 2] <= 1e-10
        assert at_b["dev"][r]["trace"].shape[0] == 1 and at_b["dev"][r]["termination"] == 1 and at_b["dev"][r]["trace"][0, 2] <= 1e-10
        assert at_b["dev"][r]["n_factors"] == 96 and at_b["dev"][r]["initial_cost"] == 0.0
    compare_outer(h.lo_debug(0), o, 0)     # sweep C, round 0: correspondences, residuals, J^T J / J^T r, the whole iteration table, termination, pose
    # round 1 starts at round 0's answer (cost ~1e-21): J^T J's unobservable diagonal is 1e-50 there, so compare_outer's entry-wise RELATIVE check
    # of J^T J has nothing to hold on to; the same quantities against the matrix's own size instead
    d1, s1 = h.lo_debug(1), o.lo_solve(1)
    oc, op = o.lo_corr(1)
    assert np.array_equal(d1["corner"], oc) and np.array_equal(d1["plane"], op)
    assert np.max(np.abs(d1["rec"]["H0"] - s1["H0"])) < 1e-9 * np.max(np.abs(s1["H0"])) and np.max(np.abs(d1["rec"]["g0"] - s1["g0"])) < 1e-9
    assert d1["rec"]["trace"].shape == s1["trace"].shape and d1["rec"]["termination"] == s1["termination"]
    assert qdist(d1["rec"]["x_out"][:4], s1["q_out"]) < POSE_TOL and np.linalg.norm(d1["rec"]["x_out"][4:] - s1["t_out"]) < POSE_TOL
    r0 = h.lo_debug(0)["rec"]
    assert np.array_equal(np.diag(r0["H0"])[2:5], [0.0, 0.0, 0.0]), "three exactly zero columns on the device too"
    assert r0["trace"].shape[0] >= 3
    qw, tw, ql, tl = o.lo_pose()
    dq, dt = h.odometry_pose()
    assert qdist(dq, qw) < POSE_TOL and np.linalg.norm(dt - tw) < POSE_TOL and abs(tl[2] + 0.0625) < 1e-9
    h.close()


@gpu
def test_exact_knn_ties_empty_and_single_cell_clouds_in_the_map(vl, orc, synth):
    from test_gpu_laser_mapping import compare_map_round, lexsort_rows, oracle_map_points
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=5)
    h = vl.Handle(0, with_mapping=1, debug=1)
    o = orc.Oracle(with_mapping=True)
    qm, tm = run_map_ties(h, seq)
    run_map_ties(o, seq, oracle=True)
    assert o.map_num_outer() == 2
    for which in (7, 8):
        assert np.array_equal(h.features(which)[:, :4].view(np.uint32), o.cloud(which)[:, :4].view(np.uint32))
    for outer in range(2):
        compare_map_round(h, o, outer)     # factor sets, the lines / planes fitted through the five TIED-BROKEN neighbours, traces, poses
    oq, ot, _, _ = o.map_pose()
    assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL
    qm, tm = run_map_empty_and_single_cell(h, seq)
    run_map_empty_and_single_cell(o, seq, oracle=True)
    assert h.features(7).shape[0] == 0 and h.features(8).shape[0] == 1
    assert np.array_equal(h.features(8)[:, :4].view(np.uint32), o.cloud(8)[:, :4].view(np.uint32)), "the single cell's f32 centroid, input order"
    oq, ot, _, _ = o.map_pose()
    assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL
    for kind in (0, 1):
        _, pts = h.map_dump(kind)
        ref = oracle_map_points(o, kind)
        assert pts.shape == ref.shape and np.array_equal(lexsort_rows(pts)[:, :4].view(np.uint32), lexsort_rows(ref)[:, :4].view(np.uint32))
    h.sync()
    h.close()


@gpu
def test_open_field_sweeps(vl, orc, synth):
    """Ground-only sweeps (no walls, no poles): nothing constrains x / y / yaw but noise features — J^T J spans four orders of magnitude more than
    on a street.  Normal equations + Cholesky (device) against DENSE_QR (oracle): every pose of every sweep."""
    clouds = B.ground_only_sequence(synth)
    h = vl.Handle(0, with_mapping=1)
    o = orc.Oracle(with_mapping=True)
    for k, c in enumerate(clouds):
        h.reset_frame(); h.scan_registration(c)
        qw, tw, _, _ = h.laser_odometry()
        qm, tm = h.laser_mapping()
        assert o.process(c) == 0
        oq, ot, _, _ = o.lo_pose()
        assert qdist(qw, oq) < POSE_TOL and np.linalg.norm(tw - ot) < POSE_TOL, "odometry pose, sweep %d" % k
        oq, ot = o.map_published_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "map pose, sweep %d" % k
    if True:
        s = o.lo_solve(1)
        ev = np.linalg.eigvalsh(s["H0"])
        assert ev[-1] / ev[0] > 1e3
    h.close()
