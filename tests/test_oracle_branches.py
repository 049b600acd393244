"""CPU: branches of the RESTATED third-party code that ordinary sweeps never take (tests/branch_cases.py) — the oracle really takes them, and the
independent second transcriptions (numpy VoxelGrid / brute-force kNN, tests/ceres_transcription.py) agree on the same inputs.  The same inputs
run on the device in tests/test_gpu_branches.py; DESIGN.md §2 holds the table branch -> oracle test -> GPU test, with the branches that inputs of
the reference's own pipeline cannot reach named as such."""
import numpy as np

import branch_cases as B


def numpy_voxel_grid(pts, leaf):
    """pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.10) in numpy f32, written from the published source (not from oracle/orc_pcl.cpp)."""
    pts = np.asarray(pts, np.float32)
    if pts.shape[0] == 0:
        return pts.reshape(0, 4)
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = pts[:, :3].min(axis=0), pts[:, :3].max(axis=0)
    d = ((mx - mn) * inv).astype(np.int64) + 1
    if int(d[0]) * int(d[1]) * int(d[2]) > np.iinfo(np.int32).max:
        return pts.copy()
    min_b = np.floor(mn * inv).astype(np.int32)
    max_b = np.floor(mx * inv).astype(np.int32)
    div = max_b - min_b + 1
    ijk = (np.floor(pts[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int32)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    start = 0
    while start < order.size:
        end = start
        while end < order.size and idx[order[end]] == idx[order[start]]:
            end += 1
        s = np.zeros(4, np.float32)
        for i in order[start:end]:
            s = s + pts[i]          # f32 accumulation in sorted order
        out.append(s / np.float32(end - start))
        start = end
    return np.array(out, np.float32)


def test_voxel_grid_empty_single_cell_and_overflow(orc):
    """VoxelGrid's three exits besides the ordinary one: empty input -> empty output; every point in ONE cell -> one centroid (f32 sums in input
    order: 500 points, the sum loses bits a pairwise sum would keep); more than INT_MAX cells in the bounding box -> output = input."""
    assert orc.voxel_grid(np.zeros((0, 4), np.float32), 0.4).shape == (0, 4)
    rng = np.random.default_rng(5)
    one = np.zeros((500, 4), np.float32)
    one[:, :3] = (rng.uniform(0.0, 0.39, (500, 3)) + np.array([12.0, -3.2, 0.8])).astype(np.float32)
    one[:, 3] = rng.uniform(0, 50, 500).astype(np.float32)
    got = orc.voxel_grid(one, 0.4)
    assert got.shape == (1, 4) and np.array_equal(got.view(np.uint32), numpy_voxel_grid(one, 0.4).view(np.uint32))
    pairwise = one.astype(np.float64).mean(axis=0).astype(np.float32)
    assert not np.array_equal(got[0], pairwise), "the case should distinguish sequential f32 accumulation from an exact mean"
    far = one.copy()
    far[0, :3] = [-3000.0, -3000.0, -300.0]
    far[1, :3] = [3000.0, 3000.0, 300.0]      # 15 000 x 15 000 x 1 500 cells of 0.4 m > INT_MAX
    got = orc.voxel_grid(far, 0.4)
    assert np.array_equal(got.view(np.uint32), far.view(np.uint32)) and np.array_equal(numpy_voxel_grid(far, 0.4).view(np.uint32), far.view(np.uint32))


def test_knn_fewer_points_than_k_and_exact_ties_across_leaves(orc):
    """nearestKSearch with fewer points than k returns what there is (laser_mapping.cpp:477,545 read pointSearchSqDis[4]: the > 10 && > 50 gate of
    :448 is what keeps that in bounds); exact distance ties — tests/branch_cases.py::tie_clouds: two neighbours at one f32 distance, then four at the next, of which three are taken; the
    structures spread over several kd-tree leaves — resolve to the LOWEST INDEX in both the tree and the brute-force search (the oracle's canonical
    rule for FLANN's open tie order), equal to a stable numpy argsort of (d2, index)."""
    few = B.lattice(3, 1, 1, 1.0, (0.0, 0.0, 0.0))
    q = np.array([[0.4, 0.0, 0.0]], np.float32)
    for tree in (True, False):
        L = orc.lib()
        idx = np.full((1, 5), -7, np.int32)
        d2 = np.full((1, 5), -7.0, np.float32)
        import ctypes as C
        L.orc_knn(few.ctypes.data_as(C.c_void_p), 3, q.ctypes.data_as(C.c_void_p), 1, 5, idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), int(tree))
        assert list(idx[0, :3]) == [0, 1, 2] and np.allclose(d2[0, :3], [0.16, 0.36, 2.56])
    seed_c, seed_s, qc, qs = B.tie_clouds()
    for pts, qq in ((seed_c, qc), (seed_s, qs)):
        assert pts.shape[0] > 48   # (leaf size of the restated kd-tree: 16 points — several leaves either way)
        it, dt = orc.knn(pts, qq[:, :3], 5, use_tree=True)
        ib, db = orc.knn(pts, qq[:, :3], 5, use_tree=False)
        assert np.array_equal(it, ib) and np.array_equal(dt, db)
        d = ((pts[None, :, :3] - qq[:, None, :3]) ** 2).astype(np.float32)
        d2 = (d[:, :, 0] + d[:, :, 1]) + d[:, :, 2]     # flann::L2_Simple: the f32 sum in coordinate order
        want = np.argsort(d2, axis=1, kind="stable")[:, :5]
        assert np.array_equal(it, want)
        sd = np.sort(d2, axis=1)
        assert np.all(sd[:, 0] == sd[:, 1]) and np.all(sd[:, 2] == sd[:, 5]) and np.all(sd[:, 6] > sd[:, 5]), "2 neighbours at one distance, then a FOUR-way tie of which the 5-NN takes three"


def plane_only_problem():
    """96 LidarPlaneFactors whose three map points lie in the plane z = -1.75: every normal is (0, 0, +-1) exactly."""
    rows = []
    for r in range(8):
        for i in range(12):
            x, y = 6.0 + 0.5 * i, -4.0 + 0.75 * r
            c = [x + 0.125, y, -1.75 + 0.0625]
            rows.append([1, *c, x, y, -1.75, x + 0.5, y, -1.75, x, y + 0.75, -1.75])
    return rows


def test_rank_deficient_normal_equations_and_gradient_tolerance_at_iteration_zero(orc):
    """(1) A Jacobian with three EXACTLY zero columns (plane factors with one common normal: x, y, yaw unobservable): the LM diagonal of those
    columns is clamped to min_lm_diagonal = 1e-6 (LevenbergMarquardtStrategy::ComputeStep), the damped system stays solvable, the step moves
    only z / roll / pitch.  (2) Started AT the minimiser every residual is 0: gradient_max_norm <= gradient_tolerance stops the minimiser in
    FinalizeIterationAndCheckIfMinimizerCanContinue at iteration 0, before any step.  C++ restatement and Python transcription agree row by row."""
    import ceres_transcription as ct
    rows = plane_only_problem()
    a = orc.solve(rows, [0, 0, 0, 1], [0, 0, 0], quaternion=True, huber_a=0.1, max_iters=4)
    b = ct.solve(rows, [0, 0, 0, 1], [0, 0, 0], quaternion=True, huber_a=0.1, max_iters=4)
    assert np.array_equal(np.diag(a["H0"])[2:5], [0.0, 0.0, 0.0]) and np.all(np.diag(a["H0"])[[0, 1, 5]] > 1.0)
    assert a["trace"].shape == b["trace"].shape and np.array_equal(a["trace"][:, 6:8], b["trace"][:, 6:8]) and a["termination"] == b["termination"]
    assert np.allclose(a["trace"][:, [0, 5]], b["trace"][:, [0, 5]], rtol=1e-9, atol=1e-20)
    assert abs(a["p1"][2] + 0.0625) < 1e-9 and np.max(np.abs(a["p1"][:2])) < 1e-12 and abs(a["p0"][2]) < 1e-12   # z found; x, y, yaw untouched
    assert np.allclose(np.concatenate([a["p0"], a["p1"]]), np.concatenate([b["p0"], b["p1"]]), rtol=0, atol=1e-12)
    at_min = orc.solve(rows, [0, 0, 0, 1], [0.0, 0.0, -0.0625], quaternion=True, huber_a=0.1, max_iters=4)
    bt_min = ct.solve(rows, [0, 0, 0, 1], [0.0, 0.0, -0.0625], quaternion=True, huber_a=0.1, max_iters=4)
    assert at_min["trace"].shape[0] == 1 and at_min["termination"] == 1 and at_min["trace"][0, 2] == 0.0 and at_min["initial_cost"] == 0.0
    assert bt_min["trace"].shape[0] == 1 and bt_min["message"] == "Gradient tolerance reached."


def test_pipeline_cases_take_their_branches(orc, synth):
    """The whole-pipeline forms of the cases above, as tests/test_gpu_branches.py feeds them to the device."""
    import test_gpu_branches as G
    o = orc.Oracle(with_mapping=True)
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=5)
    at_b = []
    G.run_plane_only(o.stage_sr, lambda w, c: o.set_sr_cloud(w, c), o.stage_lo, lambda: None, seq, lambda: at_b.extend(o.lo_solve(r) for r in range(2)))
    for s in at_b:
        assert s["trace"].shape[0] == 1 and s["termination"] == 1 and s["trace"][0, 2] == 0.0 and s["initial_cost"] == 0.0, "sweep B: all residuals 0"
    s0 = o.lo_solve(0)
    assert [c.shape[0] for c in o.lo_corr(0)] == [0, 96]
    assert np.array_equal(np.diag(s0["H0"])[2:5], [0.0, 0.0, 0.0]) and s0["trace"].shape[0] >= 3
    assert abs(o.lo_pose()[3][2] + 0.0625) < 1e-9
    # ties in the map + an empty corner cloud + a single-cell surf cloud handed to LaserMapping::input
    o = orc.Oracle(with_mapping=True)
    G.run_map_ties(o, seq, oracle=True)
    assert o.map_num_outer() == 2
    ci, cab, si, spl = o.map_factors(0)
    assert ci.size + si.size > 50, "the tied neighbourhoods must still produce factors"
    G.run_map_empty_and_single_cell(o, seq, oracle=True)
    assert o.cloud(7).shape[0] == 0 and o.cloud(8).shape[0] == 1
