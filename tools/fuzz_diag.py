"""Stage-wise replay of a whole-pipeline case of tests/test_gpu_fuzz.py: the first sweep / stage where device and oracle differ.

  python tools/fuzz_diag.py <rings> <columns> <seed> <sweeps> [vlp | rand]      (on the GPU box; e.g. 64 2040 10014 12)

Per sweep: odometry poses, correspondence sets and trust-region traces of both outer rounds, the clouds handed on (cornerLast, surfLast, the
two mapping stacks: all four floats), the map after the sweep, the mapping factor sets.  Clouds that differ are saved under
gpurun_out/fuzz_diag/.  This is how round 6 traced 1e-6 pose differences to ONE return on another scan line (OCML atanf vs glibc atanf)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "fuzz_diag")
os.makedirs(OUT, exist_ok=True)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import conftest
vl = conftest.load_pkg(); synth = conftest.load_synth()
import orc; orc.build()
import test_gpu_fuzz as fz
from test_gpu_laser_mapping import qdist
from test_gpu_laser_odometry import compare_outer

rings, n_az, seed, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = fz.VLP if (len(sys.argv) > 5 and sys.argv[5] == "vlp") else fz.KITTI
step, skip = 0.12, 1
if len(sys.argv) > 5 and sys.argv[5] == "rand":
    cfg = dict(fz._random_cfg(seed - 9000, rings)); step = cfg.pop("_step"); skip = cfg["mapping_skip_frame"]
    print("cfg", cfg, "step", step)
base = fz.random_cloud(synth, rings, n_az, seed, keep_lo=0.85)
fin = np.isfinite(base[:, :3]).all(axis=1)
rng = np.random.default_rng(seed + 1)
clouds = []
for k in range(n):
    ang, t = -0.004 * k, np.array([-step * k, 0.01 * k, 0.0])
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=np.float64)
    c = base.copy()
    p = base[fin, :3].astype(np.float64) @ R.T + t
    c[fin, :3] = (p * (1.0 + 0.0005 * rng.standard_normal((p.shape[0], 1)))).astype(np.float32)
    c[rng.random(c.shape[0]) < 0.01, :3] = np.nan
    clouds.append(c)
h = vl.Handle(0, scan_line=rings, with_mapping=1, debug=1, max_points=max(base.shape[0], 1024), **cfg)
o = orc.Oracle(scan_line=rings, with_mapping=True, minimum_range=cfg["minimum_range"], line_res=cfg["mapping_line_resolution"], plane_res=cfg["mapping_plane_resolution"], mapping_skip_frame=skip)
from test_gpu_laser_mapping import lexsort_rows, oracle_map_points
for k, c in enumerate(clouds):
    h.reset_frame(); h.scan_registration(c)
    qw, tw, ql, tl = h.laser_odometry()
    qm, tm = h.laser_mapping()
    assert o.process(c) == 0
    oqw, otw, oql, otl = o.lo_pose()
    print("sweep %d: LO f2f dq %.2e dt %.2e | world dq %.2e dt %.2e" % (k, qdist(ql, oql), np.linalg.norm(tl - otl), qdist(qw, oqw), np.linalg.norm(tw - otw)))
    if k > 0:
        for outer in range(2):
            d = h.lo_debug(outer)
            oc, op = o.lo_corr(outer)
            same = np.array_equal(d["corner"], oc) and np.array_equal(d["plane"], op)
            s = o.lo_solve(outer); rec = d["rec"]
            print("   LO outer %d: corr same %s  n %d/%d  trace %s vs %s  x_out dt %.2e" % (outer, same, rec["n_factors"], oc.shape[0] + op.shape[0], rec["trace"].shape, s["trace"].shape,
                  np.linalg.norm(rec["x_out"][4:] - s["t_out"])))
            if rec["trace"].shape == s["trace"].shape:
                print("      flags same %s  cost rel %.2e" % (np.array_equal(rec["trace"][:, 6:8], s["trace"][:, 6:8]), np.max(np.abs(rec["trace"][:, 0] - s["trace"][:, 0]) / (np.abs(s["trace"][:, 0]) + 1e-300))))
            else:
                print("      dev trace cost", rec["trace"][:, 0], "flags", rec["trace"][:, 6:8].T, "term", rec["termination"])
                print("      orc trace cost", s["trace"][:, 0], "flags", s["trace"][:, 6:8].T, "term", s["termination"])
                print("      dev trace full\n", rec["trace"]); print("      orc trace full\n", s["trace"])
    for which in (5, 6, 7, 8):
        dv, rf = h.features(which), o.cloud(which)
        same = dv.shape == rf.shape and np.array_equal(dv[:, :3].view(np.uint32), rf[:, :3].view(np.uint32))
        print("   cloud %d: %s vs %s same %s" % (which, dv.shape, rf.shape, same))
        if not same:
            m = min(dv.shape[0], rf.shape[0])
            neq = np.nonzero((dv[:m, :3].view(np.uint32) != rf[:m, :3].view(np.uint32)).any(axis=1))[0]
            i0 = int(neq[0]) if neq.size else m
            print("      first difference at row", i0, "of", m, " differing rows", neq.size)
            print("      dev rows", dv[max(i0 - 1, 0):i0 + 3]); print("      orc rows", rf[max(i0 - 1, 0):i0 + 3])
            np.save(os.path.join(OUT, "in_%d_%d.npy" % (which, k)), o.cloud(which - 2))
            np.save(os.path.join(OUT, "dev_%d_%d.npy" % (which, k)), dv); np.save(os.path.join(OUT, "orc_%d_%d.npy" % (which, k)), rf)
    oq, ot, oqm, otm = o.map_pose()
    print("   map pose dq %.2e dt %.2e" % (qdist(qm, oq), np.linalg.norm(tm - ot)))
    for kind in (0, 1):
        cnt, pts = h.map_dump(kind); ref = oracle_map_points(o, kind)
        a, b = lexsort_rows(pts), lexsort_rows(ref)
        same = a.shape == b.shape and np.array_equal(a[:, :4].view(np.uint32), b[:, :4].view(np.uint32))
        print("   map kind %d: %s vs %s same %s" % (kind, a.shape, b.shape, same))
        if not same and a.shape == b.shape:
            neq = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
            print("      differing rows", neq.size, "first", neq[:5]); print("      dev", a[neq[:4]]); print("      orc", b[neq[:4]])
        elif not same:
            sa = set(map(bytes, a.view(np.uint8).reshape(a.shape[0], -1))); sb = set(map(bytes, b.view(np.uint8).reshape(b.shape[0], -1)))
            od = [np.frombuffer(x, np.float32) for x in list(sa - sb)[:6]]; oo = [np.frombuffer(x, np.float32) for x in list(sb - sa)[:6]]
            print("      only dev", len(sa - sb), od); print("      only orc", len(sb - sa), oo)
    if o.map_num_outer() == 2:
        for outer in range(2):
            d = h.map_debug(outer)
            ci, cab, si, spl = o.map_factors(outer)
            s = o.map_solve(outer); rec = d["rec"]
            print("   map outer %d: corner set same %s (%d/%d) surf set same %s (%d/%d) trace %s vs %s x_in dt %.2e x_out dt %.2e" % (
                outer, np.array_equal(d["corner_idx"], ci), d["corner_idx"].size, ci.size, np.array_equal(d["surf_idx"], si), d["surf_idx"].size, si.size,
                rec["trace"].shape, s["trace"].shape, np.linalg.norm(rec["x_in"][4:] - s["t_in"]), np.linalg.norm(rec["x_out"][4:] - s["t_out"])))
            if not np.array_equal(d["corner_idx"], ci):
                print("      corner only dev", np.setdiff1d(d["corner_idx"], ci), "only orc", np.setdiff1d(ci, d["corner_idx"]))
            if not np.array_equal(d["surf_idx"], si):
                print("      surf only dev", np.setdiff1d(d["surf_idx"], si), "only orc", np.setdiff1d(si, d["surf_idx"]))
            if rec["trace"].shape != s["trace"].shape or not np.array_equal(rec["trace"][:, 6:8], s["trace"][:, 6:8]):
                print("      dev trace\n", rec["trace"]); print("      orc trace\n", s["trace"])
