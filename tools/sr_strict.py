"""Is scan registration now bit-exact INCLUDING intensity (relTime through atan2f)?  Scenes + fuzz clouds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import conftest
vl = conftest.load_pkg(); synth = conftest.load_synth()
import orc; orc.build()
import test_gpu_fuzz as fz
cases = []
for rings, az in [(64, 2048), (64, 512), (16, 1024), (32, 1024)]:
    seq = synth.SynthSequence(n_rings=rings, n_azimuth=az, n_sweeps=6)
    for k in (0, 3): cases.append(("scene %dx%d k%d" % (rings, az, k), rings, seq.sweep(k)))
for rings, az, seed in fz.CASES + [(64, 2040, 2000 + i) for i in range(12)]:
    cases.append(("fuzz %dx%d s%d" % (rings, az, seed), rings, fz.random_cloud(synth, rings, az, seed)))
tot = 0
for name, rings, c in cases:
    h = vl.Handle(0, scan_line=rings, debug=1, with_mapping=0, max_points=max(c.shape[0], 1024))
    h.reset_frame(); h.scan_registration(c)
    o = orc.Oracle(scan_line=rings, with_mapping=False); o.scan_registration(c)
    d, sc = h.sr_debug(), o.sr_scalars()
    res = []
    for which in range(5):
        dv, rf = h.features(which), o.cloud(which)
        ok_xyz = dv.shape == rf.shape and np.array_equal(dv[:, :3].view(np.uint32), rf[:, :3].view(np.uint32))
        ok_i = dv.shape == rf.shape and np.array_equal(dv[:, 3].view(np.uint32), rf[:, 3].view(np.uint32))
        nbad = int(np.count_nonzero(dv[:, 3].view(np.uint32) != rf[:, 3].view(np.uint32))) if dv.shape == rf.shape else -1
        res.append("%s%s(%d)" % ("x" if ok_xyz else "X!", "i" if ok_i else "I!", nbad))
    so = np.float32(d["startOri"]).view(np.uint32) == np.float32(sc["startOri"]).view(np.uint32) and np.float32(d["endOri"]).view(np.uint32) == np.float32(sc["endOri"]).view(np.uint32)
    print("%-28s n %6d  %s  start/endOri bits %s" % (name, c.shape[0], " ".join(res), so))
# whole pipeline: are the intensities of the later clouds (cornerLast, surfLast, stacks, map) the same bits as well?
from test_gpu_laser_mapping import lexsort_rows, oracle_map_points
seq = synth.SynthSequence(n_rings=64, n_azimuth=1024, n_sweeps=8)
h = vl.Handle(0, debug=1, with_mapping=1); o = orc.Oracle(with_mapping=True)
for k in range(6):
    c = seq.sweep(k)
    h.reset_frame(); h.scan_registration(c); h.laser_odometry(); h.laser_mapping(); o.process(c)
    r = []
    for which in (5, 6, 7, 8):
        dv, rf = h.features(which), o.cloud(which)
        r.append("%d:%s" % (which, dv.shape == rf.shape and np.array_equal(dv.view(np.uint32), rf.view(np.uint32))))
    for kind in (0, 1):
        cnt, pts = h.map_dump(kind); ref = oracle_map_points(o, kind)
        a, b = lexsort_rows(pts), lexsort_rows(ref)
        r.append("map%d xyz %s all4 %s" % (kind, np.array_equal(a[:, :3].view(np.uint32), b[:, :3].view(np.uint32)), a.shape == b.shape and np.array_equal(a[:, :4].view(np.uint32), b[:, :4].view(np.uint32))))
    print("pipeline sweep", k, " ".join(r))
