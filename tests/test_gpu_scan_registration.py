"""-m gpu: scanRegistration through the C ABI vs the CPU oracle — bit for bit, ALL FOUR floats of every point.

Reference: src/lidar_odometry_mapping/src/scan_registration.cpp:131-449.  Ring ids, compaction order, curvature, per-sector sort order,
picks, labels, voxel centroids AND intensity (= scan line + 0.1 relTime, relTime through atan2f and the reference's +-pi unwrap tests,
:234-265) must be identical.  Until round 6 the intensity's fraction carried a tolerance: the device called OCML's atan2f / atanf, the
reference's platform glibc's (1 - 2 ulp apart); the device now computes both the way glibc does (csrc/fdlibm_f32.h, tests/test_fdlibm_f32.py),
startOri / endOri included.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

def run_both(vl, orc, cloud, scan_line=64, minimum_range=5.0):
    h = vl.Handle(0, scan_line=scan_line, minimum_range=minimum_range, debug=1, with_mapping=0)
    h.reset_frame()
    h.scan_registration(cloud)
    o = orc.Oracle(scan_line=scan_line, minimum_range=minimum_range, with_mapping=False)
    assert o.scan_registration(cloud) == 0
    return h, o


def check_cloud(dev, ref, what, ori_bounds=None, max_flips=0):
    """x, y, z and intensity bit-exact.  (ori_bounds / max_flips: the tolerance arguments of rounds 1 - 5, kept so that callers read the
    same; no flip of relTime at an unwrap boundary is accepted any more.)  Returns 0."""
    assert dev.shape == ref.shape, "%s: %s vs %s" % (what, dev.shape, ref.shape)
    assert np.array_equal(dev[:, :3].view(np.uint32), ref[:, :3].view(np.uint32)), "%s xyz not bit-identical" % what
    assert np.array_equal(dev[:, 3].astype(np.int32), ref[:, 3].astype(np.int32)), "%s ring id differs" % what
    same = dev[:, 3].view(np.uint32) == ref[:, 3].view(np.uint32)
    assert np.all(same), "%s: intensity differs in %d of %d points (max %.3g)" % (what, np.count_nonzero(~same), same.size, np.max(np.abs(dev[:, 3] - ref[:, 3])))
    return 0


def unwrap_bounds(startOri, endOri):
    s, e = float(startOri), float(endOri)
    return [s - np.pi / 2, s + 1.5 * np.pi, s + np.pi, e - 1.5 * np.pi, e + np.pi / 2]


@pytest.mark.parametrize("shape,k", [((64, 512), 0), ((64, 512), 3), ((64, 2048), 0), ((64, 2048), 5), ((16, 1024), 1), ((32, 1024), 2),
                                     ((64, 4000), 1)])  # 4000 columns: sectors longer than 384 points (the wide register path of the picks)
def test_scan_registration_parity(vl, orc, sweeps, shape, k):
    cloud = sweeps(shape[0], shape[1], k)
    h, o = run_both(vl, orc, cloud, scan_line=shape[0])
    d = h.sr_debug()
    sc = o.sr_scalars()
    assert d["n_after_s1"] == sc["n_after_s1"]
    full_d, full_o = h.features(0), o.cloud(0)
    flips = check_cloud(full_d, full_o, "laserCloud", unwrap_bounds(sc["startOri"], sc["endOri"]))
    # startOri / endOri come straight out of atan2f: the same bits
    assert np.float32(d["startOri"]).view(np.uint32) == np.float32(sc["startOri"]).view(np.uint32)
    assert np.float32(d["endOri"]).view(np.uint32) == np.float32(sc["endOri"]).view(np.uint32)
    assert d["N2"] == full_o.shape[0]
    assert np.array_equal(d["scanStartInd"][:shape[0]], o.sr_ints(3))
    assert np.array_equal(d["scanEndInd"][:shape[0]], o.sr_ints(4))
    # inside the sectors everything the reference reads is defined: compare there
    start, end = o.sr_ints(3), o.sr_ints(4)
    cur_o, sort_o, lab_o, pick_o = o.sr_curvature(), o.sr_ints(0), o.sr_ints(2), o.sr_ints(1)
    for r in range(shape[0]):
        if end[r] - start[r] < 6:
            continue
        s, e = start[r], end[r]  # sectors cover [s, e-1]
        assert np.array_equal(d["curvature"][s:e].view(np.uint32), cur_o[s:e].view(np.uint32)), "curvature ring %d" % r
        assert np.array_equal(d["sort"][s:e], sort_o[s:e]), "sort order ring %d" % r
        assert np.array_equal(d["label"][s:e], lab_o[s:e]), "labels ring %d" % r
        assert np.array_equal(d["picked"][s - 5:e + 6], pick_o[s - 5:e + 6]), "picked ring %d" % r
    assert np.array_equal(d["sharpInd"], o.sr_ints(5))
    assert np.array_equal(d["lessSharpInd"], o.sr_ints(6))
    assert np.array_equal(d["flatInd"], o.sr_ints(7))
    for which, name in [(1, "sharp"), (2, "lessSharp"), (3, "flat"), (4, "lessFlat")]:
        check_cloud(h.features(which), o.cloud(which), name, max_flips=flips)


def test_scan_registration_edge_cases(vl, orc, sweeps):
    base = sweeps(64, 512, 2).copy()
    # leading / trailing NaN runs and close returns move the first / last surviving point
    c = base.copy()
    c[:700, :3] = np.nan
    c[-300:, :3] = np.nan
    c[1000:1100, :3] *= 0.01
    h, o = run_both(vl, orc, c)
    sc = o.sr_scalars()
    flips = check_cloud(h.features(0), o.cloud(0), "laserCloud", unwrap_bounds(sc["startOri"], sc["endOri"]))
    for which, name in [(1, "sharp"), (2, "lessSharp"), (3, "flat"), (4, "lessFlat")]:
        check_cloud(h.features(which), o.cloud(which), name, max_flips=flips)
    # shuffled (non ring-major) input exercises the stable per-ring compaction
    rng = np.random.default_rng(0)
    c = base[rng.permutation(base.shape[0])]
    h, o = run_both(vl, orc, c)
    sc = o.sr_scalars()
    check_cloud(h.features(0), o.cloud(0), "laserCloud shuffled", unwrap_bounds(sc["startOri"], sc["endOri"]))
    # all-NaN cloud: the reference would index an empty cloud; the ABI reports VLOAM_ERR_EMPTY
    c = np.full((4096, 4), np.nan, dtype=np.float32)
    hd = vl.Handle(0, with_mapping=0)
    hd.scan_registration(c)
    with pytest.raises(vl.VloamError) as ei:
        hd.laser_odometry()
    assert ei.value.status == vl.ERR_EMPTY
    with pytest.raises(vl.VloamError) as ei:
        vl.Handle(0, max_points=1024).scan_registration(np.zeros((2048, 4), dtype=np.float32))
    assert ei.value.status == vl.ERR_CAPACITY
    with pytest.raises(vl.VloamError) as ei:
        vl.Handle(0, scan_line=48)
    assert ei.value.status == vl.ERR_INVALID


def test_rings_with_a_voxel_for_almost_every_point(vl, orc, synth):
    """The per-ring VoxelGrid sorts RUNS of points sharing a 0.2 m voxel.  Returns from 120 m lie 0.3 m apart: a 2 100-point ring has
    more than 2 048 runs, i.e. more than the small ring tier's power-of-two sort network holds (rank-by-counting path), and every run
    is a single point (run keys with first == last)."""
    n_az = 2100
    el = np.deg2rad(synth.beam_elevations_deg(64))[:, None]
    az = (-2 * np.pi * np.arange(n_az) / n_az)[None, :]
    rng = np.random.default_rng(3)
    rad = 120.0 + 0.01 * rng.standard_normal((64, n_az))
    cloud = np.zeros((64, n_az, 4), dtype=np.float32)
    cloud[..., 0] = rad * np.cos(el) * np.cos(az)
    cloud[..., 1] = rad * np.cos(el) * np.sin(az)
    cloud[..., 2] = rad * np.sin(el)
    cloud = cloud.transpose(1, 0, 2).reshape(-1, 4).copy()   # firing order: all lasers of a column, then the next column
    h = vl.Handle(0, with_mapping=0, max_points=64 * 2304)
    h.scan_registration(cloud)
    o = orc.Oracle(with_mapping=False)
    assert o.scan_registration(cloud) == 0
    sc = o.sr_scalars()
    flips = check_cloud(h.features(0), o.cloud(0), "laserCloud", unwrap_bounds(sc["startOri"], sc["endOri"]))
    less_flat = o.cloud(4)
    per_ring = np.bincount(less_flat[:, 3].astype(np.int64))
    assert per_ring.max() > 2048, "the case must exceed 2 048 voxels in a ring (largest: %d)" % per_ring.max()
    for which, name in [(1, "sharp"), (2, "lessSharp"), (3, "flat"), (4, "lessFlat")]:
        check_cloud(h.features(which), o.cloud(which), name, max_flips=flips)


def test_equal_curvatures_are_picked_in_the_canonical_order(vl, orc, synth):
    """Ties.  std::sort leaves points of equal curvature in an unspecified order (scan_registration.cpp:323); the oracle's canonical order
    is (curvature, index), walked from the top for the sharp picks and from the bottom for the flat picks.  The device never sorts: a pick is
    a wavefront arg-max over pre-masked candidates, the lane that owns the winner is found by ballot and a second reduction only runs when
    two lanes tie (k_sr_ring, run_sector_q).  Every ring of this cloud holds two stretches of points TWICE — the same coordinates 80 columns
    apart (twins in different lanes: the second reduction) and 128 columns apart (twins in the same lane, two register slots apart: the
    slot order of the lane's own search) — so that every candidate inside them has a twin with the same curvature bits."""
    n_az = 2048
    el = np.deg2rad(synth.beam_elevations_deg(64))[:, None]
    az = (-2 * np.pi * np.arange(n_az) / n_az)[None, :]
    rng = np.random.default_rng(11)
    rad = 18.0 + 4.0 * np.sin(3 * az) + 0.05 * rng.standard_normal((64, n_az))
    rad += 1.2 * ((np.arange(n_az) // 37) % 2)[None, :]          # range steps: sharp candidates
    cloud = np.zeros((64, n_az, 4), dtype=np.float32)
    cloud[..., 0] = rad * np.cos(el) * np.cos(az)
    cloud[..., 1] = rad * np.cos(el) * np.sin(az)
    cloud[..., 2] = rad * np.sin(el)
    cloud[:, 100:180] = cloud[:, 20:100]      # sector 0 of every ring: twins 80 points apart
    cloud[:, 528:656] = cloud[:, 400:528]     # sector 1: twins 128 points apart
    cloud = cloud.transpose(1, 0, 2).reshape(-1, 4).copy()   # firing order
    o = orc.Oracle(with_mapping=False)
    assert o.scan_registration(cloud) == 0
    cur, start, end = o.sr_curvature(), o.sr_ints(3), o.sr_ints(4)
    twins = rings = 0
    for r in range(64):   # the construction works: interior points of the copies carry the same curvature bits as their originals
        if end[r] - start[r] < 2000:
            continue      # (the reference drops the lasers it maps to scan ids above 50, scan_registration.cpp:218-221)
        rings += 1
        a = cur[start[r] - 5 + 110:start[r] - 5 + 170].view(np.uint32)
        b = cur[start[r] - 5 + 30:start[r] - 5 + 90].view(np.uint32)
        twins += int(np.count_nonzero(a == b))
    assert rings >= 48 and twins == rings * 60, "the cloud must contain exact curvature ties (%d of %d)" % (twins, rings * 60)
    for debug in (0, 1):   # the production kernel and the debug build (which also emits the sort order)
        h = vl.Handle(0, with_mapping=0, debug=debug)
        h.scan_registration(cloud)
        for which, name in [(1, "sharp"), (2, "lessSharp"), (3, "flat"), (4, "lessFlat")]:
            dev, ref = h.features(which), o.cloud(which)
            assert dev.shape == ref.shape, name
            assert np.array_equal(dev[:, :4].view(np.uint32), ref[:, :4].view(np.uint32)), "%s: picks among equal curvatures differ (debug=%d)" % (name, debug)
        if debug:
            d = h.sr_debug()
            assert np.array_equal(d["sharpInd"], o.sr_ints(5)) and np.array_equal(d["lessSharpInd"], o.sr_ints(6)) and np.array_equal(d["flatInd"], o.sr_ints(7))


def test_scan_registration_errors_of_a_burst_are_not_lost(vl, sweeps):
    """vloam_process_scan bursts rotate four buffer sets and rewrite each set's error word every sweep: an empty sweep (all NaN) or a
    dropped over-long ring in the MIDDLE of a burst must still be reported by the vloam_sync that ends it — once."""
    good = [sweeps(64, 512, k) for k in range(10)]
    h = vl.Handle(0, with_mapping=1)
    for k in range(10):
        h.process_scan(good[k] if k != 3 else np.full((4096, 4), np.nan, dtype=np.float32))
    with pytest.raises(vl.VloamError) as ei:
        h.sync()
    assert ei.value.status == vl.ERR_EMPTY
    h.sync()   # reported once
    # one ring with more points than the LDS-resident ring buffer takes (kMaxRingLen = 4096): the ring is dropped, loudly
    long_ring = np.zeros((6000, 4), dtype=np.float32)
    az = -2 * np.pi * np.arange(6000) / 6000
    el = np.deg2rad(-10.43)   # beam 35 of the HDL-64E table
    long_ring[:, 0], long_ring[:, 1], long_ring[:, 2] = 20 * np.cos(el) * np.cos(az), 20 * np.cos(el) * np.sin(az), 20 * np.sin(el)
    h2 = vl.Handle(0, with_mapping=0)
    h2.process_scan(good[0]); h2.process_scan(long_ring)
    for k in range(1, 6):
        h2.process_scan(good[k])
    with pytest.raises(vl.VloamError) as ei:
        h2.sync()
    assert ei.value.status == vl.ERR_CAPACITY
    h2.sync()


def test_ring_tiers_follow_the_ring_length(vl, orc, synth):
    """k_sr_ring runs as a 2176-point tier (two rings per CU) followed by the 4096-point tier: its full grid once a ring has come within 32
    points of the small tier's capacity (host-mapped watch word), otherwise one catch-all workgroup that only works on rings the small tier
    had to leave.  A DEFAULT handle must take any ring of up to 4096 points on any sweep, like the reference's 400 000-point scratch takes
    any ring (scan_registration.h:90): rings that grow gradually (1 792 -> 2 160 -> 2 300 points), and rings that jump from ~2 000 to ~3 000
    points between two sweeps at sweep 12, long after the start and without any warning — stage-wise and streamed (the host runs ahead of
    the kernels and of the watch word): features equal the oracle's on every sweep."""
    def sweep(n_az, k):
        return synth.SynthSequence(n_rings=64, n_azimuth=n_az, n_sweeps=k + 1).sweep(k)

    def check(h, cloud, what):
        o = orc.Oracle(with_mapping=False)
        assert o.scan_registration(cloud) == 0
        sc = o.sr_scalars()
        flips = check_cloud(h.features(0), o.cloud(0), "laserCloud " + what, unwrap_bounds(sc["startOri"], sc["endOri"]))
        for which, name in [(1, "sharp"), (2, "lessSharp"), (3, "flat"), (4, "lessFlat")]:
            check_cloud(h.features(which), o.cloud(which), "%s %s" % (name, what), max_flips=flips)

    h = vl.Handle(0, with_mapping=0, max_points=64 * 2304)
    plan = [1792] * 10 + [2160] * 2 + [2300] * 3 + [1792] * 2
    for k, n_az in enumerate(plan):
        cloud = sweep(n_az, k)
        h.reset_frame()
        h.scan_registration(cloud)
        check(h, cloud, "sweep %d (%d columns)" % (k, n_az))
        h.laser_odometry()
    h.sync()
    # the jump without a warning: 2 000 -> 3 000 points per ring at sweep 12, and back
    jump = [2000] * 12 + [3000] * 3 + [2000] * 2
    h2 = vl.Handle(0, with_mapping=0, max_points=64 * 3072)
    for k, n_az in enumerate(jump):
        cloud = sweep(n_az, k)
        h2.reset_frame()
        h2.scan_registration(cloud)
        if k >= 10:
            check(h2, cloud, "sweep %d (jump to %d columns)" % (k, n_az))
        h2.laser_odometry()
    h2.sync()
    h3 = vl.Handle(0, with_mapping=0, max_points=64 * 3072)
    clouds = [sweep(n_az, k) for k, n_az in enumerate(jump)]
    for c in clouds:        # streamed
        h3.process_scan(c)
    h3.sync()               # (raises if any sweep was reported instead of processed)
    check(h3, clouds[-1], "last sweep of the streamed run")
    # beyond the big tier's capacity (4096 points per ring) the handle still says so
    h4 = vl.Handle(0, with_mapping=0, max_points=64 * 4400)
    with pytest.raises(vl.VloamError) as e:
        h4.process_scan(sweep(4300, 0))
        h4.sync()
    assert e.value.status == vl.ERR_CAPACITY


_OPT_OUT_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import conftest
vl = conftest.load_pkg(); synth = conftest.load_synth()
import orc
orc.build()
def sweep(n_az, k):
    return synth.SynthSequence(n_rings=64, n_azimuth=n_az, n_sweeps=k + 1).sweep(k)
# ordinary sweeps: same features as the oracle without the catch-all launch
h = vl.Handle(0, with_mapping=0)
for k in range(3):
    c = sweep(2048, k)
    h.reset_frame(); h.scan_registration(c)
    o = orc.Oracle(with_mapping=False); assert o.scan_registration(c) == 0
    for which in (1, 2, 3, 4):
        d, r = h.features(which), o.cloud(which)
        assert d.shape == r.shape and np.array_equal(d[:, :4].view(np.uint32), r[:, :4].view(np.uint32)), (k, which)
    h.laser_odometry()
h.sync()
# a ring that outgrows the small tier WITHOUT the watch word's warning is reported, not processed
h2 = vl.Handle(0, with_mapping=0, max_points=64 * 3072)
try:
    h2.process_scan(sweep(3000, 0)); h2.sync()
    print("NOT-REPORTED")
except vl.VloamError as e:
    print("REPORTED" if e.status == vl.ERR_CAPACITY else "OTHER-ERROR %d" % e.status)
# with the warning (rings growing towards the capacity) the full big tier runs as always
h3 = vl.Handle(0, with_mapping=0, max_points=64 * 2304)
for k, n_az in enumerate([2160] * 3 + [2300] * 2):
    h3.reset_frame(); h3.scan_registration(sweep(n_az, k)); h3.laser_odometry()
h3.sync()
print("GROWN-OK")
"""


def test_catch_all_opt_out_reports_instead_of_processing():
    """VLOAM_SR_CATCHALL=0 (read once per process): no catch-all workgroup behind the small ring tier — for hosts that know their ring lengths
    (13 - 16 us of single-sweep latency).  Ordinary sweeps give the oracle's features; a ring that jumps past the small tier's capacity without
    the watch word's warning comes back as VLOAM_ERR_CAPACITY (the rounds 2 - 3 behaviour, now opt-in); rings that GROW towards the capacity
    still switch the full big tier on."""
    import subprocess
    import sys
    env = dict(os.environ, VLOAM_SR_CATCHALL="0")
    r = subprocess.run([sys.executable, "-c", _OPT_OUT_SCRIPT, os.path.dirname(os.path.abspath(__file__))], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "REPORTED" in r.stdout.split() and "GROWN-OK" in r.stdout.split(), r.stdout[-1500:]
