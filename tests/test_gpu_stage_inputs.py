"""-m gpu: clouds and poses EDITED between the façade's stages.

The reference hands clouds and the odometry pose from stage to stage by value and every stage deep-copies what it is given
(LaserOdometry::input laser_odometry.cpp:135-146, LaserMapping::input laser_mapping.cpp:167-196): a caller may thin, filter or replace them
in between.  Here the stages exchange their results on the device; vloam_set_odometry_input / vloam_set_mapping_input upload what the caller
changed.  Oracle: the same stage-by-stage sequence with the same edits (orc_stage_sr / orc_set_sr_cloud / orc_stage_lo / orc_stage_map).
Rounds 1 - 5 refused a substituted cloud (std::invalid_argument from compat.hpp)."""
import numpy as np
import pytest

from test_gpu_laser_mapping import lexsort_rows, oracle_map_points, qdist

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-8


def edited(clouds5, k):
    """The caller's edit of scan registration's five clouds for sweep k (order-preserving, as a filter would)."""
    full, sharp, less_sharp, flat, less_flat = [c.copy() for c in clouds5]
    out = [None] * 5
    if k % 3 == 1:    # a thinning filter on the two less-clouds + a few features dropped
        out[2] = less_sharp[np.arange(less_sharp.shape[0]) % 7 != 3]
        out[4] = less_flat[np.arange(less_flat.shape[0]) % 5 != 0]
        out[1] = sharp[2:]
    elif k % 3 == 2:  # an object mask: everything within 6 m in front is removed from all five; the flat features nudged
        def keep(c):
            return c[~((c[:, 0] > 0) & (np.abs(c[:, 1]) < 3.0) & (c[:, 0] < 12.0))]
        out = [keep(full), keep(sharp), keep(less_sharp), keep(flat), keep(less_flat)]
        out[3] = out[3].copy()
        out[3][:, 2] += np.float32(0.003)
    return out


def test_edited_clouds_between_scan_registration_and_odometry(vl, orc, sweeps):
    n, shape = 9, (64, 512)
    h = vl.Handle(0, with_mapping=1, debug=1)
    o = orc.Oracle(with_mapping=True)
    n_edits = 0
    for k in range(n):
        cloud = sweeps(shape[0], shape[1], k)
        h.reset_frame()
        h.scan_registration(cloud)
        assert o.stage_sr(cloud) == 0
        five = [h.features(w) for w in range(5)]
        for w in range(5):   # all four floats (intensity = scan line + 0.1 relTime goes through atan2f: the device computes it the way glibc does)
            assert np.array_equal(np.ascontiguousarray(five[w][:, :4]).view(np.uint32), np.ascontiguousarray(o.cloud(w)[:, :4]).view(np.uint32))
        ed = edited(five, k)
        if any(e is not None for e in ed):
            n_edits += 1
            h.set_odometry_input(*ed)
            for w in range(5):
                if ed[w] is not None:
                    o.set_sr_cloud(w, ed[w])
                    assert np.array_equal(h.features(w).view(np.uint32), np.ascontiguousarray(ed[w]).view(np.uint32)), "the device holds the edited cloud %d" % w
        qw, tw, _, _ = h.laser_odometry()
        o.stage_lo()
        oq, ot, _, _ = o.lo_pose()
        assert qdist(qw, oq) < POSE_TOL and np.linalg.norm(tw - ot) < POSE_TOL, "odometry pose, sweep %d" % k
        if k > 0:
            for outer in range(2):
                d = h.lo_debug(outer)
                oc, op = o.lo_corr(outer)
                assert np.array_equal(d["corner"], oc) and np.array_equal(d["plane"], op), "correspondences, sweep %d round %d" % (k, outer)
        # LaserOdometry::output hands the (edited) less-clouds on as CornerLast / SurfLast
        for which in (5, 6):
            assert np.array_equal(h.features(which)[:, :4].view(np.uint32), o.cloud(which)[:, :4].view(np.uint32))
        qm, tm = h.laser_mapping()
        assert o.stage_map() == 0
        for which in (7, 8):
            assert np.array_equal(h.features(which)[:, :4].view(np.uint32), o.cloud(which)[:, :4].view(np.uint32)), "stack %d, sweep %d" % (which, k)
        oq, ot, _, _ = o.map_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "map pose, sweep %d" % k
    assert n_edits >= 5
    for kind in (0, 1):
        _, pts = h.map_dump(kind)
        ref = oracle_map_points(o, kind)
        assert pts.shape == ref.shape and np.array_equal(lexsort_rows(pts)[:, :4].view(np.uint32), lexsort_rows(ref)[:, :4].view(np.uint32))
    h.close()


@pytest.mark.parametrize("skip", [1, 2])
def test_edited_inputs_between_odometry_and_mapping(vl, orc, sweeps, skip):
    """LaserMapping::input with thinned clouds and a nudged odometry pose: this sweep's mapping works on them, the odometry's own CornerLast /
    SurfLast (next sweep's search clouds) do not change — separate copies in the reference."""
    n, shape = 8, (64, 512)
    h = vl.Handle(0, with_mapping=1, mapping_skip_frame=skip)
    o = orc.Oracle(with_mapping=True, mapping_skip_frame=skip)
    for k in range(n):
        cloud = sweeps(shape[0], shape[1], k)
        h.reset_frame()
        h.scan_registration(cloud)
        qw, tw, _, _ = h.laser_odometry()
        assert o.stage_sr(cloud) == 0
        o.stage_lo()
        oq, ot, _, _ = o.lo_pose()
        assert qdist(qw, oq) < POSE_TOL and np.linalg.norm(tw - ot) < POSE_TOL, "odometry pose, sweep %d" % k
        dq, dt = h.odometry_pose()
        assert np.array_equal(dq, qw) and np.array_equal(dt, tw)
        corner, surf, full = h.features(5), h.features(6), h.features(0)
        kw, okw = {}, {}
        if k % 2 == 1:
            c2, s2, f2 = corner[::2].copy(), surf[np.arange(surf.shape[0]) % 3 != 1].copy(), full[: full.shape[0] // 2].copy()
            kw.update(laserCloudCornerLast=c2, laserCloudSurfLast=s2, laserCloudFullRes=f2)
            okw.update(corner=c2, surf=s2, full=f2)
        if k in (2, 3, 6):
            # a UNIT quaternion, like every Eigen::Quaterniond an odometry hands on: the solver's closed-form Jacobians (d lp / d delta =
            # -2 [R p]x) are those of a rotation; with |q|^2 = 1 + 5e-6 the reference's autodiff and they part at that relative size and the
            # poses after 2 x 4 iterations by 1e-9 (measured) — stated in c_api.h
            q2 = qw + np.array([1e-3, -2e-3, 5e-4, 0.0])
            q2 = q2 / np.linalg.norm(q2)
            t2 = tw + np.array([0.02, -0.01, 0.005])
            kw.update(q_wodom_curr=q2, t_wodom_curr=t2)
            okw.update(q=q2, t=t2)
        if kw:
            h.set_mapping_input(**kw)
        qm, tm = h.laser_mapping()
        assert o.stage_map(**okw) == 0
        oq, ot = o.map_published_pose()
        assert qdist(qm, oq) < POSE_TOL and np.linalg.norm(tm - ot) < POSE_TOL, "map pose, sweep %d" % k
        skipped = ((k + 1) % skip) != 0
        if "laserCloudFullRes" in kw and not skipped:   # /velodyne_cloud_registered is the full-resolution cloud as handed in (laser_mapping.cpp:795-799)
            reg, ref = h.features(11), o.cloud(11)
            assert reg.shape == ref.shape and np.allclose(reg[:, :3], ref[:, :3], rtol=0, atol=1e-5)
        # the odometry's clouds are untouched
        assert np.array_equal(h.features(5).view(np.uint32), corner.view(np.uint32)) and np.array_equal(h.features(6).view(np.uint32), surf.view(np.uint32))
    tj = h.trajectory()
    assert tj.shape[0] == n
    for kind in (0, 1):
        _, pts = h.map_dump(kind)
        ref = oracle_map_points(o, kind)
        assert pts.shape == ref.shape and np.array_equal(lexsort_rows(pts)[:, :4].view(np.uint32), lexsort_rows(ref)[:, :4].view(np.uint32))
    h.close()


def test_stage_input_call_order_and_capacity(vl, sweeps):
    h = vl.Handle(0, with_mapping=1)
    c = sweeps(64, 512, 0)
    with pytest.raises(vl.VloamError) as e:
        h.set_odometry_input(cornerPointsSharp=c[:10])
    assert e.value.status == vl.ERR_ORDER
    h.reset_frame()
    h.scan_registration(c)
    with pytest.raises(vl.VloamError) as e:
        h.set_mapping_input(laserCloudCornerLast=c[:10])
    assert e.value.status == vl.ERR_ORDER
    with pytest.raises(vl.VloamError) as e:
        h.set_odometry_input(cornerPointsSharp=np.zeros((769, 4), np.float32))
    assert e.value.status == vl.ERR_CAPACITY
    h.set_odometry_input(cornerPointsSharp=np.zeros((0, 4), np.float32))   # an EMPTY substituted cloud is a cloud
    assert h.features(1).shape[0] == 0
    h.laser_odometry()
    h.laser_mapping()
    hb = vl.Handle(0, n_sessions=2, with_mapping=1)
    with pytest.raises(vl.VloamError) as e:
        hb.set_odometry_input(cornerPointsSharp=c[:10])
    assert e.value.status == vl.ERR_INVALID
    hb.close()
    h.close()
