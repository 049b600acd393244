"""-m gpu: host sweeps in (what the reference's callback hands over: a host cloud per sweep, scan_registration.cpp:131-152,
vloam_main_node.cpp:125-180).  vloam_process_scan / vloam_batch_process_scan copy a host sweep into a ring of four device input buffers on a copy
stream of the handle and ENQUEUE the sweep with the next call (the copy of sweep k + 1 runs beside the scan registration of sweep k;
VLOAM_STAGE_INLINE=1: copy in front of the sweep's scan registration on that stream, nothing deferred); pinned memory (hipHostMalloc /
hipHostRegister) is read by DMA, pageable memory by hipMemcpyAsync (the call returns with the source consumed).  Whatever the source and the form,
the results are those of the same sweeps handed over as device pointers, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sweeps(synth, n, shape=(64, 512), seed=7):
    seq = synth.SynthSequence(n_rings=shape[0], n_azimuth=shape[1], n_sweeps=n + 1, seed_scene=seed, seed_traj=seed + 1, seed_noise=seed + 2)
    return [seq.sweep(k) for k in range(n)]


def test_pinned_pageable_and_device_sweeps_give_the_same_results(vl, synth):
    import torch
    n = 12
    clouds = _sweeps(synth, n)
    # sweeps of different sizes inside ONE pinned allocation (interior pointers; the last sweeps are cut short: not a multiple of the copy tile)
    sizes = [c.shape[0] - 37 * k for k, c in enumerate(clouds)]
    clouds = [np.ascontiguousarray(c[:m]) for c, m in zip(clouds, sizes)]
    flat = np.concatenate(clouds)
    pinned = torch.from_numpy(flat).pin_memory()
    dev = torch.from_numpy(flat).cuda()
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    runs = {}
    for how in ("device", "pinned", "pageable"):
        h = vl.Handle(0, with_mapping=1)
        for k in range(n):
            if how == "device":
                h.process_scan_device(dev.data_ptr() + int(offs[k]) * 16, sizes[k])
            elif how == "pinned":
                h.process_scan_host_ptr(pinned.data_ptr() + int(offs[k]) * 16, sizes[k])
            else:
                c = clouds[k].copy()
                h.process_scan(c)
                c[:] = np.nan          # pageable: the call has taken its copy
        h.sync()
        runs[how] = (h.trajectory().copy(), h.get_map().copy(), [h.features(w).copy() for w in (0, 2, 4, 7, 8)], h.counts())
        h.close()
    ref = runs["device"]
    assert np.isfinite(ref[0]).all() and ref[1].shape[0] > 1000
    for how in ("pinned", "pageable"):
        t, m, f, c = runs[how]
        assert np.array_equal(t, ref[0]), how
        assert np.array_equal(m.view(np.uint32), ref[1].view(np.uint32)), how
        for a, b in zip(f, ref[2]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), how
        assert c == ref[3], how


def test_batched_sessions_from_pinned_and_pageable_sweeps(vl, synth):
    """One call, four sessions: two hand over pinned sweeps, two pageable ones."""
    import torch
    n, B = 8, 4
    seqs = [_sweeps(synth, n, seed=100 + 11 * b) for b in range(B)]
    pinned = [[torch.from_numpy(c).pin_memory() for c in seqs[b]] for b in range(2)]
    hb = vl.Handle(0, n_sessions=B, with_mapping=1)
    for k in range(n):
        hb.batch_process_scan([pinned[b][k].numpy() if b < 2 else seqs[b][k] for b in range(B)])
    hb.sync()
    hd = vl.Handle(0, n_sessions=B, with_mapping=1)
    dev = [[torch.from_numpy(c).cuda() for c in seqs[b]] for b in range(B)]
    for k in range(n):
        hd.batch_process_scan_device([dev[b][k].data_ptr() for b in range(B)], [seqs[b][k].shape[0] for b in range(B)])
    hd.sync()
    for b in range(B):
        assert np.array_equal(hb.select(b).trajectory(), hd.select(b).trajectory()), b
        assert np.array_equal(hb.get_map().view(np.uint32), hd.get_map().view(np.uint32)), b
    hb.close()
    hd.close()


def test_inline_form_gives_the_same_results():
    """VLOAM_STAGE_INLINE=1 (read when the library loads): the copy on the scan-registration stream itself, kept selectable next to the default
    (deferred ring on a copy stream).  The two tests above, in a process of their own with the inline form on."""
    import os
    import subprocess
    import sys
    if os.environ.get("VLOAM_STAGE_INLINE") == "1":
        pytest.skip("already inside the inline-form process")
    env = dict(os.environ, VLOAM_STAGE_INLINE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "pinned", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_a_deferred_host_sweep_keeps_its_place_in_the_sequence(vl, synth):
    """The host sweep of vloam_process_scan is enqueued by the NEXT call, its copy in flight meanwhile.  Whatever that next call is — another host
    sweep, a device-pointer sweep, a stage-wise call, a getter — the older sweep goes first, and it counts from the moment it was handed over."""
    import torch
    n = 12
    clouds = _sweeps(synth, n, seed=21)
    dev = [torch.from_numpy(c).cuda() for c in clouds]
    ref = vl.Handle(0, with_mapping=1)
    for k in range(n):
        ref.process_scan_device(dev[k].data_ptr(), clouds[k].shape[0])
    ref.sync()
    h = vl.Handle(0, with_mapping=1)
    poses_mid = None
    for k in range(n):
        if k % 4 == 1:
            h.process_scan_device(dev[k].data_ptr(), clouds[k].shape[0])          # behind a host sweep still in flight
        elif k % 4 == 3:
            h.reset_frame()                                                       # stage by stage, behind a host sweep still in flight
            h.scan_registration(clouds[k])
            h.laser_odometry()
            h.laser_mapping()
        else:
            c = clouds[k].copy()
            h.process_scan(c)
            c[:] = np.nan
        assert h.frame_count() == k + 1
        if k == 6:
            poses_mid = h.trajectory().copy()                                     # a getter: sweep 6 (host, just handed over) is in it
    assert poses_mid.shape[0] == 7 and np.array_equal(poses_mid, ref.trajectory()[:7])
    h.sync()
    assert np.array_equal(h.trajectory(), ref.trajectory())
    assert np.array_equal(h.get_map().view(np.uint32), ref.get_map().view(np.uint32))
    h.close()
    ref.close()


def test_a_full_trajectory_log_is_refused_by_the_call_that_overflows_it(vl, synth):
    clouds = _sweeps(synth, 4, seed=33)
    h = vl.Handle(0, with_mapping=1, max_frames=3)
    for k in range(3):
        h.process_scan(clouds[k])
    with pytest.raises(vl.VloamError) as e:
        h.process_scan(clouds[3])          # two sweeps enqueued + one in flight = 3: refused here, not one call later
    assert e.value.status == vl.ERR_CAPACITY
    h.sync()
    assert h.frame_count() == 3 and np.isfinite(h.trajectory()).all() and h.trajectory().shape[0] == 3
    h.close()
