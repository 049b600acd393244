#!/usr/bin/env python3
"""The coupled loop driven from raw inputs (vloam_process_frame_image_device: sweep + grey image per frame, detach_VO_LO = 0) on its own,
with a short LiDAR-only lead-in: frames/s, the per-kernel HIP-event table and — under rocprofv3 --kernel-trace — the stream schedule of
the leg bench.py reports as configs3_from_images.
  python tools/image_leg_probe.py [--lead 30] [--frames 24] [--table] [--no-mapping]"""
import argparse
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lead", type=int, default=30)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--reps", type=int, default=3, help="timed passes over the image frames (a fresh handle each)")
ap.add_argument("--table", action="store_true")
ap.add_argument("--no-mapping", action="store_true")
ap.add_argument("--procs", type=int, default=32)
ap.add_argument("--dummy-streams", type=int, default=0, help="HIP streams created (and kept) before the handle: shifts which hardware queues the handle's stage streams land on")
ap.add_argument("--lidar", action="store_true", help="also time the LiDAR-only loop (process_scan_device) on a handle of its own")
a = ap.parse_args()
synth = conftest.load_synth()
T = a.lead + a.frames
_SEQ = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=T)


def _w(job):
    return _SEQ.sweep(job[1]) if job[0] == "s" else synth.render_image(_SEQ, job[1])


jobs = [("s", k) for k in range(T)] + [("i", k) for k in range(a.lead, T)]
with mp.get_context("fork").Pool(min(a.procs, os.cpu_count() or 1)) as pool:   # before the HIP runtime loads
    res = pool.map(_w, jobs, chunksize=1)
host = np.stack(res[:T])
images = np.stack(res[T:])
vl = conftest.load_pkg()
import torch  # noqa: E402

dummies = [torch.cuda.Stream() for _ in range(a.dummy_streams)]
for st in dummies:
    with torch.cuda.stream(st):
        torch.zeros(16, device="cuda").add_(1)   # the runtime binds a hardware queue at first use
torch.cuda.synchronize()
d = torch.from_numpy(host).cuda()
d_img = torch.from_numpy(images).cuda()
ni, IH, IW = images.shape
npts = host.shape[1]
for rep in range(a.reps):
    h = vl.Handle(0, with_mapping=0 if a.no_mapping else 1, max_frames=T + 8, detach_VO_LO=0, image_width=IW, image_height=IH)
    h.vo_set_calib(*synth.kitti_like_calib())
    h.set_extrinsics(*synth.kitti_like_extrinsics())
    for k in range(a.lead):
        h.process_scan_device(d.data_ptr() + k * npts * 16, npts)
    h.process_frame_image_device(d.data_ptr() + a.lead * npts * 16, npts, d_img.data_ptr(), IW, IH)
    h.sync()
    if a.table and rep == a.reps - 1:
        h.profile_kernel("*", 64 * ni + 64)
    t0 = time.perf_counter()
    for j in range(1, ni):
        h.process_frame_image_device(d.data_ptr() + (a.lead + j) * npts * 16, npts, d_img.data_ptr() + j * IW * IH, IW, IH)
    h.sync()
    dt = time.perf_counter() - t0
    r = h.vo_result()
    print("pass %d: %.1f us per frame (%.0f frames/s), corners %d, counter32 %d, counter22 %d" %
          (rep, 1e6 * dt / (ni - 1), (ni - 1) / dt, h.vo_keypoints().shape[0], r["counter32"], r["counter22"]), flush=True)
    if a.table and rep == a.reps - 1:
        for nm, (ms, cnt) in sorted(h.profile_table().items(), key=lambda kv: -kv[1][0]):
            print("   %-22s %7.1f us x %5.2f per frame" % (nm, 1e3 * ms / cnt, cnt / (ni - 1)))
    h.close()

if a.lidar:
    for rep in range(a.reps):
        h = vl.Handle(0, with_mapping=0 if a.no_mapping else 1, max_frames=4 * T + 8)
        for k in range(T):
            h.process_scan_device(d.data_ptr() + k * npts * 16, npts)
        h.sync()
        t0 = time.perf_counter()
        for k in list(range(T - 1, -1, -1)) + list(range(T)):
            h.process_scan_device(d.data_ptr() + k * npts * 16, npts)
        h.sync()
        dt = time.perf_counter() - t0
        print("LiDAR-only pass %d: %.1f us per sweep (%.0f scans/s)" % (rep, 1e6 * dt / (2 * T), 2 * T / dt), flush=True)
        h.close()
