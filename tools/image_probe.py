#!/usr/bin/env python3
"""Image front-end alone (Shi-Tomasi corners + pyramidal Lucas-Kanade on the device), images resident in HBM: images/s and the per-kernel
HIP-event table, on a texture pair (1 024 corners: the maxCorners cut) and on two renders of the synthetic LiDAR scene.
  python tools/image_probe.py [--no-table] [--reps 200]        (rocprofv3 --kernel-trace --stats -- python tools/image_probe.py --no-table)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--no-table", action="store_true")
a = ap.parse_args()
vl = conftest.load_pkg()
synth = conftest.load_synth()
import torch  # noqa: E402

W, H = 1242, 375
prev, nxt, _ = synth.synth_image_pair(W, H, seed=3)
seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=4)
scene = [synth.render_image(seq, k) for k in range(2)]
for name, pair in (("texture pair", (prev, nxt)), ("rendered scene", scene)):
    d = torch.from_numpy(np.stack(pair)).cuda()
    h = vl.Handle(0, with_mapping=0, image_width=W, image_height=H)
    for k in range(20):
        h.vo_process_image_device(d.data_ptr() + (k % 2) * W * H, W, H)
    h.sync()
    t0 = time.perf_counter()
    for k in range(a.reps):
        h.vo_process_image_device(d.data_ptr() + (k % 2) * W * H, W, H)
    h.sync()
    dt = time.perf_counter() - t0
    print("%s: %.1f us per image (%.0f images/s), corners %d, candidates %d" %
          (name, 1e6 * dt / a.reps, a.reps / dt, h.vo_keypoints().shape[0], int(h.debug_raw(4, 8, np.int32)[0])))
    if not a.no_table:
        h.profile_kernel("*", 8192)
        for k in range(50):
            h.vo_process_image_device(d.data_ptr() + (k % 2) * W * H, W, H)
        for nm, (ms, cnt) in sorted(h.profile_table().items(), key=lambda kv: -kv[1][0]):
            print("   %-18s %7.1f us x %d" % (nm, 1e3 * ms / cnt, cnt // 50))
    h.close()
