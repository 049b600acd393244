#!/usr/bin/env python3
"""Idle time of the mapping stream between consecutive sweeps in an UNTRACED streaming run: k_map_prepare and k_map_finalize stamp the
constant-rate clock (100 MHz) on the device (VLOAM_TS_LOG=1), so no profiler sits between the host and the queues.
  VLOAM_TS_LOG=1 python tools/map_stream_gaps.py [--sweeps 400]"""
import argparse
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["VLOAM_TS_LOG"] = "1"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=400)
ap.add_argument("--resident", type=int, default=96)
ap.add_argument("--serial", action="store_true", help="sync after every sweep: the chain without the other streams' kernels of the next sweeps beside it")
a = ap.parse_args()
synth = conftest.load_synth()
_SEQ = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=a.resident + 1)


def _w(k):
    return _SEQ.sweep(k)


with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
    host = np.stack(pool.map(_w, range(a.resident), chunksize=2))
vl = conftest.load_pkg()
import torch  # noqa: E402

d = torch.from_numpy(host).cuda()
npts = host.shape[1]
order, pos, step = [], -1, 1
for _ in range(a.sweeps):   # back and forth over the resident sweeps: a continuous drive
    if pos + step < 0 or pos + step > a.resident - 1:
        step = -step
    pos += step
    order.append(pos)
h = vl.Handle(0, with_mapping=1, max_frames=a.sweeps + 8)
t0 = time.perf_counter()
for k in order:
    h.process_scan_device(d.data_ptr() + k * npts * 16, npts)
    if a.serial:
        h.sync()
h.sync()
dt = time.perf_counter() - t0
ts = h.debug_raw(2, 72, np.int64).reshape(1024, 2).astype(np.float64) * 0.01   # microseconds
n = a.sweeps
idx = np.arange(max(n - 300, 8), n) % 1024
prep, fin = ts[idx, 0], ts[idx, 1]
period = np.diff(prep)
span_to_finalize = fin - prep
gap = prep[1:] - fin[:-1]
print("%d sweeps, %.1f us per sweep by the host clock" % (n, 1e6 * dt / n))
for name, v in (("period (prepare to prepare)", period), ("prepare start -> finalize start", span_to_finalize), ("finalize start -> next prepare start", gap)):
    print("  %-38s median %7.1f  p10 %7.1f  p90 %7.1f  max %7.1f us" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
print("  (finalize itself runs ~9 us: whatever the last line shows beyond that is idle time of the mapping stream)")
h.close()
