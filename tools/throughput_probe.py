#!/usr/bin/env python3
"""Quick steady-state throughput probe (seconds, not minutes): one sequence and B batched sequences of 64 x 2048 sweeps with mapping,
after a short warm-up.  For A/B-ing a kernel change; bench.py remains the measurement of record.
  python tools/throughput_probe.py [--sessions 8] [--warm 60] [--steps 120] [--table]"""
import argparse
import multiprocessing as mp
import os
import sys
import time

# the stage streams of a handle (+ torch's) must not share hardware queues (the runtime's default is 4); before the HIP runtime loads
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sessions", type=int, default=8)
ap.add_argument("--warm", type=int, default=60)
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--sweeps", type=int, default=48)
ap.add_argument("--table", action="store_true", help="per-kernel HIP-event table of the batched run")
ap.add_argument("--no-single", action="store_true")
ap.add_argument("--map-log2", type=int, default=22, help="map_capacity_log2 of the handles (voxel table slots)")
ap.add_argument("--procs", type=int, default=32, help="worker processes for the synthesis (1 under rocprofv3: it follows forked children)")
ap.add_argument("--from-idle", default="", help="comma-separated sweep counts: time that many sweeps of ONE sequence from a drained pipeline (5 repeats each) "
                "and fit time = fill + period x sweeps; replaces the normal run")
a = ap.parse_args()
synth = conftest.load_synth()
_SEQ = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=a.sweeps + 1)


def _w(k):
    return _SEQ.sweep(k)


if a.procs > 1:
    with mp.get_context("fork").Pool(min(a.procs, os.cpu_count() or 1)) as pool:   # before the HIP runtime loads
        host = np.stack(pool.map(_w, range(a.sweeps), chunksize=2))
else:
    host = np.stack([_w(k) for k in range(a.sweeps)])
vl = conftest.load_pkg()
import torch  # noqa: E402

d = torch.from_numpy(host).cuda()
npts = host.shape[1]
ptr = lambda k: d.data_ptr() + (k % a.sweeps) * npts * 16   # noqa: E731  (the sequence wraps: a jump back every a.sweeps, same for every variant)


def run(B):
    h = vl.Handle(0, n_sessions=B, with_mapping=1, max_frames=a.warm + a.steps + 8, map_capacity_log2=a.map_log2)
    step = (lambda k: h.process_scan_device(ptr(k), npts)) if B == 1 else (lambda k: h.batch_process_scan_device([ptr(k)] * B, [npts] * B))
    for k in range(a.warm):
        step(k)
    h.sync()
    if a.table:
        h.profile_kernel("*", 16384)
    t0 = time.perf_counter()
    for k in range(a.warm, a.warm + a.steps):
        step(k)
    h.sync()
    dt = time.perf_counter() - t0
    print("B = %2d: %8.0f scans/s   (%.1f us per step)" % (B, B * a.steps / dt, 1e6 * dt / a.steps), flush=True)
    if a.table:
        rows = h.profile_table()
        tot = sum(ms for ms, _ in rows.values())
        for name, (ms, cnt) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:14]:
            print("   %-20s %6.1f us x %5d  (%4.1f %%)" % (name, 1e3 * ms / max(cnt, 1), cnt, 100 * ms / tot))
    tr = h.trajectory(0, a.warm + a.steps)
    h.close()
    return tr


def from_idle(counts):
    h = vl.Handle(0, with_mapping=1, max_frames=a.warm + 6 * sum(counts) + 64)
    k = 0
    for _ in range(a.warm):
        h.process_scan_device(ptr(k), npts); k += 1
    h.sync()
    xs, ys = [], []
    for n in counts:
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                h.process_scan_device(ptr(k), npts); k += 1
            h.sync()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("%4d sweeps from idle: median %.3f ms (min %.3f)  -> %6.0f scans/s" % (n, 1e3 * ts[2], 1e3 * ts[0], n / ts[2]), flush=True)
        xs.append(n); ys.append(ts[2])
    if len(xs) >= 2:
        b, c = np.polyfit(xs, ys, 1)
        print("fit: fill %.3f ms + %.1f us per sweep" % (1e3 * c, 1e6 * b))
    h.close()


if a.from_idle:
    from_idle([int(v) for v in a.from_idle.split(",")])
    sys.exit(0)
t1 = None if a.no_single else run(1)
tb = run(a.sessions)
if t1 is not None:
    print("batched session 0 identical to the single sequence:", bool(np.array_equal(t1, tb)))
