#!/usr/bin/env python3
"""Multi-session scaling probe: H handles x B sessions per handle on ONE GPU, every session its OWN synthetic sequence
(multi.rank_sequence_seeds), 64 x 2048 sweeps with mapping at the steady-state map of a short warm-up.

  python tools/batch_scaling.py --configs 1x1,1x2,1x4,1x8,1x12,1x16,2x4,2x8,4x4 [--table] [--threads]

A config "HxB" drives H independent batched handles of B sessions each (H = 1: the plain batched handle).  With --threads every
handle is enqueued from its own host thread (ctypes releases the GIL inside the library), otherwise one thread takes the handles in
turn.  Prints one line per config: scans/s over all sessions, microseconds per step of one handle, and (--table) the HIP-event
duration of every kernel.  bench.py remains the measurement of record; this is the tool behind profiles/r05_batch_scaling.txt.
The sweeps of a sequence are replayed back and forth (a continuous drive: DISTORTION = 0), so the map stays at its size."""
import argparse
import multiprocessing as mp
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1x1,1x2,1x4,1x8,1x12,1x16")
ap.add_argument("--warm", type=int, default=60)
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--sweeps", type=int, default=40, help="resident sweeps per sequence (replayed back and forth)")
ap.add_argument("--table", action="store_true", help="per-kernel HIP-event table (a second, separately timed replay)")
ap.add_argument("--threads", action="store_true", help="one host thread per handle")
ap.add_argument("--same", action="store_true", help="every session replays sequence 0 (the round-4 bench legs did)")
ap.add_argument("--procs", type=int, default=32)
ap.add_argument("--cache", default="", help=".npy file the synthesised sweeps are kept in (rocprofv3 passes: synthesise once, outside the profiler)")
ap.add_argument("--no-mapping", action="store_true", help="configs[1]: scan registration + odometry only")
a = ap.parse_args()

configs = [tuple(int(v) for v in c.split("x")) for c in a.configs.split(",") if c]
n_seq = 1 if a.same else max(h * b for h, b in configs)
synth = conftest.load_synth()
multi = __import__("importlib").import_module("vloam_amd.multi")
_SEQS = [synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=a.sweeps + 1, **multi.rank_sequence_seeds(s)) for s in range(n_seq)]


def _w(job):
    return _SEQS[job[0]].sweep(job[1])


jobs = [(s, k) for s in range(n_seq) for k in range(a.sweeps)]
t0 = time.perf_counter()
_c = np.load(a.cache, mmap_mode="r") if (a.cache and os.path.exists(a.cache)) else None
if _c is not None and _c.shape[0] >= n_seq and _c.shape[1] == a.sweeps:
    flat = None
    host = np.ascontiguousarray(_c[:n_seq])
elif a.procs > 1:
    with mp.get_context("fork").Pool(min(a.procs, os.cpu_count() or 1)) as pool:   # before the HIP runtime loads
        flat = pool.map(_w, jobs, chunksize=2)
else:
    flat = [_w(j) for j in jobs]
if flat is not None:
    host = np.stack(flat).reshape(n_seq, a.sweeps, -1, 4)
    if a.cache:
        np.save(a.cache, host)
print("# %d sequences x %d sweeps synthesised in %.1f s" % (n_seq, a.sweeps, time.perf_counter() - t0), flush=True)
vl = conftest.load_pkg()
import torch  # noqa: E402

d = torch.from_numpy(host).cuda()
npts = host.shape[2]
order, pos, step = [], -1, 1
for _ in range(a.warm + a.steps):
    if pos + step < 0 or pos + step > a.sweeps - 1:
        step = -step
    pos += step
    order.append(pos)


def ptr(seq, k):
    return d.data_ptr() + ((seq % n_seq) * a.sweeps + order[k]) * npts * 16


def run(H, B, table):
    hs = [vl.Handle(0, n_sessions=B, with_mapping=0 if a.no_mapping else 1, max_frames=a.warm + a.steps + 8) for _ in range(H)]

    def step(i, k):
        h = hs[i]
        if B == 1:
            h.process_scan_device(ptr(i, k), npts)
        else:
            h.batch_process_scan_device([ptr(i * B + b, k) for b in range(B)], [npts] * B)

    def drive(lo, hi):
        if a.threads and H > 1:
            def work(i):
                for k in range(lo, hi):
                    step(i, k)
            ts = [threading.Thread(target=work, args=(i,)) for i in range(H)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        else:
            for k in range(lo, hi):
                for i in range(H):
                    step(i, k)
        for h in hs:
            h.sync()

    drive(0, a.warm)
    if table:
        for h in hs:
            h.profile_kernel("*", 48 * a.steps + 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drive(a.warm, a.warm + a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = {}
    if table:
        for h in hs:
            for name, (ms, cnt) in h.profile_table().items():
                o = rows.get(name, (0.0, 0))
                rows[name] = (o[0] + ms, o[1] + cnt)
    fin = all(bool(np.isfinite(h.select(b).trajectory()).all()) for h in hs for b in range(B))
    for h in hs:
        h.close()
    return dt, rows, fin


for H, B in configs:
    dt, _, fin = run(H, B, False)
    print("H x B = %d x %2d (%2d sessions%s): %8.0f scans/s   %7.1f us per step of a handle   finite=%s"
          % (H, B, H * B, ", threads" if a.threads and H > 1 else "", H * B * a.steps / dt, 1e6 * dt / a.steps, fin), flush=True)
    if a.table:
        dt2, rows, _ = run(H, B, True)
        tot = sum(ms for ms, _ in rows.values()) or 1.0
        print("   (with event pairs around every launch: %.0f scans/s)" % (H * B * a.steps / dt2))
        for name, (ms, cnt) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
            print("   %-22s %7.1f us x %5.2f per step  (%4.1f %% of the summed kernel time)" % (name, 1e3 * ms / max(cnt, 1), cnt / (H * a.steps), 100 * ms / tot))
