#!/usr/bin/env python3
"""vloam_process_scan from host memory on its own: the same 64 x 2048 sweeps as device pointers, from pinned memory and from pageable memory
(a buffer of its own per sweep), mapping on, after a short lead-in.  Prints scans/s of each and the ratios (bench.py's host_input leg is the
measurement of record; this is the quick A/B behind it: VLOAM_STAGE_INLINE of c_api.cpp).  --external-ring N adds the experiment the library's
deferred ring came from: the CALLER copies pinned sweeps on a stream of its own into N device buffers and hands device pointers over, one
sweep behind (extring), not deferred (extring_nodefer), and with the copy stream created AFTER the handle's streams (extring_late: slow —
which is why vloam_create creates and uses its copy stream first).
  python tools/host_input_probe.py [--lead 60] [--sweeps 300] [--reps 2]"""
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
ap.add_argument("--lead", type=int, default=60)
ap.add_argument("--sweeps", type=int, default=300)
ap.add_argument("--resident", type=int, default=40)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--procs", type=int, default=32)
ap.add_argument("--cache", default="")
ap.add_argument("--only", default="", help="device | pinned | pageable: time this source only")
ap.add_argument("--external-ring", type=int, default=0, help="also time pinned sweeps copied by the CALLER on a stream of its own into a ring of this many device buffers, "
                "the copy of sweep k + 1 in flight while sweep k is enqueued (host-side event waits only, no cross-stream wait), then vloam_process_scan_device")
a = ap.parse_args()
synth = conftest.load_synth()
_SEQ = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=a.resident + 1)


def _w(k):
    return _SEQ.sweep(k)


if a.cache and os.path.exists(a.cache):
    host = np.load(a.cache)
else:
    with mp.get_context("fork").Pool(min(a.procs, os.cpu_count() or 1)) as pool:   # before the HIP runtime loads
        host = np.stack(pool.map(_w, range(a.resident), chunksize=1))
    if a.cache:
        np.save(a.cache, host)
vl = conftest.load_pkg()
import torch  # noqa: E402

dev = torch.from_numpy(host).cuda()
pinned = torch.from_numpy(host).pin_memory()
npts = host.shape[1]
order, pos, step = [], -1, 1
for _ in range(a.lead + a.sweeps):
    if pos + step < 0 or pos + step > a.resident - 1:
        step = -step
    pos += step
    order.append(pos)
res = {}
KINDS = ("device", "pinned", "pageable") + (("extring", "extring_nodefer", "extring_late") if a.external_ring else ())
R = max(a.external_ring, 2)
ring = torch.empty((R, npts, 4), dtype=torch.float32, device="cuda") if a.external_ring else None
cs = torch.cuda.Stream() if a.external_ring else None
evs = [torch.cuda.Event() for _ in range(R)] if a.external_ring else None
for rep in range(a.reps):
    for how in (KINDS if not a.only else (a.only,)):
        h = vl.Handle(0, with_mapping=1, max_frames=a.lead + a.sweeps + 8)
        for k in order[:a.lead]:
            h.process_scan_device(dev.data_ptr() + k * npts * 16, npts)
        h.sync()
        if how == "extring_late":   # a copy stream created AFTER the handle's streams (the order in which the library creates its own)
            cs_keep, cs = cs, torch.cuda.Stream()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j, k in enumerate(order[a.lead:]):
            if how.startswith("extring"):
                slot = j % R
                with torch.cuda.stream(cs):
                    ring[slot].copy_(pinned[k], non_blocking=True)
                    evs[slot].record(cs)
                if how == "extring_nodefer":
                    evs[slot].synchronize()
                    h.process_scan_device(ring[slot].data_ptr(), npts)
                elif j >= 1:
                    evs[(j - 1) % R].synchronize()
                    h.process_scan_device(ring[(j - 1) % R].data_ptr(), npts)
                continue
            if how == "device":
                h.process_scan_device(dev.data_ptr() + k * npts * 16, npts)
            elif how == "pinned":
                h.process_scan_host_ptr(pinned.data_ptr() + k * npts * 16, npts)
            else:
                h.process_scan(host[k])
        if how in ("extring", "extring_late"):
            evs[(a.sweeps - 1) % R].synchronize()
            h.process_scan_device(ring[(a.sweeps - 1) % R].data_ptr(), npts)
        h.sync()
        dt = time.perf_counter() - t0
        res.setdefault(how, []).append((a.sweeps / dt, h.trajectory()[-1].copy()))
        if how == "extring_late":
            cs = cs_keep
        h.close()
d = max(v for v, _ in res[a.only or "device"])
for how in (KINDS if not a.only else (a.only,)):
    best = max(v for v, _ in res[how])
    print("%-9s %s scans/s   best %.0f = %.3f x device   same last pose: %s" % (how, " ".join("%6.0f" % v for v, _ in res[how]), best, best / d,
                                                                                 all(np.array_equal(t, res[a.only or "device"][0][1]) for _, t in res[how])), flush=True)
