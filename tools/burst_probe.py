#!/usr/bin/env python3
"""How long a BURST of K sweeps takes between two synchronisations, K = 1 .. 80 (the driver's bench line is a burst of 20): the fit
T(K) = L + (K - 1) P separates the pipeline's fill / drain latency L from its steady period P.  Needs an MI355X.
Usage: tools/burst_probe.py [--warm 200] [--reps 5] [--cache /tmp/burst.npy]"""
import argparse
import os
import sys
import time

import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import bench     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ks", default="1,2,3,5,8,12,20,40,80")
    ap.add_argument("--cache", default="")
    ap.add_argument("--detail", type=int, default=0, help="bursts of this many sweeps, --reps of them: every burst's time and the return time of every call")
    a = ap.parse_args()
    ks = [int(s) for s in a.ks.split(",")]
    vl = conftest.load_pkg()
    synth = conftest.load_synth()
    total = a.warm + a.reps * (a.detail if a.detail else sum(ks)) + 8
    if a.cache and os.path.exists(a.cache) and np.load(a.cache, mmap_mode="r").shape[0] >= total:
        host = np.load(a.cache)[:total]
    else:
        seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=total)
        res = bench.synthesise(synth, [("sweep", 0, k) for k in range(total)], [seq], min(os.cpu_count() or 1, 48))
        host = np.stack(res)
        if a.cache:
            np.save(a.cache, host)
    import torch
    d = torch.from_numpy(host).to("cuda:0")
    n_pts = host.shape[1]
    base, stride = d.data_ptr(), n_pts * 16
    h = vl.Handle(0, scan_line=64, with_mapping=1, max_points=n_pts, max_frames=total + 8)
    pos = 0
    enq = {}

    def run(k):
        nonlocal pos
        t0 = time.perf_counter()
        for j in range(k):
            h.process_scan_device(base + (pos + j) * stride, n_pts)
        te = time.perf_counter()
        h.sync()
        t1 = time.perf_counter()
        pos += k
        enq.setdefault(k, []).append(1e6 * (te - t0))
        return 1e6 * (t1 - t0)

    run(a.warm)
    enq.clear()
    if a.detail:
        k = a.detail
        for r in range(a.reps):
            torch.cuda.synchronize()
            st = [time.perf_counter()]
            for j in range(k):
                h.process_scan_device(base + (pos + j) * stride, n_pts)
                st.append(time.perf_counter())
            h.sync()
            st.append(time.perf_counter())
            pos += k
            us = [1e6 * (x - st[0]) for x in st[1:]]
            print("burst %2d: %7.1f us | calls return at " % (r, us[-1]) + " ".join("%.0f" % x for x in us[:-1]))
        h.close()
        return
    rows = {}
    for r in range(a.reps):
        for k in ks:
            torch.cuda.synchronize()
            rows.setdefault(k, []).append(run(k))
    print("# burst of K sweeps between two vloam_sync (us): median of %d, min, T/K, increment per extra sweep since the previous K" % a.reps)
    prev = None
    for k in ks:
        v = np.array(rows[k])
        med = float(np.median(v))
        inc = "" if prev is None else "  +%.1f us / sweep" % ((med - prev[1]) / (k - prev[0]))
        print("K %3d   median %8.1f   min %8.1f   per sweep %7.1f   host enqueue loop %8.1f (%.1f / sweep)%s" % (k, med, v.min(), med / k, float(np.median(enq[k])), float(np.median(enq[k])) / k, inc))
        prev = (k, med)
    h.close()


if __name__ == "__main__":
    main()
