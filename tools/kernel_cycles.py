#!/usr/bin/env python3
"""In-kernel cycle counters of the two kernels with the largest per-sweep chip cost (debug = 1 builds the stamps in):
k_sr_ring's phases per scan line and k_lo_assoc's per-query cost (closest point / second+third point / stages / candidates).
Needs an MI355X.   python tools/kernel_cycles.py [--sweeps 8] [--azimuth 2048]"""
import argparse
import os
import sys

import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=8)
ap.add_argument("--azimuth", type=int, default=2048)
ap.add_argument("--sequence-sweeps", type=int, default=0, help="length of the synthetic drive the sweeps are taken from (0: sweeps + 1); bench.py's is 408")
ap.add_argument("--first", type=int, default=0, help="first sweep of the drive to process")
ap.add_argument("--no-mapping", action="store_true")
ap.add_argument("--debug-level", type=int, default=2, help="2: phase stamps only (the production kernel's timing); 1: full debug build")
a = ap.parse_args()
vl = conftest.load_pkg()
synth = conftest.load_synth()
seq = synth.SynthSequence(n_rings=64, n_azimuth=a.azimuth, n_sweeps=a.sequence_sweeps or a.sweeps + 1)
h = vl.Handle(0, debug=a.debug_level, with_mapping=0 if a.no_mapping else 1, max_frames=a.sweeps + 8)
clouds = [seq.sweep(a.first + k) for k in range(a.sweeps)]
h.profile_kernel("k_sr_ring", 4096)
for c in clouds:
    h.process_scan(c)
    h.sync()
ms, n = h.profile_read()
print("k_sr_ring by HIP events: %.1f us per launch over %d launches (one sweep at a time)" % (1e3 * ms / max(n, 1), n))


def pct(x, name, unit="cycles"):
    x = np.asarray(x, dtype=np.float64)
    if x.size == 0:
        print("  %-34s (none)" % name)
        return
    print("  %-34s n %5d  mean %9.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f %s" %
          (name, x.size, x.mean(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max(), unit))


cyc = h.debug_raw(0, 11, np.int64).reshape(-1, 8)[:64]
names = ["load ring -> LDS", "gaps / reach (+ debug sort)", "greedy picks, all sectors at once", "picks: boundary rounds (redone sectors)",
         "bbox, voxel ids, run keys", "bitonic sort of runs", "voxel heads + centroids"]
print("k_sr_ring: cycles per phase over the %d scan lines of the last sweep (debug level %d)" % (cyc.shape[0], a.debug_level))
tot = cyc[:, :7].sum(axis=1)
for r in np.argsort(-tot)[:6]:
    m = int(cyc[r, 7])
    print("  slowest: ring %2d  total %7d  phases %s   len %4d  lessFlat candidates %4d  runs %4d  voxels %4d" %
          (r, tot[r], " ".join("%6d" % v for v in cyc[r, :7]), (m >> 32) & 0xffff, (m >> 48) & 0xffff, m & 0xffff, (m >> 16) & 0xffff))
for q, n in enumerate(names):
    pct(cyc[:, q], n)
pct(cyc[:, :7].sum(axis=1), "whole workgroup")

for outer in (0, 1):
    raw = h.debug_raw(1, outer * 16 + 4, np.int64).reshape(-1, 4)
    st = raw[:, 3]
    used = (raw[:, 0] != 0) | (raw[:, 1] != 0)
    print("k_lo_assoc, outer round %d: %d queries" % (outer, int(used.sum())))
    for kind, sl in (("corner", slice(0, vl.K_MAX_SHARP)), ("plane", slice(vl.K_MAX_SHARP, None))):
        r, u = raw[sl], used[sl]
        r = r[u]
        if r.shape[0] == 0:
            continue
        e = (r[:, 3] & 0xff).astype(np.int8)
        s2 = ((r[:, 3] >> 8) & 0xff).astype(np.int8)
        cand = r[:, 3] >> 16
        print(" %s queries" % kind)
        pct(r[:, 0], "closest point")
        pct(r[:, 1], "second / third point")
        pct(cand, "candidates visited (2nd pass)", "points")
        print("  closest-point stage reached:     ", {int(v): int((e == v).sum()) for v in np.unique(e)})
        print("  second/third stage reached:      ", {int(v): int((s2 == v).sum()) for v in np.unique(s2)})
h.close()
