#!/usr/bin/env python3
"""Where a wavefront of k_map_assoc spends its cycles (debug handle: the kernel stamps its phases): setup (loads + pointAssociateToMap) |
piece tables | block probes + flatten | record probes | arg-min rounds | output.  Needs an MI355X.
  VLOAM_MAP_ASSOC_LANES=16 python tools/map_assoc_cycles.py [--sweeps 40] [--sessions 1]"""
import argparse
import os
import sys

import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=40)
ap.add_argument("--sessions", type=int, default=1)
a = ap.parse_args()
vl = conftest.load_pkg()
synth = conftest.load_synth()
seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=a.sweeps + 1)
B = a.sessions
h = vl.Handle(0, n_sessions=B, debug=1, with_mapping=1, max_frames=a.sweeps + 8)
for k in range(a.sweeps):
    c = seq.sweep(k)
    if B == 1:
        h.process_scan(c)
    else:
        h.batch_process_scan([c] * B)
h.sync()
raw = h.debug_raw(2, 71, np.int64).reshape(2, 8)
names = ["setup", "pieces", "blocks + flatten", "records", "arg-min rounds", "output"]
print("k_map_assoc lanes=%s sessions=%d: average cycles per wavefront and phase (session 0)" % (os.environ.get("VLOAM_MAP_ASSOC_LANES", "default"), B))
for outer in (0, 1):
    n = max(int(raw[outer, 7]), 1)
    print("  outer %d: %d wavefronts: " % (outer, n) + ", ".join("%s %.0f" % (nm, raw[outer, k] / n) for k, nm in enumerate(names)) + "  | total %.0f" % (raw[outer, :6].sum() / n))
h.close()
