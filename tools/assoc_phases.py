#!/usr/bin/env python3
"""k_map_assoc: where a wavefront's cycles go, per outer round (debug handle: phase stamps summed over the wavefronts of every launch since
the handle was created).  Phases: setup (point, pose, box) | pieces | blocks + flatten | records / cache entries | arg-min rounds | tail.
Also: how many second-round queries re-rank the first round's candidates and how many search again.  Needs an MI355X."""
import os
import sys

import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

vl = conftest.load_pkg()
synth = conftest.load_synth()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=n + 1)
h = vl.Handle(0, with_mapping=1, debug=1, max_frames=n + 8)
for k in range(n):
    h.process_scan(seq.sweep(k))
h.sync()
cyc = h.debug_raw(2, 71, np.int64)[:16].reshape(2, 8)
names = ["setup", "pieces", "blocks+flatten", "records", "arg-min", "tail"]
for r in range(2):
    w = max(int(cyc[r, 7]), 1)
    print("round %d: %8d wavefronts | " % (r, w) + "  ".join("%s %6.0f" % (names[k], cyc[r, k] / w) for k in range(6)) + "  | total %6.0f cycles per wavefront" % (cyc[r, :6].sum() / w))
st = h.map_state()
b = h.debug_raw(2, 73, np.int32).reshape(-1, 8)
kcap = 8192   # kStackCapCorner
for kind, lo, cnt in (("corner", 0, st["n_corner_stack"]), ("surf", kcap, st["n_surf_stack"])):
    t, tot = b[lo:lo + cnt, 6], b[lo:lo + cnt, 7]
    print("%-6s %5d stack points of the last sweep: candidates per query median %d, 99 %% %d, max %d; cached for the second round %d, not cacheable %d" %
          (kind, cnt, int(np.median(tot)), int(np.percentile(tot, 99)), int(tot.max()), int((t >= 0).sum()), int((t < 0).sum())))
