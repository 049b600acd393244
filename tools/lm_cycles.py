#!/usr/bin/env python3
"""Where a k_lm_solve launch spends its shader cycles (LMRecord.cyc: factor loops / evaluations incl. reductions and grid barriers /
serial trust-region bookkeeping / whole kernel), for the four solves of a steady-state sweep.  Needs an MI355X."""
import os
import sys

import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402

vl = conftest.load_pkg()
synth = conftest.load_synth()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seq = synth.SynthSequence(n_rings=64, n_azimuth=2048, n_sweeps=n + 1)
h = vl.Handle(0, with_mapping=1, max_frames=n + 8)
for k in range(n):
    h.process_scan(seq.sweep(k))
h.sync()
for name, st, item in (("LO round 0", 1, 2), ("LO round 1", 1, 18), ("map round 0", 2, 3), ("map round 1", 2, 19)):
    r = h.debug_lm_record(st, item)
    c = r["cyc"]
    print("%-12s factors %5d  iterations %d  evaluations %d  | cycles: factor loops %7.0f  evaluations %7.0f  serial %7.0f  whole solve %7.0f  (other %7.0f)" %
          (name, r["n_factors"], int(r["trace"].shape[0]), r["n_evals"], c[0], c[1], c[2], c[3], c[3] - c[1] - c[2]))
if os.environ.get("VLOAM_LM_STAMPS_BUILD"):   # library built with -DVLOAM_LM_STAMPS: thread 0's serial sections, summed over the iterations
    from vloam_amd import LMRecord, K_LM_MAX_TRACE
    for name, st, item in (("LO round 0", 1, 2), ("map round 0", 2, 3)):
        raw = h.debug_raw(st, item, np.uint8)
        rec = LMRecord.from_buffer_copy(raw.tobytes())
        row = np.ctypeslib.as_array(rec.trace).reshape(K_LM_MAX_TRACE, 8)[100]
        print("%-12s serial sections (cycles, all iterations): bookkeeping+LDS reads %6.0f  1/radius+diagonal %6.0f  Cholesky %6.0f  model cost+step %6.0f  plus %6.0f | acceptance %6.0f" %
              (name, row[1], row[2], row[3], row[4], row[5], row[7]))
        print("%-12s first evaluation: cache fill %6.0f cycles" % (name, row[6]))
        ev = np.ctypeslib.as_array(rec.trace).reshape(K_LM_MAX_TRACE, 8)[101]
        print("%-12s evaluation phases of thread 0 (cycles, evaluations after the first): factor loops %6.0f  LDS stores %6.0f  barrier %6.0f  column sums %6.0f  barrier %6.0f  fold + publish %6.0f  "
              "poll %6.0f  fold + barrier %6.0f" % (name, ev[0], ev[1], ev[2], ev[3], ev[4], ev[5], ev[6], ev[7]))
h.close()
