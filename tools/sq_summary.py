#!/usr/bin/env python3
"""Per-kernel shader occupancy / issue figures from one rocprofv3 --pmc pass over SQ counters (counter_collection.csv).
SQ_WAVE_CYCLES, SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY count quad-cycles summed over all wavefronts
(MI355X_MICROARCH.md: PMC slots).  wave_us = SQ_WAVE_CYCLES * 4 / 2100 per launch — the wave slots x time a launch takes from the chip, the
quantity a chip full of sessions runs out of (profiles/r05_batch_pmc.txt).  (GRBM_GUI_ACTIVE carries a ~250 k-cycle floor per dispatch under
counter collection, so no residency figure is derived from it any more.)
Usage: sq_summary.py counter_collection.csv [out.txt]"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import csrc_sha256

tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0].replace("vloam::", "").replace("void ", "")
    tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[name][row["Counter_Name"]] += 1
lines = ["# csrc_sha256: %s" % csrc_sha256(),
         "%-30s %7s %12s %12s %8s %8s %8s" % ("kernel", "calls", "wave_qcyc", "wave_us", "active%", "wait%", "stall%")]
rows = []
for k in tot:
    n = max(cnt[k].get("SQ_WAVE_CYCLES", 1), 1)
    g = tot[k].get("GRBM_GUI_ACTIVE", 0.0) / n
    w = tot[k].get("SQ_WAVE_CYCLES", 0.0) / n
    a = tot[k].get("SQ_ACTIVE_INST_ANY", 0.0) / n
    wa = tot[k].get("SQ_WAIT_ANY", 0.0) / n
    wi = tot[k].get("SQ_WAIT_INST_ANY", 0.0) / n
    rows.append((w * n, k, n, w, 4 * w / 2100.0, 100 * a / w if w else 0, 100 * wa / w if w else 0, 100 * wi / w if w else 0))
for _, k, n, w, wus, a, wa, wi in sorted(rows, reverse=True):
    lines.append("%-30s %7d %12.0f %12.0f %8.1f %8.1f %8.1f" % (k[:30], n, w, wus, a, wa, wi))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
