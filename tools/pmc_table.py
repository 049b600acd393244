#!/usr/bin/env python3
"""Per-kernel averages of whatever counters one rocprofv3 --pmc pass collected (counter_collection.csv): one row per kernel, one column
per counter, averaged over the dispatches; the last column is the number of dispatches.  Several csv files are merged column-wise.
  pmc_table.py out.txt pass1.csv [pass2.csv ...]"""
import collections
import csv
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
cols = []
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("vloam::", "").replace("void ", "")
        c = row["Counter_Name"]
        if c not in cols:
            cols.append(c)
        tot[name][c] += float(row["Counter_Value"])
        cnt[name][c] += 1
lines = ["%-34s" % "kernel" + "".join("%26s" % c[:25] for c in cols) + "%8s" % "calls"]
for k in sorted(tot, key=lambda k: -max(cnt[k].values())):
    lines.append("%-34s" % k[:33] + "".join("%26.2f" % (tot[k][c] / max(cnt[k][c], 1)) for c in cols) + "%8d" % max(cnt[k].values()))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
