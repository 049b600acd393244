#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (--kernel-trace --stats) into the per-kernel summary committed under profiles/.

Kernels that only run when a handle is created or for host-side bookkeeping (k_lm_sync_probe: the one-off placement probe of the
solver's sync words, runtime fill / copy kernels, k_map_error_fetch) are listed separately and kept OUT of the percentage column, which
is the share of the per-sweep kernels only."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import csrc_sha256

SETUP = ("k_lm_sync_probe", "__amd_rocclr_", "k_map_error_fetch", "k_map_export", "k_map_register")


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    def short(name):
        return name.split("(")[0].replace("vloam::", "").replace("void ", "")
    sweep = [r for r in rows if not any(k in r[0] for k in SETUP)]
    setup = [r for r in rows if any(k in r[0] for k in SETUP)]
    tot = sum(r[2] for r in sweep) or 1.0
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats  (durations in microseconds; pct = share of the per-sweep kernels' GPU time)\n")
        if note:
            f.write("# %s\n" % note)
        f.write("# csrc_sha256: %s\n" % csrc_sha256())   # bench.py quotes this table only while it times the same kernel sources
        f.write("%-34s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, t, avg, _ in sweep:
            f.write("%-34s %8d %14.3f %12.3f %8.2f\n" % (short(name), calls, t, avg, 100.0 * t / tot))
        f.write("# create-time / bookkeeping kernels (not part of a sweep, not in pct):\n")
        for name, calls, t, avg, _ in setup:
            f.write("# %-32s %8d %14.3f %12.3f\n" % (short(name), calls, t, avg))
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
