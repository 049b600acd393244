#!/usr/bin/env python3
"""Average duration of a kernel's launches by their position inside the sweep (first / second launch of the sweep, ...), from a rocprofv3
--kernel-trace database: the two outer rounds of an association or a solve do different work (search vs candidate cache, 4 vs 3 iterations).
Usage: tools/kernel_rounds.py <results.db> [per_sweep=2] [name substrings ...]"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    want = sys.argv[3:] or ["k_map_assoc", "k_map_fit", "k_lm_solve", "k_lo_assoc"]
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    seq = collections.defaultdict(list)
    for name, st, en in cur.execute(f"select s.display_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id = s.id order by d.start"):
        seq[name.split("(")[0]].append((en - st) / 1000.0)
    for nm, v in sorted(seq.items()):
        if not any(w in nm for w in want):
            continue
        v = v[len(v) // 2 - (len(v) // 2) % per:]   # steady state: the second half, aligned to a sweep
        print("%-44s %5d launches  " % (nm[-44:], len(v)) + "  ".join("round %d: %6.2f us" % (r, sum(v[r::per]) / max(len(v[r::per]), 1)) for r in range(per)))


if __name__ == "__main__":
    main()
