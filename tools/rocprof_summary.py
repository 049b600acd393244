#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (--kernel-trace --stats) into the per-kernel summary committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats  (durations in microseconds)\n")
        if note:
            f.write("# %s\n" % note)
        f.write("%-28s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            short = name.split("(")[0].replace("vloam::", "")
            f.write("%-28s %8d %14.3f %12.3f %8.2f\n" % (short, calls, tot, avg, pct))
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
