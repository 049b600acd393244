#!/bin/bash
# Host-side cost of the launch chain: rocprofv3 --hip-trace --stats of the default bench (no extras) -> per-API totals.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-hipapi}; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --hip-trace --stats -d $OUT/hip -- python $GRAFT_REPO_ROOT/bench.py --synth-procs 1 --no-cpu-baseline --no-kernel-timer --no-extras > $OUT/hip.log 2>&1)
DB=$(find $OUT/hip -name "*.db" | head -1)
python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if t.startswith("top") or "summary" in t.lower() or "stats" in t.lower()]
print(cand)
for t in cand:
    try:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
        rows = list(db.execute("select * from %s" % t))[:30]
        if rows: print(t, cols); [print("   ", r) for r in rows]
    except Exception as e:
        print(t, e)
PY
tail -1 $OUT/hip.log | cut -c1-200
find $OUT -name '*.db' -size +20M -delete
