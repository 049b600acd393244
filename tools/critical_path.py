#!/usr/bin/env python3
"""What bounds the sweep period: from a rocprofv3 --kernel-trace results .db of a streaming run, the stage spans of every sweep
(scan registration + NN grids | scan-feature VoxelGrid | laser odometry | laser mapping) and, for each stage start, how long after the
end of each thing it depends on it started (its own stream's previous sweep, the producing stage of the same sweep).  A stage that
starts right behind its own previous sweep is the stream that bounds the period; one that starts right behind its producer is waiting
for data.   python tools/critical_path.py <rocprof output dir>"""
import collections
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith('kernels')][0]
rows = c.execute("select name, start, end, stream_id from %s order by start" % kt).fetchall()


def short(n):
    return n.split('(')[0].replace('vloam::', '').replace('void ', '').split('<')[0]


FIRST = {'sr': 'k_sr_first_last', 'ds': 'k_map_ds_bin', 'lo': 'k_lo_assoc', 'map': 'k_map_prepare'}
LAST = {'sr': 'k_lo_grid_scatter', 'ds': 'k_map_ds_reduce', 'lo': 'k_lm_solve', 'map': 'k_map_finalize'}
ev = collections.defaultdict(list)   # stage -> list of [start, end, busy]
by_stream = collections.defaultdict(list)
for n, s, e, st in rows:
    by_stream[st].append((short(n), s, e))
stage_of_stream = {}
for st, L in by_stream.items():
    names = set(x[0] for x in L)
    for stage, f in FIRST.items():
        if f in names and (stage != 'lo' or 'k_sr_ring' not in names):
            stage_of_stream[st] = stage
for st, stage in stage_of_stream.items():
    L = by_stream[st]
    cur = None
    n_first = 0
    for name, s, e in L:
        if name == FIRST[stage] and (stage != 'lo' or n_first % 2 == 0):
            cur = [s, e, 0.0]
            ev[stage].append(cur)
        if name == FIRST[stage]:
            n_first += 1
        if cur is not None:
            cur[1] = max(cur[1], e)
            cur[2] += e - s
n = min(len(v) for v in ev.values())
print('stages found:', {k: len(v) for k, v in ev.items()}, '-> using the last half of', n, 'sweeps (stage lists aligned at their ends)')
al = {k: v[len(v) - n:] for k, v in ev.items()}
lo_i, hi_i = n // 2, n - 1


def med(x):
    x = sorted(x)
    return x[len(x) // 2] if x else float('nan')


for stage in ('sr', 'ds', 'lo', 'map'):
    if stage not in al:
        continue
    v = al[stage]
    span = [(v[i][1] - v[i][0]) / 1e3 for i in range(lo_i, hi_i)]
    busy = [v[i][2] / 1e3 for i in range(lo_i, hi_i)]
    period = [(v[i + 1][0] - v[i][0]) / 1e3 for i in range(lo_i, hi_i)]
    own = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(lo_i, hi_i)]
    line = '%-4s span %6.1f us  busy %6.1f  period %6.1f  idle before next sweep %6.1f' % (stage, med(span), med(busy), med(period), med(own))
    deps = {'ds': ['sr'], 'lo': ['sr'], 'map': ['lo', 'ds']}.get(stage, [])
    for d in deps:
        if d in al:
            # the producer's stage list may be one sweep ahead / behind: report the smallest non-negative lag over small shifts
            best = None
            for sh in (-2, -1, 0, 1, 2):
                lag = [(v[i][0] - al[d][i + sh][1]) / 1e3 for i in range(lo_i, hi_i) if 0 <= i + sh < n]
                m = med(lag)
                if m >= -1.0 and (best is None or m < best[0]):
                    best = (m, sh)
            if best:
                line += '  | starts %6.1f us after %s(k%+d) ended' % (best[0], d, best[1])
    print(line)

if len(sys.argv) > 2 and sys.argv[2] == '--dump':
    # every kernel of three steady-state periods, in start order: t (us, from the first), duration, stream, name
    sr = al['sr']
    t_lo, t_hi = sr[lo_i + 4][0], sr[lo_i + 7][0]
    sid = {st: i for i, st in enumerate(sorted(by_stream))}
    print('--- kernels between the starts of scan registration of sweeps %d and %d' % (lo_i + 4, lo_i + 7))
    for n_, s_, e_, st_ in rows:
        if t_lo <= s_ < t_hi:
            print('%9.1f %7.1f  s%d %s%s' % ((s_ - t_lo) / 1e3, (e_ - s_) / 1e3, sid[st_], '      ' * sid[st_], short(n_)))
