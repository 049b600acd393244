import sqlite3, sys, glob, collections
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith('kernels')][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kt)]
rows = c.execute("select name, start, end, stream_id from %s order by start" % kt).fetchall()
# per stream: gaps between consecutive kernels
by = collections.defaultdict(list)
for n, s, e, st in rows: by[st].append((n.split('(')[0][:28], s, e))
for st, L in by.items():
    L = L[len(L)//2:]   # steady state half
    gaps = collections.defaultdict(list)
    for a, b in zip(L, L[1:]):
        gaps[(a[0], b[0])].append((b[1] - a[2]) / 1e3)
    print('stream', st, 'n', len(L))
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v.sort()
        print('   %-28s -> %-28s n=%4d med gap %7.2f us  mean %7.2f' % (k[0], k[1], len(v), v[len(v)//2], sum(v)/len(v)))
    # period of the first kernel name
    n0 = L[0][0]
    st_ = [x[1] for x in L if x[0] == n0]
    if len(st_) > 3:
        d = sorted((b - a) / 1e3 for a, b in zip(st_, st_[1:]))
        print('   period of', n0, 'median %.1f us' % d[len(d)//2])
print('--- LO stream detail')
for st, L in by.items():
    if not any('k_lo_assoc' in x[0] for x in L) or any('k_sr_ring' in x[0] for x in L): continue
    L = L[len(L)//2:]
    g = [(b[1]-a[2])/1e3 for a, b in zip(L, L[1:]) if 'k_lm' in a[0] and 'assoc' in b[0]]
    print(st, ' '.join('%.1f' % x for x in g[:40]))
    d = [((x[2]-x[1])/1e3) for x in L]
    print('   durations', ' '.join('%.1f' % x for x in d[:24]))
    break
