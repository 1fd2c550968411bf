#!/usr/bin/env python3
"""GPU idle time in a rocprofv3 kernel trace: union of kernel intervals vs the span, and the largest gaps with their neighbours.
Usage: rocpd_gaps.py results.db [skip_fraction]  (the first skip_fraction of the span is ignored: start-up and warm-up)"""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0, t1 = rows[0][1], max(r[2] for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[1] >= cut]
short = lambda n: re.sub(r'\(anonymous namespace\)::', '', n)[:50]
busy, gaps, cur_end, prev = 0, [], rows[0][1], rows[0]
span0 = rows[0][1]
for r in rows:
    n, s, e = r
    if s > cur_end:
        gaps.append((s - cur_end, short(prev[0]), short(n)))
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end, prev = e, r
span = cur_end - span0
print('span %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps' % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
big = sorted(gaps, reverse=True)[:25]
for g, a, b in big:
    print('%8.1f us   after [%s]   before [%s]' % (g / 1e3, a, b))
import collections
hist = collections.Counter()
for g, a, b in gaps:
    hist['<5us' if g < 5e3 else '<20us' if g < 2e4 else '<100us' if g < 1e5 else '>=100us'] += g
print({k: round(v / 1e6, 3) for k, v in hist.items()}, 'ms of idle by gap size')
