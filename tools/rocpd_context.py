#!/usr/bin/env python3
"""For every launch whose name contains <pattern>: the kernels launched right before and after it (histogram of (prev, next) pairs).
Usage: rocpd_context.py results.db pattern"""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
pat = sys.argv[2]
short = lambda n: re.sub(r'\(anonymous namespace\)::', '', n)[:60]
cnt = collections.Counter()
for i, (n, s, e) in enumerate(rows):
    if pat in n:
        p = short(rows[i - 1][0]) if i else '-'
        q = short(rows[i + 1][0]) if i + 1 < len(rows) else '-'
        cnt[(p, q, (e - s) // 1000)] += 1
for (p, q, us), v in cnt.most_common(30):
    print('%4d x  %3d us   after [%s]   before [%s]' % (v, us, p, q))
