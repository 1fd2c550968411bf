#!/usr/bin/env python3
"""Per-(kernel, grid) launch statistics from a rocprofv3 rocpd database. Usage: rocpd_calls.py results.db name-substring [out.csv]"""
import csv
import re
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    g = [c for c in ('grid_x', 'grid_y', 'grid_z', 'grid_size_x', 'grid_size_y', 'grid_size_z') if c in cols]
    if not g:
        print('columns:', cols)
    gsel = ', '.join(g) if g else "''"
    rows = cur.execute("select name, %s, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels where name like ? group by name, %s order by sum(end-start) desc"
                       % (gsel, gsel), ('%' + pat + '%',)).fetchall()
    out = csv.writer(open(sys.argv[3], 'w', newline='')) if len(sys.argv) > 3 else None
    for r in rows:
        name = re.sub(r'\(anonymous namespace\)::', '', r[0])[:60]
        line = [name] + list(r[1:])
        print('%-60s grid %-22s n=%-4d avg %9.1f us  min %9.1f  max %9.1f  total %9.3f ms' % (name, 'x'.join(str(v) for v in r[1:1 + len(g)]), r[-5], r[-4] / 1e3, r[-3] / 1e3, r[-2] / 1e3, r[-1] / 1e6))
        if out:
            out.writerow(line)


if __name__ == '__main__':
    main()
