#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results database (rocpd sqlite) into the per-kernel stats table rocprofv3 --stats
prints (name, calls, total, average, min, max, percentage). Usage: rocpd_stats.py results.db out.csv [steps]"""
import csv
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage', 'MsPerStep'])
        for r in rows:
            name = re.sub(r'\(anonymous namespace\)::', '', r[0])
            w.writerow([name, r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3), round(r[2] / 1e6 / steps, 4)])
    print('wrote %s: %d kernels, %.3f ms total (%.3f ms/step)' % (out, len(rows), tot / 1e6, tot / 1e6 / steps))


if __name__ == '__main__':
    main()
