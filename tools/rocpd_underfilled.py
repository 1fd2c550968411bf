#!/usr/bin/env python3
"""Launches that run long on few workgroups: (kernel, grid) pairs with < min_blocks blocks and average duration > min_us.
Usage: rocpd_underfilled.py results.db [min_blocks=1024] [min_us=25]"""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
minb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
wg = [c for c in ('workgroup_x', 'workgroup_y', 'workgroup_z', 'workgroup_size_x', 'workgroup_size_y', 'workgroup_size_z') if c in cols]
rows = cur.execute("select name, grid_x, grid_y, grid_z, %s, count(*), avg(end-start), sum(end-start) from kernels group by name, grid_x, grid_y, grid_z order by sum(end-start) desc" % ', '.join(wg[:3])).fetchall()
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r[0])[:58]
    gx, gy, gz, wx, wy, wz, n, avg, tot = r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]
    blocks = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
    if blocks < minb and avg / 1e3 > minus:
        print('%-58s blocks %6d  n=%-4d avg %8.1f us  total %8.3f ms' % (name, blocks, n, avg / 1e3, tot / 1e6))
