#!/usr/bin/env python3
"""Compare two pm_profile_dump CSVs (PM_PROFILE_DUMP) shape by shape: usage conv_compare.py a.csv b.csv [steps]"""
import csv, sys
def load(p):
    agg = {}
    for r in csv.DictReader(open(p)):
        key = tuple(int(r[k]) for k in ('mode', 'M', 'N', 'K', 'batch'))
        a = agg.setdefault(key, [0, 0.0, ''])
        a[0] += 1; a[1] += float(r['ms']); a[2] = '%sx%s ks%s nst%s' % (r['bm'], r['bn'], r['ksplit'], r['nst'])
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
rows = []
for k in a:
    if k in b:
        rows.append(((a[k][1] - b[k][1]) / steps, k, a[k], b[k]))
rows.sort()
print('total A %.3f  B %.3f ms/step' % (sum(v[1] for v in a.values()) / steps, sum(v[1] for v in b.values()) / steps))
for d, k, x, y in rows[:12] + rows[-25:]:
    print('%+7.3f ms/step  mode %d M %7d N %5d K %6d batch %2d   A %-22s %7.3f   B %-22s %7.3f' % (d, k[0], k[1], k[2], k[3], k[4], x[2], x[1] / steps, y[2], y[1] / steps))
