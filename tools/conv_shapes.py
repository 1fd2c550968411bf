#!/usr/bin/env python3
"""Aggregate a pm_profile_dump CSV (PM_PROFILE_DUMP=path python bench.py) by launch shape: time, TFLOP/s and the time that a
135 TFLOP/s kernel would save. Usage: conv_shapes.py dump.csv [steps [rows]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
agg = {}
for r in rows:
    key = tuple(int(r[k]) for k in ('mode', 'bm', 'bn', 'km', 'nst', 'M', 'N', 'K', 'batch', 'ksplit')) + (int(r.get('prec', 0)),)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r['ms']); a[2] += float(r['gflop'])
tot = sum(a[1] for a in agg.values()) / steps
print('total %.3f ms/step over %d shapes' % (tot, len(agg)))
out = []
for k, (n, ms, gf) in agg.items():
    tf = gf / ms
    lost = ms - gf / 135.0
    out.append((lost / steps, ms / steps, n / steps, tf, k))
out.sort(reverse=True)
print('%8s %8s %6s %7s  mode bm  bn km nst      M     N      K  batch ksplit prec' % ('lost/st', 'ms/st', 'n/st', 'TF'))
for lost, ms, n, tf, k in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print('%8.3f %8.3f %6.1f %7.1f  %4d %3d %3d %2d %3d %7d %5d %6d %5d %5d %4d' % ((lost, ms, n, tf) + k))
