#!/usr/bin/env python3
"""From a pm_profile_dump CSV (launch order): the forward GEMMs of the training forward vs the same shapes in the eval-mode commit forward.
A forward starts at the stem launch (N = 64, K = 196); launches up to the first dgrad / wgrad belong to the training forward, those after the
step's second stem launch to the commit forward. Usage: fwd_train_vs_eval.py dump.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
stems = [i for i, r in enumerate(rows) if r['mode'] == '0' and r['N'] == '64' and r['K'] == '196']
train, evalf = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(stems[0::2], stems[1::2]):
    nxt = min([s for s in stems if s > b] + [len(rows)])
    i = a
    while i < b and rows[i]['mode'] == '0':
        r = rows[i]
        train[(r['bm'], r['bn'], r['M'], r['N'], r['K'], r['batch'])].append(float(r['ms']))
        i += 1
    for r in rows[b:nxt]:
        if r['mode'] == '0':
            evalf[(r['bm'], r['bn'], r['M'], r['N'], r['K'], r['batch'])].append(float(r['ms']))
nst = len(stems) // 2
tt = te = 0.0
out = []
for k in train:
    if k in evalf:
        t, e = sum(train[k]) / nst, sum(evalf[k]) / nst
        # compare per launch (the training forward has a few launches more: dsn, writenet)
        tl, el = sum(train[k]) / len(train[k]), sum(evalf[k]) / len(evalf[k])
        out.append((tl * len(evalf[k]) / nst - e, k, len(train[k]) / nst, len(evalf[k]) / nst, tl, el))
        tt += t
        te += e
out.sort(reverse=True)
print('training forward GEMMs %.2f ms/step, commit forward GEMMs %.2f ms/step (shapes present in both)' % (tt, te))
print('%9s  %5s %5s %8s %6s %6s %5s   n_train n_eval   train_us  eval_us' % ('extra_ms', 'bm', 'bn', 'M', 'N', 'K', 'batch'))
for d, k, nt, ne, tl, el in out[:30]:
    print('%9.3f  %5s %5s %8s %6s %6s %5s   %5.1f %5.1f   %8.1f %8.1f' % ((d,) + k + (nt, ne, tl * 1e3, el * 1e3)))
