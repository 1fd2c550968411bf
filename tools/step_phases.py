#!/usr/bin/env python3
"""Phase timeline of the agg step from a rocprofv3 --kernel-trace database: train forward | backward | SGD | memory-commit forward, per step,
cut at marker kernels (nchw_to_nhwc = a forward starts, the first CE kernel after it = the losses, sgd_multi_kernel = the optimizer), plus the
per-family kernel time inside each phase. Usage: step_phases.py results.db [steps_to_average]"""
import collections
import re
import sqlite3
import sys


def fam(n):
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)
    for key, f in (('conv_igemm_kernel<2', 'gemm wgrad'), ('conv_igemm_kernel<1', 'gemm dgrad'), ('conv_igemm_kernel<0', 'gemm fwd'), ('wino_', 'wino transforms'),
                   ('bn_bwd', 'bn bwd'), ('pm_bn_bwd', 'bn bwd'), ('bn_', 'bn fwd'), ('pm_bn_', 'bn fwd'), ('ce_', 'losses'), ('mem_', 'memory'), ('splitk', 'splitk reduce'),
                   ('resize', 'resize/pool'), ('maxpool', 'resize/pool'), ('gap_', 'resize/pool'), ('sgd', 'sgd'), ('colsum', 'bias grad')):
        if key in n:
            return f
    return 'other'


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    fw = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r[0]]          # once per step: both forwards read the same converted batch
    sg = [i for i, r in enumerate(rows) if 'sgd_multi' in r[0]]
    steps = []
    for a, c in zip(fw[:-1], fw[1:]):
        s = [i for i in sg if a < i < c]
        if not s:
            continue
        bw = [i for i in range(a, s[0]) if 'ce_bwd_cols' in rows[i][0]]        # the first backward kernels: the losses' row passes
        if not bw:
            continue
        steps.append((a, bw[0], s[0], s[-1], c, c))
    steps = steps[-nsteps:]
    tot = collections.OrderedDict((k, 0.0) for k in ('train forward', 'backward', 'sgd', 'commit forward', 'step'))
    famt = {k: collections.Counter() for k in tot}
    for a, ce, s0, s1, b, c in steps:
        cuts = {'train forward': (rows[a][1], rows[ce][1]), 'backward': (rows[ce][1], rows[s0][1]), 'sgd': (rows[s0][1], rows[s1][2]),
                'commit forward': (rows[s1][2], rows[c][1]), 'step': (rows[a][1], rows[c][1])}
        for k, (t0, t1) in cuts.items():
            tot[k] += (t1 - t0) / 1e6
            for n, st, en in rows[a:c]:
                if t0 <= st < t1:
                    famt[k][fam(n)] += (en - st) / 1e6
    n = float(len(steps))
    for k, v in tot.items():
        print('%-16s %7.2f ms/step   ' % (k, v / n) + '  '.join('%s %.2f' % (f, t / n) for f, t in famt[k].most_common(9)))


if __name__ == '__main__':
    main()
