#!/bin/bash
# idle-gap analysis of the timed steps of a plain bench run (kernel trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $O/kt.log 2>&1
DB=$(find $O/kt -name '*.db' | head -1)
grep '^{' $O/kt.log | cut -c1-160
python - $DB <<'PY' | tee $O/gaps.txt
import sqlite3, sys, re, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# timed steps = the last 6 occurrences of nchw_to_nhwc_kernel mark step starts
marks = [r[1] for r in rows if 'nchw_to_nhwc' in r[0]]
t0, t1 = marks[-5], marks[-1]          # four whole steps
sel = [r for r in rows if t0 <= r[1] < t1]
busy, cur_end, gaps = 0, sel[0][1], []
prev = sel[0]
for n, s, e in sel:
    if s > cur_end:
        gaps.append((s - cur_end, prev[0][:40], n[:40]))
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev = (n, s, e)
span = t1 - t0
print('4 steps: span %.2f ms/step, busy %.2f ms/step, idle %.2f ms/step in %d gaps/step, %d kernels/step' % (span/4e6, busy/4e6, (span-busy)/4e6, len(gaps)//4, len(sel)//4))
h = collections.Counter(); c = collections.Counter()
for g, a, b in gaps:
    k = '<2us' if g < 2e3 else '<5us' if g < 5e3 else '<10us' if g < 1e4 else '<30us' if g < 3e4 else '>=30us'
    h[k] += g / 4e6; c[k] += 1
print({k: (round(v, 3), c[k] // 4) for k, v in h.items()}, '(ms/step, gaps/step) by gap size')
pair = collections.Counter()
for g, a, b in gaps:
    pair[(re.sub(r'\(anonymous namespace\)::|void ', '', a)[:28], re.sub(r'\(anonymous namespace\)::|void ', '', b)[:28])] += g / 4e3
for (a, b), v in pair.most_common(14):
    print('%8.1f us/step  after [%s] before [%s]' % (v, a, b))
PY
find $O/kt -name '*.db' -delete
