"""Average / minimum duration of the kernels whose name contains PATTERN in a rocprofv3 --kernel-trace output directory. usage: python tools/kernel_avg.py DIR PATTERN [label]"""
import sqlite3, glob, sys
d, pat = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else d
db = glob.glob(d + '/**/*.db', recursive=True)
if not db:
    print(label, 'no .db under', d)
    sys.exit(0)
c = sqlite3.connect(db[0])
for name, n, avg, mn in c.execute("select name, count(*), avg(end - start), min(end - start) from kernels group by name"):
    if pat in name:
        print('%s: %s calls %d avg %.1f us min %.1f us' % (label, name[name.find(pat):][:60], n, avg / 1e3, mn / 1e3))
