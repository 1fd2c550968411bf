#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA_[A-Z_]*\|TCP_[A-Z_]*\|TCC_[A-Z_0-9]*\|SQ_LDS[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_BUSY[A-Z_]*\)\b" | sort -u | tr '\n' ' ' > $O/counters.txt; head -c 3000 $O/counters.txt; echo
for route in 1 3; do
  for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/p_${route}_$tag -- python tools/one_conv.py 8 256 192 192 256 3 1 1 $route 6 > $O/p_${route}_$tag.log 2>&1
    python - <<PY
import sqlite3, glob, sys
db = glob.glob('$O/p_${route}_$tag/**/*.db', recursive=True)
if not db: print('no db for route $route $set'); sys.exit()
c = sqlite3.connect(db[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
try:
    rows = c.execute("select kernel_name, counter_name, sum(value) * 1.0 / count(distinct dispatch_id), count(distinct dispatch_id), avg(end - start) from counters_collection group by kernel_name, counter_name").fetchall()
except Exception as e:
    rows = []
    print('query failed', e, [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()][:8])
for r in rows:
    if 'conv' in r[0]: print('route $route', r[0][:70], r[1], 'per launch %.5g' % r[2], 'launches', r[3], 'avg ns %.0f' % r[4])
PY
    find $O/p_${route}_$tag -name '*.db' -delete
  done
done
