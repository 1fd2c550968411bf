#!/bin/bash
# round 6: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of ONE fp32 forward convolution per shape. usage: gpu_r6_traffic1.sh <tag> ["n cin h w cout k pad dil" ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
shift
[ $# -eq 0 ] && set -- "8 512 48 48 2048 1 0 1" "8 256 192 192 256 3 1 1" "8 2048 48 48 512 1 0 1" "8 1024 48 48 256 1 0 1"
for ARGS in "$@"; do
  echo "== $ARGS"
  for set in FETCH_SIZE WRITE_SIZE; do
    tag=$(echo "$ARGS $set" | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/p_$tag -- python tools/one_conv32.py $ARGS 6 > $O/p_$tag.log 2>&1
    python - <<PY
import sqlite3, glob, sys
db = glob.glob('$O/p_$tag/**/*.db', recursive=True)
if not db: print('no db for $set'); print(open('$O/p_$tag.log').read()[-600:]); sys.exit()
c = sqlite3.connect(db[0])
rows = c.execute("select kernel_name, counter_name, sum(value) * 1.0 / count(distinct dispatch_id), count(distinct dispatch_id), avg(end - start) from counters_collection group by kernel_name, counter_name").fetchall()
for r in rows:
    if 'conv' in r[0] or 'wino' in r[0]:
        mb = r[2] * (2 if r[1] == 'FETCH_SIZE' else 1) * 1024 / 1e6     # KB units; gfx950: FETCH_SIZE counts half (MI355X_MICROARCH.md)
        print('%-70s %s %.1f MB per launch, launches %d, avg us %.1f' % (r[0][:70], r[1], mb, r[3], r[4] / 1e3))
PY
    find $O/p_$tag -name '*.db' -delete
  done
done 2>&1 | tee $O/traffic.log
