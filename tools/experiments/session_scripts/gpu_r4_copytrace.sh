#!/bin/bash
# round 4: where do the ~85 blit copies per step come from? memory-copy trace (direction, bytes) of a short bench run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
DT=${2:-bf16}
timeout 600 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $O/mc -- python bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/mc.log 2>&1
f=$(find $O/mc -name '*memory_copy_trace.csv' | head -1); echo $f; head -3 $f
python - "$f" <<'PY'
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), 'copies; columns', list(rows[0].keys()))
c = collections.Counter()
for r in rows:
    c[(r.get('Direction') or r.get('Kind'), r.get('Size') or r.get('Bytes') or '')] += 1
for k, n in sorted(c.items(), key=lambda kv: -kv[1])[:40]: print(n, k)
PY
find $O/mc -name '*kernel_trace.csv' -size +20M -delete
