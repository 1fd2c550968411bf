#!/bin/bash
# round 4: LDS bank-conflict counters of the bf16 convolution kernels (probe shapes, LDS-DMA kernel everywhere)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
export PROBE_CONV16=2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/lds -- python tools/conv16_probe.py > $O/lds.log 2>&1
tail -3 $O/lds.log
f=$(find $O/lds -name '*counter_collection.csv' | head -1); echo $f
python - "$f" <<'PY'
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys())[:14])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:8]:
    act = c.get('SQ_LDS_IDX_ACTIVE', 0) or 1
    print('%-60s' % k, {n: ('%.3g' % v) for n, v in c.items()}, 'conflict/active = %.3f' % (c.get('SQ_LDS_BANK_CONFLICT', 0) / act))
PY
find $O/lds -name '*.csv' -size +5M -delete
