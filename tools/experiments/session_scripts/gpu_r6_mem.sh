#!/bin/bash
# round-4 (session 2) memory-path check: kernel tests of the memory ops + kernel-trace durations of tools/mem_probe.py. usage: gpu_r6_mem.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "memory or upsample_ce" 2>&1 | tail -6
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python tools/mem_probe.py > $O/kt.log 2>&1
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/mem_probe_kernel_stats.csv
python - <<PY
import csv
for r in csv.reader(open('$O/mem_probe_kernel_stats.csv')):
    if any(k in r[0] for k in ('mem_', 'ce_', 'reduce_partials')): print(r[0][:58].ljust(58), r[1], r[3], r[4], r[5])
PY
grep -v "^[WEI]2026" $O/kt.log | grep mem_
find $O -name '*.db' -delete
