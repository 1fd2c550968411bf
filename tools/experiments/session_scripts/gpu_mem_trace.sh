#!/bin/bash
# kernel-trace durations (GPU time, no host overhead) of the memory-path probe. usage: gpu_mem_trace.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python tools/mem_probe.py > $O/kt.log 2>&1
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/mem_probe_kernel_stats.csv
grep -e mem_ -e ce_ -e pm_copy $O/mem_probe_kernel_stats.csv | cut -d, -f1-6 | cut -c1-60,100-400
find $O -name '*.db' -delete
