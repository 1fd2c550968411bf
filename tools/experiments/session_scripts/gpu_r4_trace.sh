#!/bin/bash
# kernel trace of the bench in one dtype (overlapped as timed): usage gpu_r4_trace.sh <tag> <dtype> [extra env assignments...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; DT=$2; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace -d $O/kt_$DT -- python bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/kt_$DT.log 2>&1
python tools/rocpd_stats.py $(find $O/kt_$DT -name '*.db' | head -1) $O/kernel_stats_$DT.csv 4 | head -40
find $O -name '*.db' -delete
