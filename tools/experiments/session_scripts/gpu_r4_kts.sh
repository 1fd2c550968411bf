#!/bin/bash
# round 4: serialised kernel statistics of one tier (every launch on one stream) + per-shape convolution dump. usage: gpu_r4_kts.sh <tag> <dtype>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
DT=${2:-bf16}
PM_OVERLAP_WGRAD=0 PM_COMMIT_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts_$DT -- python bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kts_$DT.log 2>&1
python tools/rocpd_stats.py $(find $O/kts_$DT -name '*.db' | head -1) $O/kernel_stats_serialised_$DT.csv 4 | head -2
find $O -name '*.db' -delete
PM_PROFILE_DUMP=$O/prof_dump_$DT.txt timeout 600 python bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-side > $O/bench_$DT.log 2>&1
grep '^{' $O/bench_$DT.log | cut -c1-200
