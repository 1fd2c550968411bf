#!/bin/bash
# round 6: kernel trace of the fp32 default step, serialised (one stream) and overlapped -> per-kernel table. usage: gpu_r6_ktrace.sh <tag> [f32|bf16]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
DT=${2:-f32}
PM_OVERLAP_WGRAD=0 PM_COMMIT_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts_$DT -- python bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kts_$DT.log 2>&1
python tools/rocpd_stats.py $(find $O/kts_$DT -name '*.db' | head -1) $O/kernel_stats_serialised_$DT.csv 4 | head -3
timeout 600 rocprofv3 --kernel-trace -d $O/kt_$DT -- python bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kt_$DT.log 2>&1
python tools/rocpd_stats.py $(find $O/kt_$DT -name '*.db' | head -1) $O/kernel_stats_$DT.csv 4 | head -3
find $O -name '*.db' -delete
