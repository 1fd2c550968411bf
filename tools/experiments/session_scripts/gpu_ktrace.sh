#!/bin/bash
# kernel trace of the timed steps only (no profile steps): per-kernel table. usage: gpu_ktrace.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile "$@" > $O/kt.log 2>&1
grep '^{' $O/kt.log | cut -c1-200
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/kernel_stats.csv 5
find $O -name '*.db' -delete
