#!/bin/bash
# kernel trace of serialised steps (PM_OVERLAP_WGRAD=0: one stream, per-kernel durations undisturbed) + HBM counters -> per-kernel GB/s table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PM_OVERLAP_WGRAD=0
O=gpurun_out/$1; mkdir -p $O; shift
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile "$@" > $O/kt.log 2>&1
grep '^{' $O/kt.log | cut -c1-200
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/kernel_stats_serial.csv 4
find $O -name '*.db' -delete
