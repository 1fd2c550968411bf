#!/bin/bash
# phase timeline of the step, overlapped (as timed) and serialised. usage: gpu_phases.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile "$@" > $O/kt.log 2>&1
grep '^{' $O/kt.log | cut -c1-200
python tools/step_phases.py $(find $O/kt -name '*.db' | head -1) 3 | tee $O/phases.txt
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/kernel_stats.csv 6 | tail -1
find $O -name '*.db' -delete
