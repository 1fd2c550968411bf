#!/bin/bash
# round 6: static wave priority on the split path: one shape (timed + SQ counters), then the bench line on / off
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for pr in 0 1; do echo "PM_SPLIT_PRIO=$pr"; PM_SPLIT_PRIO=$pr python tools/one_conv32.py 8 512 48 48 2048 1 0 1 20; PM_SPLIT_PRIO=$pr python tools/one_conv32.py 8 256 192 192 256 3 1 1 20; PM_SPLIT_PRIO=$pr python tools/one_conv32.py 8 64 192 192 256 1 0 1 20; done
run() { tag=$1; shift; env "$@" PM_PROFILE_DUMP=$O/shapes_$tag.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
run prio1 PM_SPLIT_PRIO=1
run prio0 PM_SPLIT_PRIO=0
run prio1b PM_SPLIT_PRIO=1
