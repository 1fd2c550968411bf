#!/bin/bash
# round 6: knobs of the split path on the default bench line (same box): LDS stages, tile preference
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
run() { tag=$1; shift; env "$@" PM_PROFILE_DUMP=$O/shapes_$tag.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
run base PM_SPLIT=1
run nst2 PM_SPLIT_NST=2
run tile0 PM_SPLIT_TILE=0
run tile0nst2 PM_SPLIT_TILE=0 PM_SPLIT_NST=2
run bm128 PM_WINO_BM=128
