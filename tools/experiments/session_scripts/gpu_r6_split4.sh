#!/bin/bash
# round 6: the 256 x 128 tile of the split path: kernel tests, then the bench line with it off / by the cost model / everywhere
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino" > $O/pytest_conv.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_conv.log
PM_SPLIT_BM256=2 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino" > $O/pytest_conv256.log 2>&1; echo "pytest(256 everywhere) exit $?"; tail -3 $O/pytest_conv256.log
run() { tag=$1; shift; env "$@" PM_PROFILE_DUMP=$O/shapes_$tag.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
run m1 PM_SPLIT_BM256=1
run m0 PM_SPLIT_BM256=0
run m2 PM_SPLIT_BM256=2
run m1c6 PM_SPLIT_BM256=1 PM_SPLIT_256_COST=0.6
run m1b PM_SPLIT_BM256=1
