#!/bin/bash
# round 6: the persistent producer / consumer split GEMM (gemm_splitp.hip): kernel tests with the route forced / default, then the bench line off / on
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_SPLITP=2 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -s -k "conv or wino or split" > $O/pytest_forced.log 2>&1; echo "pytest (forced) exit $?"; tail -3 $O/pytest_forced.log
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino or split" > $O/pytest_default.log 2>&1; echo "pytest (default) exit $?"; tail -2 $O/pytest_default.log
run() { tag=$1; shift; env "$@" PM_PROFILE_DUMP=$O/shapes_$tag.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])")"; }
run on PM_SPLITP=1
run off PM_SPLITP=0
run on2 PM_SPLITP=1
run units256 PM_SPLITP=1 PM_SPLITP_MIN_UNITS=256
