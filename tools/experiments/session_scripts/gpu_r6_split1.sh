#!/bin/bash
# round 6, first look at the split-operand fp32 path (PREC 5): convolution kernel tests with the route on, then the default bench line with it on / off, and the per-shape table
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino" > $O/pytest_conv.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_conv.log
PM_PROFILE_DUMP=$O/shapes_split.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_split.json 2> $O/bench_split.err; echo "bench split exit $?"; cut -c1-400 $O/bench_split.json
PM_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_nosplit.json 2> $O/bench_nosplit.err; echo "bench nosplit exit $?"; cut -c1-400 $O/bench_nosplit.json
python tools/conv_shapes.py $O/shapes_split.txt > $O/conv_shapes_split.txt 2>&1; head -40 $O/conv_shapes_split.txt
