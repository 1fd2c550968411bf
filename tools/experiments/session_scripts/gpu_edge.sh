#!/bin/bash
# Stitch kernel + input edge: the new GPU tests, config5 bench (stitching in it) and the --input-edge side measurement. usage: gpu_edge.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -rf -k "sliding or stitch or prefetcher or prepare_batch or config5" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
timeout 600 python bench.py --workload config5 --steps 5 --warmup 2 > $O/config5.log 2>&1; grep '^{' $O/config5.log | cut -c1-260
timeout 600 python bench.py --no-cpu-baseline --no-profile > $O/plain.log 2>&1; grep '^{' $O/plain.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-profile --input-edge > $O/edge.log 2>&1; grep '^{' $O/edge.log > $O/bench_input_edge.json; cat $O/bench_input_edge.json | cut -c1-1500; tail -3 $O/edge.log | cut -c1-300
