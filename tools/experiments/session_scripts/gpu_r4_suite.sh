#!/bin/bash
# round 4: the whole GPU suite + the side workloads. usage: gpu_r4_suite.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -E "^\[|passed|failed|error" $O/pytest.log | tail -40
timeout 600 python bench.py --workload mldg --steps 5 --warmup 2 > $O/bench_mldg.log 2>&1; grep '^{' $O/bench_mldg.log > $O/bench_mldg.json; cut -c1-600 $O/bench_mldg.json
timeout 600 python bench.py --workload mldg --dtype bf16 --steps 5 --warmup 2 > $O/bench_mldg_bf16.log 2>&1; grep '^{' $O/bench_mldg_bf16.log > $O/bench_mldg_bf16.json; cut -c1-400 $O/bench_mldg_bf16.json
