#!/bin/bash
# round-5 closing call: what the driver runs at round end (full GPU suite, smoke) + the evidence set of tools/gpu_round5_profiles.sh. usage: gpu_r5_final.sh <tag> <prefix>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
bash tools/gpu_round5_profiles.sh $1 $2
