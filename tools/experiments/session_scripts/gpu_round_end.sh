#!/bin/bash
# what the driver runs at round end, plus the evidence set: full GPU suite, smoke, default bench, profiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
bash tools/gpu_final_profile.sh $1
