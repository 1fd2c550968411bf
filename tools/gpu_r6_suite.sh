#!/bin/bash
# round 6: what the driver runs at round end -- the full -m gpu suite and smoke() -- with durations. usage: gpu_r6_suite.sh <tag>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -40 $O/pytest.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
