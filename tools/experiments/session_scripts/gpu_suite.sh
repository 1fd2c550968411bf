#!/bin/bash
# GPU suite (verbose, not -x: every failure is listed) + smoke + the default bench line. usage: gpu_suite.sh <tag> [pytest args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
timeout 1700 python -m pytest tests -m gpu -q -rf -s "$@" > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -e '^\[' -e FAILED -e passed -e failed $O/pytest.log | tail -40
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench exit $?"; grep '^{' $O/bench.log > $O/bench_1gpu.json; cut -c1-300 $O/bench_1gpu.json; tail -3 $O/bench.log | cut -c1-400
