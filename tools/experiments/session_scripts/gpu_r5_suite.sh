#!/bin/bash
# full GPU suite + smoke at HEAD, then the captured bf16 step with the weight gradients inline (PM_OVERLAP_WGRAD=0) vs on their side stream
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -16 $O/pytest.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
for i in 1 2; do
  PM_OVERLAP_WGRAD=0 timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('wgrad inline (graph)', j['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default      (graph)', j['ms_per_step'])"
done
