#!/bin/bash
# wgrad16.hip: parity tests, then the captured bf16 step with and without it (PM_WGRAD16=0) on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad16 or conv_bf16_activations or bf16" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for i in 1 2 3; do
  PM_WGRAD16=0 timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('PM_WGRAD16=0 (graph)', j['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default      (graph)', j['ms_per_step'])"
done
