#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16 or conv16 or pw16" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python -m pytest tests/test_model_parity.py -x -q -m gpu -k "bf16" > $O/pytest_model.log 2>&1; tail -4 $O/pytest_model.log
echo "== default routing"; PROBE_CONV16=1 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_route1.txt
for i in 1 2 3; do
  PM_C16W=0 timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('PM_C16W=0 (graph)', j['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default  (graph)', j['ms_per_step'])"
done
