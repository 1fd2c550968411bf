#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "pw16 or conv_bf16_activations or conv16_on" > $O/pytest.log 2>&1; tail -25 $O/pytest.log
for r in 1 6 1 6; do
  echo "== PROBE_CONV16=$r (1 default: streaming 1x1 kernel on, 6: off)"; PROBE_CONV16=$r timeout 300 python tools/conv16_probe.py 2>&1 | grep "1x1" | tee -a $O/probe_route$r.txt
done
for i in 1 2; do
  PM_PW16=0 timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('PM_PW16=0', j['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default ', j['ms_per_step'])"
done
