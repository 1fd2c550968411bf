#!/bin/bash
# round 5: tap skipping + planner rule of the wide kernel: kernel tests, per-shape probe (default routing vs no wide kernel), bf16 step. usage: gpu_r5_c16w2.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16 or conv16 or bf16_filter or epilogue_bn_statistics_bf16" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for r in 1 4; do
  echo "== PROBE_CONV16=$r (1: default routing, 4: the same without the wide kernel)"; PROBE_CONV16=$r timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_route$r.txt
done
for i in 1 2; do
  PM_C16W=0 timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('PM_C16W=0', j['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default ', j['ms_per_step'])"
done
