#!/bin/bash
# same-box A/B of two library builds under ab/: gpu_lib_ab.sh <tag> <libA> <libB> [rounds]; also runs the conv kernel tests on libB first
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
PM_LIB=$PWD/ab/$3 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv" 2>&1 | tail -2
for i in $(seq 1 ${4:-3}); do
  for v in $2 $3; do
    PM_LIB=$PWD/ab/$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'])" | tee -a $O/ab.log
  done
done
