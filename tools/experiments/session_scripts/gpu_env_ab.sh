#!/bin/bash
# same-box A/B of one environment knob: gpu_env_ab.sh <tag> <VAR> <val_a> <val_b> [rounds] [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
for i in $(seq 1 ${5:-3}); do
  for v in $3 $4; do
    env $2=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile $6 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$2=$v', d['ms_per_step'])" | tee -a $O/ab.log
  done
done
