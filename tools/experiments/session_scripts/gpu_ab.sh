#!/bin/bash
# same-box A/B of ab/libpinmem_base.so vs the in-tree build: alternating short bench runs. usage: gpu_ab.sh <tag> [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
for i in $(seq 1 ${2:-3}); do
  for v in base new; do
    if [ $v = base ]; then export PM_LIB=$PWD/ab/libpinmem_base.so; else unset PM_LIB; fi
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'])" | tee -a $O/ab.log
  done
done
