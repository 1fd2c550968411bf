#!/bin/bash
# alternated bench runs of several library builds on one box: gpu_lib_abn.sh <reps> <lib.so>...
cd "$GRAFT_REPO_ROOT" || exit 1
REPS=$1; shift
for rep in $(seq $REPS); do
  for L in "$@"; do
    PM_LIB=$PWD/$L timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-32s' % '$L', d['ms_per_step'])"
  done
done
