#!/bin/bash
# alternated bench runs under several environment settings on one box: gpu_env_abn.sh <reps> "<VAR=v VAR2=v2>" "<...>" ...  ("-" = default)
cd "$GRAFT_REPO_ROOT" || exit 1
REPS=$1; shift
for rep in $(seq $REPS); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
    env $EV timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-52s' % '$E', d['ms_per_step'])"
  done
done
