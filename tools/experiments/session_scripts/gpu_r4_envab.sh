#!/bin/bash
# round 4: same-box A/B of one environment knob on the bench line. usage: gpu_r4_envab.sh <tag> <VAR> <dtype> [reps]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
VAR=$2; DT=$3; REPS=${4:-2}
for rep in $(seq 1 $REPS); do for F in 0 1; do
  env $VAR=$F timeout 600 python bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-side > $O/bench_${VAR}${F}_$rep.log 2>&1
  grep '^{' $O/bench_${VAR}${F}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$VAR=$F rep $rep ms/step', d['ms_per_step'])"
done; done
