#!/bin/bash
# round 4: same-box A/B of one environment knob over arbitrary values. usage: gpu_r4_envab2.sh <tag> <VAR> <dtype> <reps> <value> [<value> ...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
VAR=$2; DT=$3; REPS=$4; shift 4
for rep in $(seq 1 $REPS); do for F in "$@"; do
  env $VAR=$F PM_PROFILE_DUMP=$O/prof_${VAR}${F}.txt timeout 600 python bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-side > $O/bench_${VAR}${F}_$rep.log 2>&1
  grep '^{' $O/bench_${VAR}${F}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$VAR=$F rep $rep ms/step', d['ms_per_step'], '| dominant', r['kernel'][:36], r['achieved'], '| all conv', r['all_conv_kernels']['achieved'], r['all_conv_kernels']['ms_per_step'])"
done; done
