#!/bin/bash
# per-shape weight-gradient table of the bf16 step, register-staged kernel (PM_WGRAD16=0) vs wgrad16.hip, one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for v in 0 1; do
  PM_WGRAD16=$v PM_PROFILE_DUMP=$O/dump_$v.txt timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-side --steps 10 > $O/bench_$v.log 2>&1
  grep '^{' $O/bench_$v.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('PM_WGRAD16=$v', j['ms_per_step'], j['roofline']['kernel'][:60], j['roofline']['achieved'], j['roofline'].get('all_conv_kernels'))"
  python tools/conv_shapes.py $O/dump_$v.txt 2 200 > $O/shapes_$v.txt
  echo "== PM_WGRAD16=$v: weight-gradient rows"; head -2 $O/shapes_$v.txt; awk '$5==2' $O/shapes_$v.txt | sort -k2 -n -r | head -40
done
