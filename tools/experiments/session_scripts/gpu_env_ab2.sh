#!/bin/bash
# in-situ A/B of one environment knob: gpu_env_ab2.sh <tag> <VAR> <valA> <valB>  -> bench lines (alternated twice) + per-shape comparison
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for v in $3 $4; do env $2=$v PM_PROFILE_DUMP=$O/prof_$v.txt timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2=$v', d['ms_per_step'], d['roofline']['all_conv_kernels']['ms_per_step'])"; done; done
python tools/conv_compare.py $O/prof_$3.txt $O/prof_$4.txt 2 > $O/compare.txt; head -1 $O/compare.txt
