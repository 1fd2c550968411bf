#!/bin/bash
# round 6: N-group tile order (PM_N_GROUP, default 8) against the plain row-major order (0): traffic and time of the wide 1x1s, then the bench line both ways
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for g in 8 0 4; do
  echo "#### PM_N_GROUP=$g"
  PM_N_GROUP=$g bash tools/gpu_r6_traffic1.sh $1/g$g "8 512 48 48 2048 1 0 1" "8 1024 48 48 2048 1 0 1" "8 2048 48 48 512 1 0 1" | grep -v "^\[" 
  for shape in "8 512 48 48 2048 1 0 1" "8 1024 48 48 2048 1 0 1" "8 2048 48 48 512 1 0 1"; do PM_N_GROUP=$g timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
done 2>&1 | tee $O/ngroup.log
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino or split" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])")"; }
run g8 PM_N_GROUP=8
run g0 PM_N_GROUP=0
run g8b PM_N_GROUP=8
run g0b PM_N_GROUP=0
