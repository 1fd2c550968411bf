#!/bin/bash
# round 6: whole Winograd point GEMMs per XCD (PM_BATCH_XCD, default 1) against the per-launch tile remap (0): traffic and time per shape, kernel tests, the bench line both ways
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for g in 1 0; do
  echo "#### PM_BATCH_XCD=$g"
  PM_BATCH_XCD=$g bash tools/gpu_r6_traffic1.sh $1/g$g "8 512 48 48 512 3 1 1" "8 2048 48 48 256 3 6 6" "8 256 48 48 2048 3 1 1" "8 256 192 192 256 3 1 1" | grep -v "^\[" | grep -v "wino_"
  for shape in "8 512 48 48 512 3 1 1" "8 2048 48 48 256 3 6 6" "8 256 48 48 2048 3 1 1" "8 256 192 192 256 3 1 1" "8 128 96 96 128 3 1 1"; do PM_BATCH_XCD=$g timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
done 2>&1 | tee $O/batchxcd.log
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino or split" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])")"; }
run b1 PM_BATCH_XCD=1
run b0 PM_BATCH_XCD=0
run b1b PM_BATCH_XCD=1
run b0b PM_BATCH_XCD=0
