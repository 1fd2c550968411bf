#!/bin/bash
# round 6: the wave-streamed pointwise GEMM (csrc/pwstream.hip, PM_PWSTREAM default 1): its kernel test, per-shape time with an output ring (on / off), kernel tests, the bench line both ways
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -s -k "pwstream" > $O/pytest_pwstream.log 2>&1; echo "pytest pwstream exit $?"; grep -E "pwstream vs|passed|failed|Error|assert" $O/pytest_pwstream.log | head -20
for on in 1 0; do
  echo "#### PM_PWSTREAM=$on"
  for shape in "8 64 192 192 256 1 0 1" "8 128 96 96 512 1 0 1" "8 64 192 192 64 1 0 1"; do ONE_RING=4 PM_PWSTREAM=$on timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
  for shape in "8 256 192 192 64 1 0 1" "8 512 96 96 128 1 0 1"; do ONE_MODE=dgrad PM_PWSTREAM=$on timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
done 2>&1 | tee $O/shapes.log
[ "$2" = "quick" ] && exit 0
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino or split" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" PM_PROFILE_DUMP=$O/shapes_$tag.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])")"; }
run p1 PM_PWSTREAM=1
run p0 PM_PWSTREAM=0
run p1b PM_PWSTREAM=1
run p0b PM_PWSTREAM=0
