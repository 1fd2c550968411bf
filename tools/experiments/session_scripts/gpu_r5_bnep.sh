#!/bin/bash
# round 5: BatchNorm statistics out of the producing convolution's epilogue, PER SHAPE (outputs of at least N MB): same-box alternated A/B on both tiers. usage: gpu_r5_bnep.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu > $O/pytest_kernels.log 2>&1; tail -4 $O/pytest_kernels.log
run() { timeout 600 python bench.py --no-cpu-baseline --no-profile --no-side --steps 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"; }
for i in 1 2; do
  for mb in 0 60 120; do echo "bf16 PM_BN_EP16_MIN_MB=$mb: $(PM_BN_EP16_MIN_MB=$mb run --dtype bf16)" | tee -a $O/ab.txt; done
done
for i in 1 2; do
  for mb in 0 120 250; do echo "f32 PM_BN_EP_MIN_MB=$mb: $(PM_BN_EP_MIN_MB=$mb run)" | tee -a $O/ab.txt; done
done
