#!/bin/bash
# round 5: new / changed parity tests + the small-kernel attribution trace. usage: gpu_r5_second.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests/test_model_parity.py -q -m gpu -k "mldg or graphed or bf16_tier_assembled or config3_bf16 or config5_full or bf16_tier_production" --durations=12 > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log
grep -v "^$" $O/pytest_new.log | tail -60
DTYPE=bf16 timeout 600 python tools/small_kernel_trace.py > $O/small_kernels_bf16.txt 2>&1; tail -90 $O/small_kernels_bf16.txt
