#!/bin/bash
# interval-form CE: kernel tests + block-size / segment sweep (HIP-event timing of tools/mem_probe.py). usage: gpu_r6_ce.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "upsample_ce" 2>&1 | tail -2
PM_CE_ROWS2_SEG=192 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "upsample_ce" 2>&1 | tail -2
for ST in "48 256" "48 128" "96 256" "96 384" "192 192" "192 384" "192 576"; do set -- $ST; echo "SEG=$1 THREADS=$2"; PM_CE_ROWS2_SEG=$1 PM_CE_ROWS2_THREADS=$2 timeout 300 python tools/mem_probe.py 2>&1 | grep -E "main_ce_fwd"; done
