#!/bin/bash
# round 4: per-shape forward TFLOP/s of the LDS-DMA kernel under stage-count / tile / hybrid-tail configurations (one process each: the knobs are read at load)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
run() { echo "== $*"; env "$@" PROBE_CONV16=2 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu.ids; }
{
run PM_C16_NST=0
run PM_C16_NST=3
run PM_C16_NST=3 PM_C16_HYBRID=1
run PM_C16_NST=4 PM_C16_BM=128
run PM_C16_NST=4 PM_C16_BM=128 PM_C16_HYBRID=1
run PM_C16_NST=3 PM_C16_BM=128 PM_C16_HYBRID=1
run PM_C16_NST=4 PM_C16_BM=64
echo "== default routing (PM_CONV16=1)"; timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/c16probe.txt 2>&1
cat $O/c16probe.txt
PM_C16_NST=3 timeout 900 python -m pytest tests -m gpu -x -q -k "bf16_activations or hybrid_tail or tier_mixed" > $O/pytest_nst3.log 2>&1; echo "pytest nst3 exit $?"; tail -3 $O/pytest_nst3.log
PM_C16_NST=4 PM_C16_HYBRID=1 timeout 900 python -m pytest tests -m gpu -x -q -k "bf16_activations or hybrid_tail or tier_mixed" > $O/pytest_nst4h.log 2>&1; echo "pytest nst4+hybrid exit $?"; tail -3 $O/pytest_nst4h.log
