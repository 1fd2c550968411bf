#!/bin/bash
# persistent producer / consumer form of the ring kernels (conv16w.hip conv16p_kernel, PM_C16P=1): parity tests and per-shape probe against the default routing, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_C16P=1 timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16 or conv16 or lds_dma" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
echo "== default routing"; PROBE_CONV16=1 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_default_$rep.txt
echo "== persistent wide everywhere, the cost model picks tile and split"; PM_C16P=1 PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_persist_$rep.txt
done
echo "== persistent 256 x 128"; PM_C16P=1 PM_C16W_CFG=0 PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_persist_cfg0.txt
echo "== persistent 128 x 256"; PM_C16P=1 PM_C16W_CFG=1 PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_persist_cfg1.txt
