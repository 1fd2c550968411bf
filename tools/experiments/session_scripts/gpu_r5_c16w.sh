#!/bin/bash
# round 5: the wide LDS-DMA kernel (conv16w.hip) -- kernel tests on its route, then the per-shape probe: narrow tiles vs wide tiles at forced K-splits. usage: gpu_r5_c16w.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16_activations or conv16_on_the_48x48" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for r in 4 3; do
  echo "== PROBE_CONV16=$r (4: per shape without the wide kernel, 3: wide wherever possible, planner's K-split)"; PROBE_CONV16=$r timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_route$r.txt
done
for ks in 1 2 3 4 6; do
  echo "== wide, K-split $ks"; PM_C16W_KS=$ks PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_wide_ks$ks.txt
done
