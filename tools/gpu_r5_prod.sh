#!/bin/bash
# producer-wave form of the ring kernels (conv16w.hip, PM_C16P=4): parity tests and per-shape probe against the all-waves-fetch form, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_C16P=4 timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16 or conv16 or lds_dma" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in 0 1; do
for p in 0 4 0 4; do
  echo "== wide everywhere, cfg $cfg, producers $p"; PM_C16P=$p PM_C16W_CFG=$cfg PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_cfg${cfg}_p$p.txt
done
done
