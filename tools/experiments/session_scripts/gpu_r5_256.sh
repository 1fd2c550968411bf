#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16_activations or conv16_on" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for c in -1 2 -1 2; do
  echo "== wide everywhere, PM_C16W_CFG=$c (-1: cost model over 256x128 / 128x256, 2: 256x256 two-stage)"; PM_C16W_CFG=$c PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep "final1\|aspp\|dsn\|layer4.conv2\|layer2.conv2" | tee -a $O/probe_cfg$c.txt
done
