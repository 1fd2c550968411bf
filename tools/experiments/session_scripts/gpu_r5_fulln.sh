#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16_activations or conv16_on" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for f in 0 1 0 1; do
  echo "== PM_C16_FULLN=$f"; PM_C16_FULLN=$f PROBE_ONLY=1x1 timeout 300 python tools/conv16_probe.py 2>&1 | grep "1x1" | tee -a $O/probe_fulln$f.txt
done
