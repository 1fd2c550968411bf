#!/bin/bash
# round 6: which launches of the split path move the assembled-gradient error of the bs=2 128^2 test (vs fp64)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 600 python -m pytest tests/test_model_parity.py -m gpu -q -s -k "test_train_forward_backward_vs_oracle" 2>&1 | grep -E "^.grads vs fp64|passed|failed" | cut -c1-330
done
