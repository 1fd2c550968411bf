#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for pf in 0 1 0 1; do
  echo "== wide everywhere, PM_C16W_PREFETCH=$pf"; PM_C16W_PREFETCH=$pf PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep "final1\|aspp\|dsn\|layer4.conv2" | tee -a $O/probe_prefetch$pf.txt
done
timeout 900 python -m pytest tests/test_dist_gloo.py tests/test_bench_launcher.py -q -m gpu --durations=8 -k "two_ranks_on_gpu or self_launch" > $O/pytest_ranks.log 2>&1; tail -14 $O/pytest_ranks.log
