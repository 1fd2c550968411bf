#!/bin/bash
# CU-partitioned side streams, same-box A/B of the fp32 step. usage: gpu_cumask_ab.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
run() { # label, env...
  L=$1; shift
  env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-profile --no-side 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', d['ms_per_step'])" | tee -a $O/cumask_ab.txt
}
for rep in 1 2; do
  run "baseline(all CUs shared)" PM_X=0
  run "wgrad 64 CUs" PM_WGRAD_CUS=64
  run "wgrad 96 CUs" PM_WGRAD_CUS=96
  run "wgrad 128 CUs" PM_WGRAD_CUS=128
  run "wgrad 192 CUs" PM_WGRAD_CUS=192
  run "commit 64 CUs" PM_COMMIT_CUS=64
  run "commit 128 CUs" PM_COMMIT_CUS=128
  run "wgrad 128 + commit 64" PM_WGRAD_CUS=128 PM_COMMIT_CUS=64
  run "wgrad 96 + commit 96" PM_WGRAD_CUS=96 PM_COMMIT_CUS=96
done
