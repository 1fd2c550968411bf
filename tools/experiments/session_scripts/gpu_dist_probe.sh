#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 300 python -m pytest tests/test_dist_gloo.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/cpu_enqueue_time.py 2>&1 | grep "host\|enqueue\|idle" | tee $O/enq_plain.log
PM_DIST_FORCE=1 PM_DIRECT_RCCL=0 timeout 300 python tools/cpu_enqueue_time.py 2>&1 | grep "host\|enqueue\|idle\|Error\|error" | tee $O/enq_force_torch.log
PM_DIST_FORCE=1 timeout 300 python tools/cpu_enqueue_time.py 2>&1 | grep "host\|enqueue\|idle\|Error\|error\|warn" | tee $O/enq_force_direct.log
PM_DIST_FORCE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rccl1_direct.log 2>&1; echo "bench exit $?"; grep metric $O/bench_rccl1_direct.log | cut -c1-260
