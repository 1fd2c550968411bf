#!/bin/bash
# the mldg step on both tiers, three runs each (box-to-box and run-to-run spread), with config.host_enqueue_ms
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
timeout 600 python bench.py --workload mldg --dtype bf16 --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mldg bf16', j['ms_per_step'], 'host', j['config']['host_enqueue_ms'])"
done
timeout 600 python bench.py --workload mldg --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mldg f32', j['ms_per_step'], 'host', j['config']['host_enqueue_ms'])"
