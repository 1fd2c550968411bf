#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | grep '^{' | tee $O/bench_bf16_$i.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 ms/step', d['ms_per_step'], d['value'], d['roofline']['kernel'][:50], d['roofline']['achieved'])"; done
