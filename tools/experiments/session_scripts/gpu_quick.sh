#!/bin/bash
# quick A/B: selected GPU tests + short bench (no cpu baseline). usage: gpu_quick.sh <tag> [pytest -k expr]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$2" 2>&1 | tail -4; fi
PM_PROFILE_DUMP=$O/prof_dump.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench exit $?"
grep '^{' $O/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('ms/step', d['ms_per_step'], 'img/s', d['value'], 'dominant', r['kernel'][:44], r['achieved'], 'all conv', r['all_conv_kernels']['achieved'], r['all_conv_kernels']['ms_per_step'])"
