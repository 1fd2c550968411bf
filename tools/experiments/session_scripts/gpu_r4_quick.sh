#!/bin/bash
# round 4 quick loop: selected GPU tests, then short bench lines for the dtypes given. usage: gpu_r4_quick.sh <tag> "<pytest -k expr or ''>" [dtype ...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
K="$2"; shift 2
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -25 $O/pytest.log; fi
for DT in "$@"; do
  PM_PROFILE_DUMP=$O/prof_dump_$DT.txt timeout 600 python bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$DT.log 2>&1; echo "bench $DT exit $?"
  grep '^{' $O/bench_$DT.log > $O/bench_$DT.json
  python - $O/bench_$DT.json <<'PY' || tail -30 $O/bench_$DT.log
import sys, json
d = json.loads(open(sys.argv[1]).readline()); r = d['roofline']
print(d['dtype'], 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'loss', d['config']['final_loss'], '| dominant', r['kernel'][:48], r['achieved'], 'TF | all conv', r['all_conv_kernels']['achieved'], 'TF', r['all_conv_kernels']['ms_per_step'], 'ms')
PY
done
