#!/bin/bash
# round 6: same-box A/B of environment knobs on the default bench line. usage: gpu_r6_ab.sh <tag> "ENV1=.. ENV2=.." "ENV.." ...   (first argument after the tag may be empty: "")
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg PM_PROFILE_DUMP=$O/shapes_$i.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$i.json 2> $O/bench_$i.err
  echo "[$i] $cfg: $(python -c "import json,sys; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])")"
done
