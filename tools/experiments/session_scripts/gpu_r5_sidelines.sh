#!/bin/bash
# the side-workload bench lines (mldg on both tiers, configs[4], memory initialisation) with config.host_enqueue_ms; usage: gpu_r5_sidelines.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 600 python bench.py --workload mldg --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_mldg.json; cut -c1-160 $O/bench_mldg.json
timeout 600 python bench.py --workload mldg --dtype bf16 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_mldg_bf16.json; cut -c1-160 $O/bench_mldg_bf16.json
timeout 600 python bench.py --workload config5 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_config5.json; cut -c1-160 $O/bench_config5.json
timeout 600 python bench.py --workload meminit --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_meminit.json; cut -c1-160 $O/bench_meminit.json
for f in mldg mldg_bf16 config5 meminit; do python -c "import json; j=json.load(open('$O/bench_$f.json')); print('$f', j['ms_per_step'], j['config'].get('step_form'), j['config'].get('host_enqueue_ms'))"; done
