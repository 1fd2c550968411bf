#!/bin/bash
# the driver's own bench command at HEAD (default flags), and the bf16 line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default.json; python -c "
import json; j=json.load(open('$O/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('traffic_stale'), j['side']['bf16']['ms_per_step'], j['side'].get('bf16_graphed',{}).get('ms_per_step'), j['cpu_baseline']['value'], j['config'].get('host_enqueue_ms'))"
