#!/bin/bash
# HBM-side traffic of single conv shapes (PROBE_ONLY filter of tools/conv_probe.py): FETCH_SIZE / WRITE_SIZE passes. usage: gpu_shape_traffic.sh <tag> "<name filter>"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
export PROBE_ONLY="$2"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -- python tools/conv_probe.py > $O/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -- python tools/conv_probe.py > $O/w.log 2>&1
python tools/pmc_bench_summary.py $(find $O/f -name '*.db' | head -1) $(find $O/w -name '*.db' | head -1) $O/traffic.json | head -12
tail -3 $O/f.log
find $O -name '*.db' -delete
