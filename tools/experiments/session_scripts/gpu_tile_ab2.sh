#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; env "$@" PM_PROFILE_DUMP=$O/prof_$name.txt timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | grep '^{' | cut -c1-160; }
run base PM_X=0
run w64 PM_WINO_BM=64 PM_WINO_BN=64
run w64x128 PM_WINO_BM=64 PM_WINO_BN=128
run w128x64 PM_WINO_BM=128 PM_WINO_BN=64
