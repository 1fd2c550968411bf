#!/bin/bash
# in-situ tile A/B: per-shape launch times of the serialised profile steps of bench.py under the planner's tuning knobs. usage: gpu_tile_ab.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; env "$@" PM_PROFILE_DUMP=$O/prof_$name.txt timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | grep '^{' | cut -c1-160; }
run base PM_X=0
run bm128 PM_PREFER_BM64=0
run wide PM_WIDE_BN=1
run wide128 PM_WIDE_BN=1 PM_PREFER_BM64=0
for v in bm128 wide wide128; do echo "== base vs $v"; python tools/conv_compare.py $O/prof_base.txt $O/prof_$v.txt 2 | head -1; done
