#!/bin/bash
# round 5: N > 1 rehearsals on the one-GPU box: bench.py --gpus 8 over gloo at a reduced crop, and 8 ranks x 8 images == one process x 64 images. usage: gpu_r5_ranks.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --size 256 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_8rank_gloo.log 2>&1; tail -c 2500 $O/bench_8rank_gloo.log
timeout 2400 python tools/gloo_ranks_probe.py 8 8 256 > $O/ranks_probe_8x8_256.log 2>&1; tail -20 $O/ranks_probe_8x8_256.log
