#!/bin/bash
# round-end evidence: bench line, kernel trace (per-kernel + per-shape tables), FETCH_SIZE / WRITE_SIZE passes. usage: gpu_final_profile.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_PROFILE_DUMP=$O/prof_dump.txt timeout 600 python bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log > $O/bench_1gpu.json; cut -c1-200 $O/bench_1gpu.json
python tools/conv_shapes.py $O/prof_dump.txt > $O/conv_shapes.txt 2>&1; head -3 $O/conv_shapes.txt
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/kernel_stats.csv 8 | head -5
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
python tools/pmc_bench_summary.py $(find $O/pmc_fetch -name '*.db' | head -1) $(find $O/pmc_write -name '*.db' | head -1) $O/hbm_counters.json | head -6
find $O -name '*.db' -delete
