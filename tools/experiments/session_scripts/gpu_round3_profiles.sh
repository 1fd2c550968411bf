#!/bin/bash
# final round-3 evidence set in one gpurun call. Counter passes FIRST (bench.py quotes roofline.traffic from the latest committed counter
# summaries: they are copied into profiles/ on the box before the bench lines are produced, and again from gpurun_out/ into the repo afterwards).
# usage: gpu_round3_profiles.sh <tag> <prefix>   -> gpurun_out/<tag>/, profiles/<prefix>_* on the box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; P=$2; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
python tools/pmc_bench_summary.py $(find $O/pmc_fetch -name '*.db' | head -1) $(find $O/pmc_write -name '*.db' | head -1) $O/hbm_counters.json | head -5
cp $O/hbm_counters.json profiles/${P}_bench_1gpu_hbm_counters.json
bash tools/gpu_mem_path.sh $1/mem | tail -14
cp $O/mem/mem_path_hbm_counters.json profiles/${P}_memory_path_hbm_counters.json
bash tools/gpu_mem_trace.sh $1/mem | tail -14
PM_PROFILE_DUMP=$O/prof_dump.txt timeout 900 python bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log > $O/bench_1gpu.json; cut -c1-200 $O/bench_1gpu.json
python tools/conv_shapes.py $O/prof_dump.txt 2 200 > $O/conv_shapes.txt 2>&1; head -2 $O/conv_shapes.txt
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/kernel_stats.csv 8 | head -3
PM_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/kts.log 2>&1
python tools/rocpd_stats.py $(find $O/kts -name '*.db' | head -1) $O/kernel_stats_serial.csv 4 | head -3
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_mfma -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma_summary.py $(find $O/pmc_mfma -name '*.db' | head -1) $O/mfma_util.json | head -6
find $O -name '*.db' -delete
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_1gpu_bf16.json; cut -c1-200 $O/bench_1gpu_bf16.json
timeout 600 python bench.py --workload config5 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_config5.json; cut -c1-200 $O/bench_config5.json
timeout 600 python bench.py --no-cpu-baseline --no-profile --input-edge 2>/dev/null | grep '^{' > $O/bench_input_edge.json; cut -c1-200 $O/bench_input_edge.json
