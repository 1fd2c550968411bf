#!/bin/bash
# round-2 evidence set in one gpurun call: bench lines (fp32 metric, bf16 tier, config5), kernel trace, HBM and MFMA counter passes (separate
# --pmc runs), memory-path probe with counters, the GEMM lab. usage: gpu_round2_profiles.sh <tag>  -> gpurun_out/<tag>/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_PROFILE_DUMP=$O/prof_dump.txt timeout 900 python bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log > $O/bench_1gpu.json; cut -c1-200 $O/bench_1gpu.json
python tools/conv_shapes.py $O/prof_dump.txt > $O/conv_shapes.txt 2>&1; head -2 $O/conv_shapes.txt
timeout 600 rocprofv3 --kernel-trace -d $O/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) $O/kernel_stats.csv 8 | head -6
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
python tools/pmc_bench_summary.py $(find $O/pmc_fetch -name '*.db' | head -1) $(find $O/pmc_write -name '*.db' | head -1) $O/hbm_counters.json | head -5
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_mfma -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma_summary.py $(find $O/pmc_mfma -name '*.db' | head -1) $O/mfma_util.json | head -6
find $O -name '*.db' -delete
bash tools/gpu_mem_path.sh $1/mem | tail -12
bash tools/gpu_bf16_counters.sh $1/bf16 | tail -12
timeout 600 python bench.py --workload config5 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_config5.json; cut -c1-200 $O/bench_config5.json
(cd tools/micro && timeout 300 ./gemm_lab > ../../$O/gemm_lab.txt 2>&1); tail -3 $O/gemm_lab.txt
