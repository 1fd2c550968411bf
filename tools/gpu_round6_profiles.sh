#!/bin/bash
# round-6 evidence set in one gpurun call. usage: gpu_round6_profiles.sh <tag> <prefix>   -> gpurun_out/<tag>/ (copy what is to be judged into profiles/<prefix>_*)
# Order matters: the FETCH / WRITE counter passes come first -- bench.py quotes the latest counter summary under profiles/ (and its build stamp) in roofline.traffic.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; P=$2; mkdir -p $O
for DT in f32 bf16; do
  SFX=$([ $DT = bf16 ] && echo _bf16 || echo "")
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch$SFX -- python bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/pmc_fetch$SFX.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write$SFX -- python bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/pmc_write$SFX.log 2>&1
  python tools/pmc_bench_summary.py $(find $O/pmc_fetch$SFX -name '*.db' | head -1) $(find $O/pmc_write$SFX -name '*.db' | head -1) $O/hbm_counters$SFX.json | head -4
  cp $O/hbm_counters$SFX.json profiles/${P}_bench_1gpu${SFX}_hbm_counters.json
  find $O -name '*.db' -delete
done
bash tools/gpu_mem_path.sh $1/mem | tail -6
cp $O/mem/mem_path_hbm_counters.json profiles/${P}_memory_path_hbm_counters.json 2>/dev/null
PM_PROFILE_DUMP=$O/prof_dump.txt timeout 1200 python bench.py > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log > $O/bench_1gpu_full.json; cut -c1-200 $O/bench_1gpu_full.json
python tools/conv_shapes.py $O/prof_dump.txt 2 200 > $O/conv_shapes.txt 2>&1; head -2 $O/conv_shapes.txt
python tools/conv_shapes.py $O/prof_dump.txt.bf16 2 200 > $O/conv_shapes_bf16.txt 2>&1; head -2 $O/conv_shapes_bf16.txt
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-side > $O/bench_bf16.log 2>&1; grep '^{' $O/bench_bf16.log > $O/bench_1gpu_bf16.json; cut -c1-160 $O/bench_1gpu_bf16.json
for DT in f32 bf16; do
  timeout 600 rocprofv3 --kernel-trace -d $O/kt_$DT -- python bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kt_$DT.log 2>&1
  python tools/rocpd_stats.py $(find $O/kt_$DT -name '*.db' | head -1) $O/kernel_stats_$DT.csv 4 | head -2
  PM_OVERLAP_WGRAD=0 PM_COMMIT_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts_$DT -- python bench.py --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kts_$DT.log 2>&1
  python tools/rocpd_stats.py $(find $O/kts_$DT -name '*.db' | head -1) $O/kernel_stats_serialised_$DT.csv 4 | head -2
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_mfma_$DT -- python bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/pmc_mfma_$DT.log 2>&1
  python tools/pmc_mfma_summary.py $(find $O/pmc_mfma_$DT -name '*.db' | head -1) $O/mfma_util_$DT.json | head -5
  find $O -name '*.db' -delete
done
PM_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline --no-side 2>/dev/null | grep '^{' > $O/bench_split_off.json; cut -c1-160 $O/bench_split_off.json
timeout 600 python bench.py --workload mldg --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_mldg.json; cut -c1-160 $O/bench_mldg.json
timeout 600 python bench.py --workload mldg --dtype bf16 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_mldg_bf16.json; cut -c1-160 $O/bench_mldg_bf16.json
timeout 600 python bench.py --workload config5 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_config5.json; cut -c1-160 $O/bench_config5.json
timeout 600 python bench.py --workload meminit --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_meminit.json; cut -c1-160 $O/bench_meminit.json
timeout 600 python bench.py --no-cpu-baseline --no-profile --no-side --input-edge 2>/dev/null | grep '^{' > $O/bench_input_edge.json; cut -c1-160 $O/bench_input_edge.json
PM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --size 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo.log 2>&1; grep '^{' $O/bench_2rank_gloo.log | cut -c1-200
timeout 600 python tools/soak.py 6 --deterministic 2>&1 | grep steps > $O/soak_f32.txt; timeout 600 python tools/soak.py 6 --deterministic --bf16 2>&1 | grep steps > $O/soak_bf16.txt; paste -d'\n' $O/soak_f32.txt $O/soak_bf16.txt | cut -c1-120
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-batch 8 --no-profile --no-side 2>/dev/null | grep '^{' > $O/bench_cpu_batch8.json; python -c "import json; print(json.load(open('$O/bench_cpu_batch8.json'))['cpu_baseline'])" | cut -c1-300
bash tools/gpu_r6_pmc1.sh $1/pmc1 > $O/one_conv_sq_counters.txt 2>&1; tail -3 $O/one_conv_sq_counters.txt
