#!/bin/bash
# here (not on the GPU box): copy the summaries of a gpu_round4_profiles.sh run from gpurun_out/<tag>/ into profiles/<prefix>_*. usage: collect_profiles.sh <tag> <prefix>
R=/root/repo; O=$R/gpurun_out/$1; P=$R/profiles/$2
cp $O/bench_1gpu_full.json ${P}_bench_1gpu_full.json
cp $O/bench_1gpu_bf16.json ${P}_bench_1gpu_bf16.json
cp $O/hbm_counters.json ${P}_bench_1gpu_hbm_counters.json
cp $O/hbm_counters_bf16.json ${P}_bench_1gpu_bf16_hbm_counters.json
cp $O/mem/mem_path_hbm_counters.json ${P}_memory_path_hbm_counters.json
[ -f $R/gpurun_out/mem_probe.json ] && cp $R/gpurun_out/mem_probe.json ${P}_memory_path_probe.json
for f in conv_shapes.txt conv_shapes_bf16.txt kernel_stats_f32.csv kernel_stats_bf16.csv kernel_stats_serialised_f32.csv kernel_stats_serialised_bf16.csv mfma_util_f32.json mfma_util_bf16.json \
         bench_mldg.json bench_mldg_bf16.json bench_config5.json bench_meminit.json bench_input_edge.json bench_2rank_gloo.log clock_probe_step_bf16.txt clock_probe_step_f32.txt \
         soak_f32.txt soak_bf16.txt bench_cpu_batch8.json; do
  [ -f $O/$f ] && cp $O/$f ${P}_$f
done
ls -la ${P}_* | wc -l
