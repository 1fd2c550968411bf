#!/bin/bash
# local (not on the GPU box): copy what gpu_round6_profiles.sh <tag> <prefix> merged back under gpurun_out/<tag>/ into profiles/<prefix>_*. usage: copy_round6_profiles.sh <tag> <prefix>
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; P=profiles/$2
cp $O/hbm_counters.json ${P}_bench_1gpu_hbm_counters.json
cp $O/hbm_counters_bf16.json ${P}_bench_1gpu_bf16_hbm_counters.json
cp $O/mem/mem_path_hbm_counters.json ${P}_memory_path_hbm_counters.json
cp $O/mem/mem_probe.json ${P}_memory_path_probe.json
for f in bench_1gpu_full.json bench_1gpu_bf16.json bench_config5.json bench_cpu_batch8.json bench_input_edge.json bench_meminit.json bench_mldg.json bench_mldg_bf16.json bench_split_off.json bench_2rank_gloo.log \
         conv_shapes.txt conv_shapes_bf16.txt kernel_stats_bf16.csv kernel_stats_f32.csv kernel_stats_serialised_bf16.csv kernel_stats_serialised_f32.csv mfma_util_bf16.json mfma_util_f32.json one_conv_sq_counters.txt soak_bf16.txt soak_f32.txt; do
  cp $O/$f ${P}_$f
done
python - <<PY
import json
d = json.loads(open('${P}_bench_1gpu_full.json').read().strip().splitlines()[-1])
r = d['roofline']
print('ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'traffic x', r['traffic_over_algorithmic'], 'stale', r['traffic_stale'], 'conv ms', r['all_conv_kernels']['ms_per_step'])
print('stamp', json.load(open('${P}_bench_1gpu_hbm_counters.json'))['lib_stamp'][:16], 'lib', open('pinthememory_amd/libpinmem_hip.so.stamp').read()[:16])
PY
