cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_SPLIT_MIN_K=0 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv or wino or split" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" PM_PROFILE_DUMP=$O/shapes_$tag.txt timeout 600 python bench.py --no-cpu-baseline --no-side > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])")"; }
run k129 PM_SPLIT_MIN_K=129
run k0 PM_SPLIT_MIN_K=0
run k65 PM_SPLIT_MIN_K=65
run k129b PM_SPLIT_MIN_K=129
run k0b PM_SPLIT_MIN_K=0
run k65b PM_SPLIT_MIN_K=65
