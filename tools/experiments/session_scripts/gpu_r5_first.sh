#!/bin/bash
# round 5, first call: the new / changed parity tests, the vendor bf16 calibration, bench lines (default, --graph, bf16, mldg, config5). usage: gpu_r5_first.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_parity.py -x -q -m gpu -k "mldg or graphed or bf16_tier_assembled or config3_bf16 or config5_full or bf16_tier_production" --durations=8 > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log
tail -30 $O/pytest_new.log
timeout 600 python tools/blas_probe16.py $O/vendor_bf16_gemm_calibration.txt > $O/blas16.log 2>&1; tail -30 $O/blas16.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
timeout 600 python bench.py --no-cpu-baseline --graph --no-side > $O/bench_graph.json 2> $O/bench_graph.err; tail -c 1500 $O/bench_graph.json; tail -3 $O/bench_graph.err
timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --graph > $O/bench_bf16_graph.json 2> $O/bench_bf16_graph.err; tail -c 1500 $O/bench_bf16_graph.json; tail -3 $O/bench_bf16_graph.err
timeout 600 python bench.py --workload mldg --steps 5 --warmup 2 > $O/bench_mldg.json 2> $O/bench_mldg.err; tail -c 1200 $O/bench_mldg.json
timeout 600 python bench.py --workload config5 --steps 5 --warmup 2 > $O/bench_config5.json 2> $O/bench_config5.err; tail -c 1500 $O/bench_config5.json
