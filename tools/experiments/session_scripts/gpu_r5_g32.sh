#!/bin/bash
# gemm32p.hip (fp32 persistent GEMM): kernel + model parity tests with the route on, then the default fp32 bench line with and without it (PM_GEMM32P=0), one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "not bf16 and not conv16 and not pw16 and not wgrad16" > $O/pytest_k.log 2>&1; tail -4 $O/pytest_k.log
timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu -k "agg_train_step_vs_oracle" > $O/pytest_m.log 2>&1; tail -4 $O/pytest_m.log
for i in 1 2; do
  PM_GEMM32P=0 timeout 600 python bench.py --no-cpu-baseline --no-side --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('PM_GEMM32P=0', j['ms_per_step'], j['roofline']['kernel'][:50], j['roofline']['achieved'], j['roofline']['all_conv_kernels']['achieved'], j['roofline']['all_conv_kernels']['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --no-side --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default     ', j['ms_per_step'], j['roofline']['kernel'][:50], j['roofline']['achieved'], j['roofline']['all_conv_kernels']['achieved'], j['roofline']['all_conv_kernels']['ms_per_step'])"
done
