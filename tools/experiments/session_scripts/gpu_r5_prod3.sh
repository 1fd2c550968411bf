#!/bin/bash
# this build against tools/ab/libpinmem_prev.so (the previous commit's library) on one box: conv16 parity tests, per-shape probe, captured bf16 step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16 or conv16 or lds_dma" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
echo "== previous build"; PM_LIB=tools/ab/libpinmem_prev.so PROBE_CONV16=1 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_prev_$rep.txt
echo "== this build"; PROBE_CONV16=1 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_new_$rep.txt
done
for i in 1 2 3; do
  PM_LIB=tools/ab/libpinmem_prev.so timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('previous (graph)', j['ms_per_step'])"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('this     (graph)', j['ms_per_step'])"
done
