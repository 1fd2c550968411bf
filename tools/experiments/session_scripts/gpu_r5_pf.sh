#!/bin/bash
# fragment requests one 16-k group ahead in the persistent kernels' multiplying waves (experimental library tools/ab/libpinmem_pf.so) against the shipped build, one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PM_LIB=tools/ab/libpinmem_pf.so timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv_bf16 or conv16 or lds_dma or wgrad16" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
echo "== shipped"; PROBE_CONV16=1 timeout 300 python tools/conv16_probe.py 2>&1 | grep "aspp 3x3 d12\|final1\|sum"
echo "== prefetch"; PM_LIB=tools/ab/libpinmem_pf.so PROBE_CONV16=1 timeout 300 python tools/conv16_probe.py 2>&1 | grep "aspp 3x3 d12\|final1\|sum"
done
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shipped  (graph)', j['ms_per_step'])"
  PM_LIB=tools/ab/libpinmem_pf.so timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile --graph --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('prefetch (graph)', j['ms_per_step'])"
done
