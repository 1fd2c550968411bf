#!/bin/bash
# round 4: tests of the bf16 convolutions, the per-shape probe of the LDS-DMA kernel (everywhere / default routing) and two bench lines of the tier
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16_activations or 48x48 or tier_mixed or bn_statistics_bf16" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
{
echo "== conv16 everywhere"; PROBE_CONV16=2 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu.ids
echo "== default routing"; timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/c16probe.txt 2>&1
cat $O/c16probe.txt
for i in 1 2; do timeout 600 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-side 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 ms/step', d['ms_per_step'], 'dominant', d['roofline']['kernel'][:40], d['roofline']['achieved'], 'all conv', d['roofline']['all_conv_kernels']['achieved'])"; done
