#!/bin/bash
# round 6: the restructured gradient tests, the mldg graph test, the new kernel cases, then the default bench line with its side measurements
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests/test_model_parity.py -m gpu -x -q -s -k "test_train_forward_backward_vs_oracle or test_agg_train_step_vs_oracle_and_golden or graphed_mldg" > $O/pytest_a.log 2>&1; echo "pytest a exit $?"; grep -E "grads vs fp64|passed|failed|Error" $O/pytest_a.log | cut -c1-300 | tail -12
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "wgrad16 or conv16_on_the_48x48" > $O/pytest_b.log 2>&1; echo "pytest b exit $?"; tail -3 $O/pytest_b.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; tail -1 $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('kernel','achieved','frac')} if d.get('roofline') else None)
for k,v in (d.get('side') or {}).items():
    print(k, {kk: v.get(kk) for kk in ('ms_per_step','host_enqueue_ms','form_rule','error')}, 'graphed' in v and v['graphed'])
"
