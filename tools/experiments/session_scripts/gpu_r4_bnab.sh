#!/bin/bash
# round 4: A/B of the fixed-channel-group bf16 BatchNorm passes (PM_BN16_FIXED=0|1) on the same box: tests, then two bench lines each way + serialised kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "act16 or bf16_activations or config3_bf16_mfma" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
for rep in 1 2; do for F in 0 1; do
  PM_BN16_FIXED=$F timeout 600 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f${F}_$rep.log 2>&1
  grep '^{' $O/bench_f${F}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fixed=$F rep $rep ms/step', d['ms_per_step'])"
done; done
for F in 0 1; do
  PM_BN16_FIXED=$F PM_OVERLAP_WGRAD=0 PM_COMMIT_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts_f$F -- python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kts_f$F.log 2>&1
  python tools/rocpd_stats.py $(find $O/kts_f$F -name '*.db' | head -1) $O/kernel_stats_serialised_f$F.csv 4 | head -2
  grep -E "bn16|ew16" $O/kernel_stats_serialised_f$F.csv | cut -c1-200
  find $O -name '*.db' -delete
done
