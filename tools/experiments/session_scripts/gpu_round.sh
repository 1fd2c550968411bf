#!/bin/bash
# one gpurun call: GPU test suite, default bench line, one-rank RCCL rehearsal, MFMA counter pass. Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log; fi
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
if [ -n "$DO_DIST" ]; then PM_DIST_FORCE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rccl1.log 2>&1; echo "rccl1 exit $?"; tail -1 $O/bench_rccl1.log | cut -c1-300; fi
if [ -n "$DO_PMC" ]; then
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_mfma -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_mfma.log 2>&1
  DB=$(find $O/pmc_mfma -name '*.db' | head -1); echo "db $DB"
  python tools/pmc_mfma_summary.py $DB $O/mfma_util.json 2>&1 | tail -16
  find $O/pmc_mfma -name '*.db' -size +40M -delete
fi
