#!/bin/bash
# configs[2] evidence: bench line + MFMA-utilisation pass + FETCH_SIZE / WRITE_SIZE passes of the bf16-MFMA tier
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_bf16.json; cut -c1-200 $O/bench_bf16.json
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_mfma -- python bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma_summary.py $(find $O/pmc_mfma -name '*.db' | head -1) $O/mfma_util_bf16.json | head -8
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
python tools/pmc_bench_summary.py $(find $O/pmc_fetch -name '*.db' | head -1) $(find $O/pmc_write -name '*.db' | head -1) $O/hbm_counters_bf16.json | head -8
find $O -name '*.db' -delete
