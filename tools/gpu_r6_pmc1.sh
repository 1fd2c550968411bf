#!/bin/bash
# round 6: SQ counter passes of ONE fp32 pointwise convolution (18432 x 2048 x 512) on the split path, one counter set per pass. usage: gpu_r6_pmc1.sh <tag> [n cin h w cout k pad dil]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
shift
ARGS="${@:-8 512 48 48 2048 1 0 1}"
python tools/one_conv32.py $ARGS 20
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/p_$tag -- python tools/one_conv32.py $ARGS 6 > $O/p_$tag.log 2>&1
  python - <<PY
import sqlite3, glob, sys
db = glob.glob('$O/p_$tag/**/*.db', recursive=True)
if not db: print('no db for $set'); print(open('$O/p_$tag.log').read()[-600:]); sys.exit()
c = sqlite3.connect(db[0])
try:
    rows = c.execute("select kernel_name, counter_name, sum(value) * 1.0 / count(distinct dispatch_id), count(distinct dispatch_id), avg(end - start) from counters_collection group by kernel_name, counter_name").fetchall()
except Exception as e:
    rows = []; print('query failed', e)
for r in rows:
    if 'conv' in r[0]: print(r[0][:60], r[1], 'per launch %.6g' % r[2], 'launches', r[3], 'avg ns %.0f' % r[4])
PY
  find $O/p_$tag -name '*.db' -delete
done
