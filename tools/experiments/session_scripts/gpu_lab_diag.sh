#!/bin/bash
# SQ / TA / TCP counter passes over the gemm_lab "diag" variants (production 64x64 tile and its ablations on the dominant-kernel shape)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/labdiag; mkdir -p $O
cd tools/micro
./gemm_lab diag | tee $O/times.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_LDS" \
         "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/p$i -- ./gemm_lab diag > $O/p$i.log 2>&1
done
python - $O <<'PY'
import sqlite3, sys, glob, collections, re
O = sys.argv[1]
res = collections.OrderedDict()
for d in sorted(glob.glob(O + '/p*/')):
    for db in glob.glob(d + '/**/*.db', recursive=True):
        cur = sqlite3.connect(db).cursor()
        for kn, cn, n, v in cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name"):
            m = re.search(r'gemm_lab<([^>]*)>', kn)
            if not m: continue
            res.setdefault(m.group(1), {})[cn] = v / n
for k, d in res.items():
    print(k)
    for c, v in d.items():
        print('   %-36s %14.0f' % (c, v))
PY
find $O -name '*.db' -delete
