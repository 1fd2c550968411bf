#!/bin/bash
# EXPERIMENT: is the empty persistent kernel's time per tile or per K-step? (dbg 7: no fetch, no multiplication, no epilogue)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
export PM_C16P=1 PM_C16W_CFG=1
i=0
for shape in "8 256 192 192 256 3 1 1" "8 256 192 192 256 1 0 1" "8 64 192 192 256 3 1 1" "8 1024 192 192 256 1 0 1" "2 256 192 192 256 3 1 1"; do
for dbg in 7 0; do
  i=$((i+1))
  PM_C16W_DBG=$dbg timeout 120 rocprofv3 --kernel-trace -d $O/u_$i -- python tools/one_conv.py $shape 3 20 > $O/u_$i.log 2>&1 < /dev/null
  timeout 60 python tools/kernel_avg.py $O/u_$i conv16 "shape $shape dbg $dbg" < /dev/null
  rm -rf $O/u_$i
done; done
