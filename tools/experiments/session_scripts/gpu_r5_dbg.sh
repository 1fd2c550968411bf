#!/bin/bash
# EXPERIMENT: which part of the ring kernel's time is skeleton (prologue / epilogue / barriers), fetch, multiplication -- kernel durations from the trace, not the host clock
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for p in 0; do for dbg in 0 3 4 7; do
  export PM_C16P=$p PM_C16W_DBG=$dbg PM_C16W_CFG=0
  timeout 120 rocprofv3 --kernel-trace -d $O/t_${p}_$dbg -- python tools/one_conv.py 8 256 192 192 256 3 1 1 3 20 > $O/t_${p}_$dbg.log 2>&1 < /dev/null
  timeout 60 python tools/kernel_avg.py $O/t_${p}_$dbg conv16w "producers $p dbg $dbg" < /dev/null
  rm -rf $O/t_${p}_$dbg
done; done
