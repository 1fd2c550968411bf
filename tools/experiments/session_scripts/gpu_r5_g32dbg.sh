#!/bin/bash
# EXPERIMENT (tools/experiments/gemm32p): where the fp32 persistent GEMM's time goes -- run-time switches in an experimental library (tools/ab/libpinmem_exp.so), kernel durations from the trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
i=0
for shape in "8 512 48 48 2048 1 0 1" "8 2048 48 48 512 1 0 1"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace -d $O/r_$i -- python tools/one_conv32.py $shape 20 > $O/r_$i.log 2>&1 < /dev/null
  timeout 60 python tools/kernel_avg.py $O/r_$i conv_igemm "shape $shape shipped kernel" < /dev/null
  rm -rf $O/r_$i
  for dbg in 0 1 2 3 4 5 6 7; do
    PM_LIB=tools/ab/libpinmem_exp.so PM_GEMM32P_MIN_UNITS=1 PM_G32_DBG=$dbg timeout 120 rocprofv3 --kernel-trace -d $O/g_${i}_$dbg -- python tools/one_conv32.py $shape 20 > $O/g_${i}_$dbg.log 2>&1 < /dev/null
    timeout 60 python tools/kernel_avg.py $O/g_${i}_$dbg gemm32p "shape $shape dbg $dbg" < /dev/null
    rm -rf $O/g_${i}_$dbg
  done
done
