#!/bin/bash
# round 6: per-shape timings of the fp32 forward with the persistent split GEMM on (PM_SPLITP=1) / off (0): the 3x3 256->256 at 192^2, the 1x1 512->2048 at 48^2, the 1x1 1024->256 at 48^2, the 1x1 256->64 at 192^2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for sp in 1 0; do
  echo "PM_SPLITP=$sp"
  for shape in "8 256 192 192 256 3 1 1" "8 512 48 48 2048 1 0 1" "8 1024 48 48 256 1 0 1" "8 2048 48 48 512 1 0 1" "8 256 192 192 64 1 0 1" "8 64 192 192 256 1 0 1"; do
    PM_SPLITP=$sp timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1
  done
done | tee $O/shapes.log
