#!/bin/bash
# Build the library of a git revision (default HEAD) into ab/libpinmem_base.so for same-box A/B runs (PM_LIB=ab/libpinmem_base.so).
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); mkdir -p $ROOT/ab $T/csrc $T/include
git -C $ROOT archive $REV pinthememory_amd/csrc include | tar -x -C $T
OBJS=""
for s in misc conv_igemm winograd bn pool_resize loss memory; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -c $T/pinthememory_amd/csrc/$s.hip -o $T/$s.o &
  OBJS="$OBJS $T/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/ab/libpinmem_base.so && echo "built ab/libpinmem_base.so from $REV"
rm -rf $T
