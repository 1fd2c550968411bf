#!/bin/bash
# round 4: FETCH / WRITE calibration by access width + the one-rank RCCL rehearsal of the N > 1 path. usage: gpu_r4_misc.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
# the calibration binary is git-ignored: build it here when it is missing, fail loudly when that is impossible
if [ ! -x tools/micro/fetch_calib ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib || { echo "cannot build tools/micro/fetch_calib" >&2; exit 1; }
fi
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cal_f -- tools/micro/fetch_calib > $O/cal_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cal_w -- tools/micro/fetch_calib > $O/cal_w.log 2>&1
DBF=$(find $O/cal_f -name '*.db' | head -1); DBW=$(find $O/cal_w -name '*.db' | head -1)
if [ -n "$DBF" ] && [ -n "$DBW" ]; then python tools/fetch_calib_summary.py "$DBF" "$DBW" $O/fetch_calibration.json; else echo "no counter databases: see $O/cal_f.log $O/cal_w.log" >&2; fi
find $O -name '*.db' -delete
python tools/cpu_enqueue_time.py 2>&1 | grep -v amdgpu | tail -4 > $O/enqueue_plain.txt; cat $O/enqueue_plain.txt
PM_DIST_FORCE=1 python tools/cpu_enqueue_time.py 2>&1 | grep -v amdgpu | tail -4 > $O/enqueue_one_rank_rccl.txt; cat $O/enqueue_one_rank_rccl.txt
PM_DIST_FORCE=1 timeout 600 rocprofv3 --kernel-trace -d $O/kt_rccl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-side > $O/kt_rccl.log 2>&1
python tools/rocpd_stats.py $(find $O/kt_rccl -name '*.db' | head -1) $O/kernel_stats_one_rank_rccl.csv 4 | head -2
grep -i "nccl\|rccl\|AllReduce\|AllGather" $O/kernel_stats_one_rank_rccl.csv | cut -c1-200
find $O -name '*.db' -delete
