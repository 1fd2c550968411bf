#!/bin/bash
# memory read / write / read-loss path: HIP-event timing + FETCH_SIZE / WRITE_SIZE passes of tools/mem_probe.py
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 300 python tools/mem_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/mem_probe.txt; cp gpurun_out/mem_probe.json $O/mem_probe.json
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -- python tools/mem_probe.py > $O/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -- python tools/mem_probe.py > $O/w.log 2>&1
python - $(find $O/f -name '*.db' | head -1) $(find $O/w -name '*.db' | head -1) $O/mem_path_hbm_counters.json <<'PY'
import sqlite3, sys, json, re, collections
res = collections.OrderedDict()
for path, ctr in ((sys.argv[1], 'FETCH_SIZE'), (sys.argv[2], 'WRITE_SIZE')):
    cur = sqlite3.connect(path).cursor()
    for kn, n, v in cur.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        k = re.sub(r'\(anonymous namespace\)::', '', kn).split('(')[0].replace('void ', '').strip()
        res.setdefault(k, {})[ctr + '_KB_per_launch'] = v * 1.024 / n      # the counters report KiB (profiles/r04a_fetch_calibration.json); kept in units of 1000 bytes
for k, d in res.items():
    d['hbm_MB_per_launch'] = (2 * d.get('FETCH_SIZE_KB_per_launch', 0) + d.get('WRITE_SIZE_KB_per_launch', 0)) / 1e3
keep = {k: d for k, d in res.items() if any(s in k for s in ('mem_', 'ce_', 'pm_copy', 'reduce_partials'))}
import os
try:
    stamp = open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'pinthememory_amd', 'libpinmem_hip.so.stamp')).read().strip()
except OSError:
    stamp = None
json.dump(dict(lib_stamp=stamp, note='read = 2 x FETCH_SIZE (gfx950), KB = 1000 B (converted from the counters\' KiB); separate --pmc passes of tools/mem_probe.py', kernels=keep), open(sys.argv[3], 'w'), indent=1)
for k, d in keep.items():
    print('%-44s hbm %8.2f MB/launch' % (k[:44], d['hbm_MB_per_launch']))
PY
find $O -name '*.db' -delete
