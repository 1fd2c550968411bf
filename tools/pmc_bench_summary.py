"""Summarise FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, separate runs) of bench.py into per-kernel HBM bytes per launch.
FETCH_SIZE is doubled (gfx950 counts 128-B read requests at 64 B, MI355X_MICROARCH.md section HBM). Calibrated in round 4 on 512 MiB streams of known size
(tools/micro/fetch_calib.hip, profiles/r04a_fetch_calibration.json): FETCH_SIZE x 2 and WRITE_SIZE x 1 are exact for 4-, 8- (the int64 label read) and 16-byte
lanes and for the 4-byte accumulator-layout store, and the counters' "KB" is 1024 bytes -- rounds 1-3 multiplied by 1000 (2.4 % low); from r04 on the MB / GB
figures below are in units of 10^6 / 10^9 bytes computed from KiB."""
import collections, json, os, re, sqlite3, sys
fetch_db, write_db, out = sys.argv[1:4]
try:
    stamp = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pinthememory_amd', 'libpinmem_hip.so.stamp')).read().strip()
except OSError:
    stamp = None
res = collections.OrderedDict()
for path, ctr in ((fetch_db, 'FETCH_SIZE'), (write_db, 'WRITE_SIZE')):
    cur = sqlite3.connect(path).cursor()
    for kn, n, v in cur.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        k = re.sub(r'\(anonymous namespace\)::', '', kn).split('(')[0]
        res.setdefault(k, {})[ctr] = dict(total_KB=v * 1.024, launches=n)      # KiB -> units of 1000 bytes
rows = []
for k, d in res.items():
    f, w = d.get('FETCH_SIZE', {}), d.get('WRITE_SIZE', {})
    rows.append(dict(kernel=k, launches=f.get('launches', w.get('launches', 0)), read_MB_per_launch=2 * f.get('total_KB', 0) / max(f.get('launches', 1), 1) / 1e3,
                     write_MB_per_launch=w.get('total_KB', 0) / max(w.get('launches', 1), 1) / 1e3,
                     read_GB_total=2 * f.get('total_KB', 0) / 1e6, write_GB_total=w.get('total_KB', 0) / 1e6))
rows.sort(key=lambda r: -(r['read_GB_total'] + r['write_GB_total']))
json.dump(dict(lib_stamp=stamp, note='2 steps (1 warm-up + 1 timed); read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE; the counters report KiB (calibrated: profiles/r04a_fetch_calibration.json), figures here are 10^6 / 10^9 bytes',
               total_read_GB=sum(r['read_GB_total'] for r in rows), total_write_GB=sum(r['write_GB_total'] for r in rows), kernels=rows[:40]), open(out, 'w'), indent=1)
print('total read %.1f GB write %.1f GB over 2 steps' % (sum(r['read_GB_total'] for r in rows), sum(r['write_GB_total'] for r in rows)))
for r in rows[:12]:
    print('%-70s n=%4d read %8.1f MB/launch write %8.1f MB/launch' % (r['kernel'][:70], r['launches'], r['read_MB_per_launch'], r['write_MB_per_launch']))
