"""FETCH_SIZE / WRITE_SIZE per launch of tools/micro/fetch_calib against the true byte count (512 MiB per launch). usage: fetch_calib_summary.py fetch.db write.db out.json"""
import json, re, sqlite3, sys
TRUE = 512 << 20
out = {}
for path, ctr in ((sys.argv[1], 'FETCH_SIZE'), (sys.argv[2], 'WRITE_SIZE')):
    cur = sqlite3.connect(path).cursor()
    for kn, n, v in cur.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        k = re.sub(r'\(anonymous namespace\)::', '', kn).split('(')[0].replace('void ', '').strip()
        out.setdefault(k, {})[ctr + '_bytes_per_launch'] = v * 1000.0 / n      # the counter's KB = 1000 B
rows = {}
for k, d in sorted(out.items()):
    rd = k.startswith('rd_')
    c = d.get('FETCH_SIZE_bytes_per_launch' if rd else 'WRITE_SIZE_bytes_per_launch', 0.0)
    rows[k] = {'counter': 'FETCH_SIZE' if rd else 'WRITE_SIZE', 'counter_bytes_per_launch': c, 'true_bytes_per_launch': TRUE, 'counter_over_true': round(c / TRUE, 4),
               'correction_factor': round(TRUE / c, 3) if c else None}
    print('%-28s %-10s counter %8.1f MB  true %8.1f MB  counter / true %.3f' % (k, rows[k]['counter'], c / 1e6, TRUE / 1e6, c / TRUE))
json.dump({'note': 'each launch streams 512 MiB once; correction_factor = multiply the counter by this to get bytes', 'kernels': rows}, open(sys.argv[3], 'w'), indent=1)
