"""MFMA utilisation per kernel symbol from one `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY` pass of bench.py (rocpd database).

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 256 CUs x 4 SIMDs): the gfx94x formula of derived_counters.xml
(ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots"). GRBM_GUI_ACTIVE is reported once per XCD and summed
by the query, hence the division by 8. The wait / active shares are fractions of SQ_WAVE_CYCLES (quad-cycles, disjoint buckets).

usage: pmc_mfma_summary.py results.db out.json
"""
import collections, json, re, sqlite3, sys

db, out = sys.argv[1:3]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select dispatch_id, kernel_name, counter_name, sum(value), (end-start) from counters_collection "
                   "group by dispatch_id, counter_name").fetchall()
disp = {}
for did, kn, cn, v, dur in rows:
    e = disp.setdefault(did, dict(name=re.sub(r'\(anonymous namespace\)::', '', kn).split('(')[0].replace('void ', '').strip(), dur=dur))
    e[cn] = v
agg = collections.OrderedDict()
for e in disp.values():
    a = agg.setdefault(e['name'], collections.Counter())
    a['launches'] += 1
    for k, v in e.items():
        if k != 'name':
            a[k] += v
res = []
tot = collections.Counter()
for name, a in agg.items():
    gui = a['GRBM_GUI_ACTIVE'] / 8.0
    wc = a['SQ_WAVE_CYCLES'] or 1
    r = dict(kernel=name, launches=a['launches'], total_ms=a['dur'] / 1e6,
             mfma_util=a['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024.0) if gui else None,
             wait_any=a['SQ_WAIT_ANY'] / wc, wait_inst_any=a['SQ_WAIT_INST_ANY'] / wc, active_inst_any=a['SQ_ACTIVE_INST_ANY'] / wc)
    res.append(r)
    for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'dur'):
        tot[k] += a[k]
res.sort(key=lambda r: -r['total_ms'])
conv = [r for r in res if any(k in r['kernel'] for k in ('conv_igemm', 'conv16', 'wgrad16', 'pw16_kernel'))]
conv_busy = sum(agg[r['kernel']]['SQ_VALU_MFMA_BUSY_CYCLES'] for r in conv)
conv_gui = sum(agg[r['kernel']]['GRBM_GUI_ACTIVE'] for r in conv) / 8.0
summary = dict(note='MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); counters serialise the kernels, so durations are '
                    'longer than in the un-profiled run',
               all_kernels_mfma_util=tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0),
               conv_kernels_mfma_util=conv_busy / (conv_gui * 1024.0) if conv_gui else None,
               kernels=res[:40])
json.dump(summary, open(out, 'w'), indent=1)
print('all kernels MfmaUtil %.3f   conv kernels %.3f' % (summary['all_kernels_mfma_util'], summary['conv_kernels_mfma_util'] or 0))
for r in res[:14]:
    print('%-64s n=%4d %8.2f ms  MfmaUtil %5.1f%%  wait_any %.2f wait_inst %.2f active %.2f' % (
        r['kernel'][:64], r['launches'], r['total_ms'], 100 * (r['mfma_util'] or 0), r['wait_any'], r['wait_inst_any'], r['active_inst_any']))
