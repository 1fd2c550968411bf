import sqlite3, re, collections, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select dispatch_id, kernel_name, counter_name, sum(value), grid_size_x, grid_size_z, (end-start) from counters_collection group by dispatch_id, counter_name order by dispatch_id").fetchall()
d = collections.OrderedDict()
for did, kn, cn, v, gx, gz, dur in rows:
    e = d.setdefault(did, dict(name=re.sub(r'\(anonymous namespace\)::','',kn)[:52], gx=gx, gz=gz, dur=dur)); e[cn] = v
seen = set()
for did, e in d.items():
    if 'conv_igemm' not in e['name']: continue
    key = (e['name'], e['gx'], e['gz'])
    if key in seen: continue
    seen.add(key)
    mf, wc, gui = e.get('SQ_VALU_MFMA_BUSY_CYCLES',0), e.get('SQ_WAVE_CYCLES',0), e.get('GRBM_GUI_ACTIVE',0)
    print('%-52s grid %7d z%3d dur %7.1f us  MFMA-busy %.1f%%  wait_any %.2f wait_inst %.2f active %.2f  valu/mfma %.2f' % (
        e['name'], e['gx'], e['gz'], e['dur']/1e3, 100*mf/(gui/8*1024.0+1),
        e.get('SQ_WAIT_ANY',0)/(wc+1), e.get('SQ_WAIT_INST_ANY',0)/(wc+1), e.get('SQ_ACTIVE_INST_ANY',0)/(wc+1), e.get('SQ_INSTS_VALU',0)/(mf/64.0+1)))
