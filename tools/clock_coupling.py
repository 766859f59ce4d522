"""Effective clock of a kernel and of its successor from a `rocprofv3 --pmc GRBM_GUI_ACTIVE` database of tools/ab_rows_mfma.py (the two arms of
"fft_rows_mfma" interleaved): GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration.   python tools/clock_coupling.py results.db > profiles/r06_clock_coupling.csv"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]; pm = [t for t in tabs if 'pmc_event' in t][0]
val = collections.defaultdict(float)
for eid, v in db.execute('select event_id, value from %s' % pm):
    val[eid] += v
rows = db.execute('select s.kernel_name, d.start, d.end, d.event_id from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()
seq = [(n, (e - s) / 1e3, val.get(eid, 0.0)) for n, s, e, eid in rows]
agg = collections.defaultdict(list)
for i, (n, us, v) in enumerate(seq):
    if 'kxfold' in n and us > 500 and i > 0:
        pn, pus, pv = seq[i - 1]
        arm = 'rows_inv_mfma_kernel' if 'rows_inv_mfma' in pn else 'rows_inv_reg_kernel<96, 2, true>' if 'rows_inv_reg' in pn else None
        if arm:
            agg[arm].append((pus, pv / 8 / pus / 1e3, us, v / 8 / us / 1e3))
print('inverse row kernel of conv5,launches,its mean us,its GHz,conv_kxfold_bf16_kernel behind it: mean us,GHz')
for arm, l in sorted(agg.items()):
    n = len(l)
    print('"%s",%d,%.1f,%.3f,%.1f,%.3f' % (arm, n, sum(x[0] for x in l) / n, sum(x[1] for x in l) / n, sum(x[2] for x in l) / n, sum(x[3] for x in l) / n))
