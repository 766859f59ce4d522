#!/usr/bin/env python
"""Turn rocprofv3's rocpd SQLite output (gpurun_out/...) into the small text summaries that
are committed under profiles/.

  python profiles/summarize.py stats gpurun_out/prof_r1/r1_results.db  > profiles/rNN_kernel_stats.csv
  python profiles/summarize.py pmc   gpurun_out/pmc_x/x_results.db ... > profiles/rNN_pmc.csv
  python profiles/summarize.py pmc_min 10000 db...   (only dispatches of >= 10 ms: the dominant 60x90 launches)
  python profiles/summarize.py table profiles/rNN_pmc.csv [profiles/rNN_kernel_stats.csv [forwards images]]   (markdown: bytes, us, TB/s per kernel; per-step totals)

`stats` = per-kernel calls / total / average / share (what `--kernel-trace --stats` tabulates);
`pmc`   = per-kernel, per-counter mean value per dispatch (one --pmc pass per database).
Durations are nanoseconds in the database and are written as microseconds."""
import re
import sqlite3
import sys


def demangle_light(name):
    """rocprofv3 leaves some template instantiations mangled; keep the function name and mark bf16 (DF16b)."""
    m = re.match(r'_ZN3jcm(?:12_GLOBAL__N_1)?', name)
    if not m:
        return name
    parts, pos = [], m.end()
    while True:      # nested names: <length><identifier>... up to the E / I that ends them (jcm::cfft::rows_inv_mfma_kernel)
        d = re.match(r'\d+', name[pos:])
        if not d:
            break
        n = int(d.group(0))
        parts.append(name[pos + d.end():pos + d.end() + n])
        pos += d.end() + n
    if not parts:
        return name
    return '::'.join(parts) + ('<bf16>' if 'DF16b' in name and len(parts) == 1 else '')


def short(name):
    name = demangle_light(name)
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '').replace('jcm::', '')


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                       'from kernels group by name order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows)
    print('kernel,calls,total_us,avg_us,min_us,max_us,percent')
    for n, c, s, a, mn, mx in rows:
        print('"%s",%d,%.1f,%.1f,%.1f,%.1f,%.2f' % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))


def pmc(dbs, min_us=0.0):
    """min_us > 0: only dispatches at least that long (e.g. the two 60x90 launches of the strip kernel, which shares its
    name with the shorter 30x45 launch)."""
    print('kernel,counter,dispatches,mean_value_per_dispatch,mean_duration_us')
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                           'where duration >= ? group by kernel_name, counter_name order by avg(duration)*count(*) desc', (min_us * 1e3,)).fetchall()
        for n, cn, c, v, d in rows:
            if d * c < 2e5:      # skip kernels below 0.2 ms total
                continue
            print('"%s",%s,%d,%.6g,%.1f' % (short(n), cn, c, v, d / 1e3))


def trace(db, pattern, last=0):
    """Dispatches of the kernels whose name contains `pattern`, in time order (the last `last` of them if > 0): which layer costs what."""
    con = sqlite3.connect(db)
    rows = con.execute('select name, start, duration from kernels where name like ? order by start', ('%' + pattern + '%',)).fetchall()
    if last:
        rows = rows[-last:]
    print('kernel,start_us,duration_us')
    t0 = rows[0][1] if rows else 0
    for n, st, d in rows:
        print('"%s",%.1f,%.1f' % (short(n), (st - t0) / 1e3, d / 1e3))


def table(pmc_csv, stats_csv=None, forwards=0, images=1):
    """Markdown table of the HBM-side figures per kernel from a committed pmc CSV: bytes per dispatch = 2 x FETCH_SIZE + WRITE_SIZE
    (both in KB; FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 bytes), mean dispatch time of the kernel-trace run if
    `stats_csv` is given (counter passes run slower), achieved TB/s, clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and MFMA-busy share."""
    import csv
    ctr = {}
    for r in csv.DictReader(open(pmc_csv)):
        ctr.setdefault(r['kernel'], {})[r['counter']] = (float(r['mean_value_per_dispatch']), float(r['mean_duration_us']), int(r['dispatches']))
    avg = {}
    if stats_csv:
        for r in csv.DictReader(open(stats_csv)):
            avg[r['kernel']] = (float(r['avg_us']), int(r['calls']))
    rows = []
    for k, c in ctr.items():
        if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
            continue
        rd, wr = 2 * c['FETCH_SIZE'][0] * 1024, c['WRITE_SIZE'][0] * 1024
        us = avg.get(k, (c['GRBM_GUI_ACTIVE'][1] if 'GRBM_GUI_ACTIVE' in c else c['FETCH_SIZE'][1], 0))[0]
        n = c['FETCH_SIZE'][2]
        ghz = c['GRBM_GUI_ACTIVE'][0] / 8.0 / (c['GRBM_GUI_ACTIVE'][1] * 1e3) if 'GRBM_GUI_ACTIVE' in c else float('nan')
        busy = ''
        if c.get('SQ_VALU_MFMA_BUSY_CYCLES', (0,))[0] > 0 and 'SQ_BUSY_CU_CYCLES' in c:
            busy = '%.0f %%' % (100.0 * c['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (4.0 * c['SQ_BUSY_CU_CYCLES'][0]))
        rows.append((us * n, k, n, us, rd, wr, (rd + wr) / (us * 1e-6) / 1e12, ghz, busy))
    print('| kernel | dispatches | mean us | read MB | written MB | TB/s | GHz | MFMA busy |')
    print('|---|---|---|---|---|---|---|---|')
    for _, k, n, us, rd, wr, tbs, ghz, busy in sorted(rows, reverse=True):
        print('| `%s` | %d | %.1f | %.0f | %.0f | %.2f | %.2f | %s |' % (k, n, us, rd / 1e6, wr / 1e6, tbs, ghz, busy))
    if forwards:
        # per step: everything except the once-per-engine filter-spectra kernels
        step = [r for r in rows if 'weight_spectra' not in r[1] and 'weight_bound' not in r[1]]
        b = sum(r[2] * (r[4] + r[5]) for r in step) / forwards
        t = sum(r[2] * r[3] for r in step) / forwards
        print()
        print('**Per step** (%d forwards in the run, %d images per step; filter-spectra kernels excluded): **%.2f GB = %.0f MB per image**, %.2f ms of kernel time '
              '-> %.2f TB/s = %.0f %% of the 8 TB/s HBM peak.' % (forwards, images, b / 1e9, b / images / 1e6, t / 1e3, b / (t * 1e-6) / 1e12, b / (t * 1e-6) / 8e12 * 100))
    return rows


if __name__ == '__main__':
    if sys.argv[1] == 'table':            # table pmc.csv [kernel_stats.csv]
        table(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, int(sys.argv[4]) if len(sys.argv) > 4 else 0, int(sys.argv[5]) if len(sys.argv) > 5 else 1)
    elif sys.argv[1] == 'trace':            # trace db pattern [last]
        trace(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2])
    elif sys.argv[1] == 'pmc_min':          # pmc_min MIN_US db...
        pmc(sys.argv[3:], float(sys.argv[2]))
    else:
        pmc(sys.argv[2:])
