"""Idle time between kernels of the last forward in a rocprofv3 --kernel-trace database: python tools/gaps.py results.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute('select name, start, end from kernels order by start').fetchall()
# the last forward = from the last conv1 launch triple backwards: take the last 60 kernels that follow the last 'pack' / setup kernel
names = [r[0] for r in rows]
last_sm = max(i for i, n in enumerate(names) if 'softmax_argmax' in n)
first = max(i for i, n in enumerate(names[:last_sm - 5]) if 'softmax_argmax' in n) + 1
seg = rows[first:last_sm + 1]
busy = sum(e - s for _n, s, e in seg)
span = seg[-1][2] - seg[0][1]
print('kernels %d  busy %.3f ms  span %.3f ms  idle %.3f ms' % (len(seg), busy / 1e6, span / 1e6, (span - busy) / 1e6))
gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0][:40], seg[i + 1][0][:40]) for i in range(len(seg) - 1))
for g in gaps[-12:]:
    print('%8.1f us  after %-40s before %s' % g)
