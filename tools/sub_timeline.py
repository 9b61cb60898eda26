"""Developer tool: one subtraction leg of a rocprofv3 kernel trace (rocpd db), kernel by kernel, from the end of a
fused coadd kernel on: start, length, gap to the previous end on ANY queue, queue and stream ids.
usage: sub_timeline.py trace.db [rows]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 45
cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
sel = 'select name, start, end, queue_id, stream_id from kernels order by start' if 'stream_id' in cols else \
    'select name, start, end, queue_id, 0 from kernels order by start'
rows = con.execute(sel).fetchall()
names = [re.sub(r'^void ', '', n).split('(')[0][:40] for n, *_ in rows]
starts = [i for i, n in enumerate(names) if n.startswith('k_coadd_fused')]
a = starts[len(starts) // 2]
t0 = rows[a][2]
last = rows[a][1]
for i in range(a, min(a + nrows, len(rows))):
    n, s, e, q, st = rows[i]
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - last) / 1e3:7.1f}  q{q} s{st}  {names[i]}')
    last = max(last, e)
