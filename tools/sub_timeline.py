import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
rows = con.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall() if 'stream_id' in cols else con.execute('select name, start, end, queue_id, 0 from kernels order by start').fetchall()
names = [re.sub(r'^void ', '', n).split('(')[0][:40] for n, *_ in rows]
starts = [i for i, n in enumerate(names) if n.startswith('k_coadd_fused')]
a = starts[len(starts)//2]
t0 = rows[a][2]
for i in range(a, min(a + 45, len(rows))):
    n, s, e, q, st = rows[i]
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{q} s{st}  {names[i]}')
