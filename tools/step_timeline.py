"""timeline of the last step of a rocprofv3 kernel trace (rocpd db): start offset, duration, name"""
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute('select name, start, end from kernels order by start').fetchall()
names = [re.sub(r'^void ', '', n).split('(')[0][:46] for n, _, _ in rows]
# last step: from the last k_mesh_stats_fast start back to the previous
# a step starts with the mesh statistics of the stack (k_mesh_stats_fast<2>: image + variance statistic of every frame)
starts = [i for i, n in enumerate(names) if n.startswith('k_mesh_stats_fast<2>')]
a, b = starts[-3], starts[-2]
t0 = rows[a][1]
last_end = t0
for i in range(a, b):
    n, s, e = names[i], rows[i][1], rows[i][2]
    gap = (s - last_end) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {n}')
    last_end = max(last_end, e)
print('step span', (rows[b][1] - t0) / 1e3)
