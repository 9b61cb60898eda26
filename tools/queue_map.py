"""Developer tool: which hardware queue each stream's kernels ran on (rocprofv3 kernel trace, rocpd db), per kernel name
pattern.  usage: queue_map.py trace.db [pattern]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else 'k_chol_tp_b'
rows = con.execute('select stream_id, queue_id, count(*), min(start), max(end) from kernels where name like ? '
                   'group by stream_id, queue_id order by min(start)', (f'%{pat}%',)).fetchall()
t0 = con.execute('select min(start) from kernels').fetchone()[0]
for s, q, n, a, b in rows:
    print(f'stream {s:3d}  queue {q:3d}  {n:5d} launches  {1e-6 * (a - t0):9.1f} .. {1e-6 * (b - t0):9.1f} ms')
