"""Developer tool: concurrency in a rocprofv3 --kernel-trace database (rocpd SQLite): wall span of the
last `frac` of the kernels, sum of kernel durations, time with 0 / 1 / 2 / ... kernels in flight, and
the mean duration per kernel name.   usage: rocpd_overlap.py results.db [frac]"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = con.execute('select name, start, end from kernels order by start').fetchall()
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = []
for n, s, e in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
tot = sum(e - s for _, s, e in rows)
print(f'{len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms, kernel time {tot / 1e6:.2f} ms, mean depth {tot / (t1 - t0):.2f}')
print('time by kernels in flight:', {k: f'{100.0 * v / (t1 - t0):.1f} %' for k, v in sorted(hist.items())})
per = defaultdict(lambda: [0, 0])
for n, s, e in rows:
    k = n.split('(')[0].replace('void ', '')[:40]
    per[k][0] += e - s
    per[k][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f'  {k:40s} {c:5d} x {t / c / 1e3:8.1f} us = {t / 1e6:7.2f} ms')
