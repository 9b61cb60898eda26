"""Developer tool: per-kernel summary (calls, total, average, share) of a rocprofv3
--kernel-trace results database (rocpd SQLite), as CSV on stdout or to a file.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv] [--top N]
"""
import csv
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 0
con = sqlite3.connect(args[0])
rows = con.execute('select name, count(*), sum(end - start), avg(end - start), min(end - start), '
                   'max(end - start) from kernels group by name order by 3 desc').fetchall()
# the median too: the first launches of a process run cold (page faults, code upload) and pull the mean up
med = {}
for name, dur in con.execute('select name, end - start from kernels order by name, 2'):
    med.setdefault(name, []).append(dur)
med = {k: v[len(v) // 2] for k, v in med.items()}
total = sum(r[2] for r in rows) or 1
out = open(args[1], 'w', newline='') if len(args) > 1 else sys.stdout
w = csv.writer(out)
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'MedianNs'])
for r in rows:
    w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 2), r[4], r[5], med.get(r[0])])
if top:
    for r in rows[:top]:
        print(f'{r[0][:60]:60s} {r[1]:6d} calls {r[3] / 1e3:9.1f} us avg {med.get(r[0], 0) / 1e3:9.1f} us median {100.0 * r[2] / total:6.2f} %', file=sys.stderr)
