"""Developer tool: what the GPU does during the software-pipelined leg of bench.py (rocprofv3 --kernel-trace rocpd
database): the last `frac` of the trace cut into the time with nothing in flight, with only the kernel-fit
factorisations in flight (k_chol_tp: 9 workgroups - 3.5 % of the CUs), and with a kernel that fills the GPU in flight;
per kernel family the summed duration and the share of it spent ALONE.   usage: pipelined_trace.py results.db [frac]"""
import re
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
q = 'select name, start, end' + (', queue_id' if 'queue_id' in cols else ', 0') + ' from kernels order by start'
rows = con.execute(q).fetchall()
# the pipelined leg is the only part of a bench run whose subtractions factor on k_chol_tp: the window is the middle
# of the span of those launches (its warm-up steps and drain cut off)
tp = [(s_, e_) for n_, s_, e_, _ in rows if 'k_chol_tp' in n_]
if tp:
    w0, w1 = tp[int(len(tp) * 0.3)][0], tp[int(len(tp) * 0.9)][1]
    rows = [r for r in rows if r[1] >= w0 and r[2] <= w1]
else:
    rows = rows[int(len(rows) * (1 - frac)):]
SMALL = ('k_chol_tp', 'k_chol_back', 'k_hp_merit', 'k_hp_reject', 'k_hp_solved', 'k_rsel_scan', 'k_rsel_init', 'k_rsel_out',
         'k_hp_init_active', '__amd_rocclr', 'k_hp_diag', 'k_hp_kbasis', 'k_mesh_filter', 'k_mesh_guess', 'k_ff_vscale')


def fam(n):
    n = re.sub(r'^void ', '', n)
    return re.match(r'[A-Za-z_0-9:]+', n).group(0)


ev = []
for i, (n, s, e, qid) in enumerate(rows):
    f = fam(n)
    small = f.startswith(SMALL)
    ev.append((s, 1, small))
    ev.append((e, -1, small))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
big = small = 0
last = t0
acc = defaultdict(int)
for t, d, sm in ev:
    key = 'idle' if big + small == 0 else ('only small / latency kernels' if big == 0 else f'{min(big, 3)}{"+" if big >= 3 else ""} GPU-filling kernel(s)')
    acc[key] += t - last
    last = t
    if sm:
        small += d
    else:
        big += d
span = t1 - t0
nfused = sum(1 for r in rows if 'k_coadd_fused' in r[0])
print(f'{len(rows)} kernels over {span / 1e6:.2f} ms ({nfused} coadds: {span / 1e6 / max(nfused, 1):.2f} ms per step); queues: {len(set(r[3] for r in rows))}')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f'  {k:34s} {v / 1e6:8.2f} ms  {100.0 * v / span:5.1f} %')
per = defaultdict(lambda: [0, 0])
for n, s, e, _ in rows:
    per[fam(n)][0] += e - s
    per[fam(n)][1] += 1
print('per family: total ms (launches, mean us)')
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f'  {k:28s} {t / 1e6:8.2f} ({c}, {t / c / 1e3:.1f})')
