"""Developer tool: where one bench step spends its time, from a rocprofv3
--kernel-trace CSV.  Takes the last `k_hp_apply`-terminated step (or the whole
trace) and prints busy time, idle gaps and the per-kernel totals in launch order."""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z_0-9:]+)', name)
    s = m.group(1) if m else name[:40]
    if s.startswith('at::native'):
        s = 'torch:' + re.sub(r'.*::', '', name.split('<')[0])
    t = re.search(r'k_resample<(\d+), (\d+)>', name)
    if t:
        s = f'k_resample<{t.group(1)},{t.group(2)}>'
    return s


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    # steps end with the last k_hp_apply of a burst; take the last full step
    ends = [i for i, r in enumerate(rows) if r[2].startswith('k_hp_apply')]
    cuts = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 50]
    if len(cuts) >= 2:
        rows = rows[cuts[-2] + 1:cuts[-1] + 2]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, gaps, cur_end = 0, [], rows[0][0]
    per = OrderedDict()
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, n))
            busy += e - s
        else:
            busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
        d = per.setdefault(n, [0, 0, 0])
        d[0] += e - s
        d[1] += 1
    for g, n in gaps:
        per[n][2] += g
    print(f'span {1e-6 * (t1 - t0):.3f} ms, busy {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms, '
          f'{len(rows)} launches')
    print(f'{"kernel":34s} {"calls":>6s} {"total ms":>9s} {"avg us":>8s} {"gap-before ms":>13s}')
    for n, (tot, cnt, gap) in sorted(per.items(), key=lambda kv: -kv[1][0] - kv[1][2]):
        print(f'{n:34s} {cnt:6d} {1e-6 * tot:9.3f} {1e-3 * tot / cnt:8.1f} {1e-6 * gap:13.3f}')
    big = sorted(gaps, reverse=True)[:8]
    print('largest gaps (us, before kernel):', [(round(g / 1e3, 1), n) for g, n in big])


if __name__ == '__main__':
    main(sys.argv[1])
