"""Developer tool: per-kernel mean of one rocprofv3 --pmc counter (KiB for FETCH_SIZE /
WRITE_SIZE) from counter_collection.csv files.  usage: pmc_summary.py <csv> [<csv> ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name)
    t = re.search(r'k_resample<(\d+), (\d+)>', name)
    if t:
        return f'k_resample<{t.group(1)},{t.group(2)}>'
    m = re.match(r'([A-Za-z_0-9:<>]+)', name)
    return (m.group(1) if m else name)[:48]


def main(paths):
    out = {}
    for p in paths:
        acc = defaultdict(lambda: [0.0, 0])
        with open(p) as f:
            for r in csv.DictReader(f):
                k = (short(r['Kernel_Name']), r['Counter_Name'])
                acc[k][0] += float(r['Counter_Value'])
                acc[k][1] += 1
        for (kn, cn), (tot, cnt) in acc.items():
            if kn.startswith('k_'):
                out[(kn, cn)] = (tot / cnt, cnt)
    for (kn, cn), (mean, cnt) in sorted(out.items()):
        print(f'{kn:32s} {cn:12s} mean {mean:14.1f}  launches {cnt}')


if __name__ == '__main__':
    main(sys.argv[1:])
