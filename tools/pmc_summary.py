"""Developer tool: per (kernel, grid size) mean of one rocprofv3 --pmc counter (KiB for FETCH_SIZE /
WRITE_SIZE) from counter_collection.csv files.  A kernel that is launched on batches of different
sizes (the mesh statistics: the 32 frames of a stack, then the one science frame of the subtraction)
shows one line per grid size - the mean over both would describe neither.
usage: pmc_summary.py <csv> [<csv> ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name)
    t = re.search(r'k_resample<(\d+), (\d+)>', name)
    if t:
        return f'k_resample<{t.group(1)},{t.group(2)}>'
    t = re.match(r'(k_coadd_fused_dma|k_coadd_fused_own)<(\d+), (\w+), (\w+), (\w+)>', name)
    if t:     # <MOP, AVG, STACK, DEV>
        return f'{t.group(1)}<{"stack" if t.group(4) == "true" else "sum"}>'
    m = re.match(r'([A-Za-z_0-9:<>]+)', name)
    return (m.group(1) if m else name)[:48]


def main(paths):
    out = {}
    for p in paths:
        acc = defaultdict(lambda: [0.0, 0])
        with open(p) as f:
            for r in csv.DictReader(f):
                k = (short(r['Kernel_Name']), int(r['Grid_Size']), r['Counter_Name'])
                acc[k][0] += float(r['Counter_Value'])
                acc[k][1] += 1
        for (kn, gs, cn), (tot, cnt) in acc.items():
            if kn.startswith('k_'):
                out[(kn, gs, cn)] = (tot / cnt, cnt)
    for (kn, gs, cn), (mean, cnt) in sorted(out.items()):
        print(f'{kn:32s} grid {gs:10d} {cn:20s} mean {mean:16.1f}  launches {cnt}')


if __name__ == '__main__':
    main(sys.argv[1:])
