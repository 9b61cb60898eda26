"""Developer tool: static instruction counts of one kernel per source line and per basic block.

Compile a .hip file with line tables (`hipcc -O3 -gline-tables-only -S --cuda-device-only`) and
attribute every instruction of the named kernel to the source line its `.loc` directive carries:
VALU / SALU / LDS / VMEM / other per line, readlane / writelane (SGPR spills) apart.  Static counts:
an instruction inside a loop counts once - read it next to the trip counts.

    python tools/isa_lines.py file.s kernel_substring [--blocks] [--range lo hi]
"""
import re
import sys
from collections import defaultdict


def classify(op):
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'):
        return 'lane'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    return 'other'


def main():
    path, name = sys.argv[1], sys.argv[2]
    blocks = '--blocks' in sys.argv
    lo, hi = 0, 10 ** 9
    if '--range' in sys.argv:
        i = sys.argv.index('--range')
        lo, hi = int(sys.argv[i + 1]), int(sys.argv[i + 2])
    text = open(path).read().split('\n')
    start = None
    for i, l in enumerate(text):
        if l.startswith('_Z') and name in l and l.rstrip().split(':')[0].startswith('_Z') and ':' in l:
            start = i
            break
    assert start is not None, 'kernel not found'
    per_line = defaultdict(lambda: defaultdict(int))
    per_block = []
    cur_line = 0
    cur_block = ['entry', defaultdict(int), set()]
    files = {}
    for l in text[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('.end_amdhsa_kernel'):
            break
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur_line = int(m.group(2))
            continue
        if re.match(r'^\.LBB\d+_\d+:', s):
            per_block.append(cur_block)
            cur_block = [s.split(':')[0], defaultdict(int), set()]
            continue
        if not s or s.startswith('.') or s.startswith(';') or s.startswith('//'):
            continue
        op = s.split()[0]
        k = classify(op)
        if 'pk_' in op:
            per_line[cur_line]['pk'] += 1
            cur_block[1]['pk'] += 1
        per_line[cur_line][k] += 1
        cur_block[1][k] += 1
        cur_block[2].add(cur_line)
    per_block.append(cur_block)
    keys = ['valu', 'pk', 'lane', 'salu', 'smem', 'lds', 'vmem', 'wait', 'branch']
    tot = defaultdict(int)
    if blocks:
        print('%-14s' % 'block' + ''.join('%7s' % k for k in keys) + '  lines')
        for nm, c, ls in per_block:
            if sum(c.values()) < 8:
                continue
            ls = sorted(x for x in ls if x)
            print('%-14s' % nm + ''.join('%7d' % c[k] for k in keys) + '  %s..%s' % (ls[0] if ls else 0, ls[-1] if ls else 0))
        return
    print('%6s' % 'line' + ''.join('%7s' % k for k in keys))
    for ln in sorted(per_line):
        c = per_line[ln]
        if lo <= ln <= hi:
            print('%6d' % ln + ''.join('%7d' % c[k] for k in keys))
            for k in keys:
                tot[k] += c[k]
    print('%6s' % 'sum' + ''.join('%7d' % tot[k] for k in keys))


if __name__ == '__main__':
    main()
