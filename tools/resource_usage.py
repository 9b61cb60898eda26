#!/usr/bin/env python3
"""Developer tool: registers, spills and scratch of every kernel of libzudsmi, as the compiler reports them.

    python3 tools/resource_usage.py [out.txt] [file.hip ...]

Compiles each translation unit of zuds-pipeline_amd/csrc (device code only, the build's own flags) with
`-Rpass-analysis=kernel-resource-usage` and prints one row per kernel: SGPRs, VGPRs, AGPRs, scratch bytes
per lane, occupancy (waves per SIMD), SGPR / VGPR spills, static LDS.  Rows with spills or scratch come
first.  Needs hipcc only (no GPU): VERDICT r4 item 2 asks for this table under profiles/.
"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CSRC = ROOT / 'zuds-pipeline_amd' / 'csrc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']
KEYS = [('TotalSGPRs', 'sgpr'), ('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), ('ScratchSize [bytes/lane]', 'scratch'),
        ('Occupancy [waves/SIMD]', 'occ'), ('SGPRs Spill', 'sspill'), ('VGPRs Spill', 'vspill'),
        ('LDS Size [bytes/block]', 'lds')]


def demangle(names):
    p = subprocess.run(['c++filt'], input='\n'.join(names), text=True,
                       capture_output=True)
    out = p.stdout.split('\n') if p.returncode == 0 else names
    short = []
    for d in out[:len(names)]:
        d = re.sub(r'^void ', '', d)
        d = re.sub(r'\(.*$', '', d)                      # drop the argument list
        short.append(d)
    return short


def one(src):
    cmd = ['hipcc'] + FLAGS + ['--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-c', str(src),
                              '-o', '/dev/null']
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f'{src.name}: hipcc failed\n{p.stderr[-2000:]}')
    rows, cur = [], None
    for line in p.stderr.split('\n'):
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            cur = {'file': src.name, 'name': m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for k, short in KEYS:
            m = re.search(r'remark:\s+' + re.escape(k) + r': (\S+)', line)
            if m:
                cur[short] = int(m.group(1))
    return rows


def main():
    args = [a for a in sys.argv[1:]]
    out = None
    if args and args[0].endswith('.txt'):
        out = args.pop(0)
    srcs = [CSRC / a for a in args] if args else sorted(CSRC.glob('*.hip'))
    with ThreadPoolExecutor(max_workers=6) as ex:
        rows = [r for rs in ex.map(one, srcs) for r in rs]
    for r, d in zip(rows, demangle([r['name'] for r in rows])):
        r['short'] = d
    rows.sort(key=lambda r: (-(r.get('vspill', 0) * 1000 + r.get('scratch', 0) * 10 + r.get('sspill', 0)),
                             r['file'], r['short']))
    lines = ['# kernel resource usage, hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage',
             '# (tools/resource_usage.py; kernels with spills or scratch first)',
             f'{"file":<16} {"sgpr":>4} {"vgpr":>4} {"agpr":>4} {"scratch":>7} {"occ":>3} {"s-spill":>7} '
             f'{"v-spill":>7} {"lds":>6}  kernel']
    for r in rows:
        lines.append(f'{r["file"]:<16} {r.get("sgpr", 0):>4} {r.get("vgpr", 0):>4} {r.get("agpr", 0):>4} '
                     f'{r.get("scratch", 0):>7} {r.get("occ", 0):>3} {r.get("sspill", 0):>7} '
                     f'{r.get("vspill", 0):>7} {r.get("lds", 0):>6}  {r["short"]}')
    n_spill = sum(1 for r in rows if r.get('sspill', 0) or r.get('vspill', 0) or r.get('scratch', 0))
    lines.append(f'# {len(rows)} kernels, {n_spill} with spills or scratch')
    text = '\n'.join(lines) + '\n'
    if out:
        Path(out).write_text(text)
    sys.stdout.write(text)


if __name__ == '__main__':
    main()
