"""Developer tool: look for serialised memory accesses in the gfx950 ISA of the HIP sources.

Two patterns cost a memory (or LDS) latency each and are invisible in the source:
  * a load whose value is used at once - e.g. negated, converted or selected inside the
    condition that guards it - is followed by `s_waitcnt vmcnt(0)` / `lgkmcnt(0)` on the spot;
  * a copy loop `for (...) dst[i] = src[i]` the compiler did not unroll waits once per trip.
Both matter in latency-bound kernels (few waves per SIMD); occupancy hides them elsewhere.

    python tools/isa_lint.py [source.hip ...]      (default: every .hip under csrc/)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'zuds-pipeline_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def asm_of(src):
    out = tempfile.NamedTemporaryFile(suffix='.s', delete=False).name
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-S', '--cuda-device-only',
                    '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-o', out, src],
                   check=True, stderr=subprocess.DEVNULL)
    with open(out) as f:
        text = f.read().split('\n')
    os.unlink(out)
    return text


def kernels(text):
    cur, start = None, 0
    for i, line in enumerate(text):
        m = re.match(r'^(_Z\w+|k_\w+):', line)
        if m:
            if cur:
                yield cur, text[start:i]
            cur, start = m.group(1), i
        elif cur and 's_endpgm' in line:
            yield cur, text[start:i + 1]
            cur = None


def scan(name, body):
    vm = [i for i, l in enumerate(body) if re.search(r'\b(global|buffer|flat)_load', l)]
    ds = [i for i, l in enumerate(body) if 'ds_read' in l]
    vm_now = sum('s_waitcnt vmcnt(0)' in ' '.join(body[i + 1:i + 3]) for i in vm)
    ds_now = sum('s_waitcnt lgkmcnt(0)' in body[i + 1] for i in ds if i + 1 < len(body))
    loops = []
    for i, line in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):.*Inner Loop Header', line)
        if not m:
            continue
        lab, loads, waits = m.group(1), 0, 0
        for j in range(i + 1, min(i + 400, len(body))):
            t = body[j]
            loads += bool(re.search(r'\b(global|buffer|flat)_load', t))
            waits += 's_waitcnt vmcnt(0)' in t
            if re.search(r's_cbranch\w+\s+' + re.escape(lab) + r'\s*$', t):
                if 0 < loads <= 2 and waits:
                    loops.append((lab, loads, j - i))
                break
    return len(vm), vm_now, len(ds), ds_now, loops


def main(argv):
    srcs = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    for src in srcs:
        for name, body in kernels(asm_of(src)):
            nvm, vm_now, nds, ds_now, loops = scan(name, body)
            if vm_now >= 3 or ds_now >= 6 or loops:
                print(f'{os.path.basename(src)}: {name[:60]}')
                print(f'    global loads {nvm}, waited for at once {vm_now}; LDS reads {nds}, waited for at once {ds_now}')
                for lab, loads, length in loops:
                    print(f'    loop {lab}: {loads} load(s) and a full wait per trip ({length} lines)')


if __name__ == '__main__':
    main(sys.argv[1:])
