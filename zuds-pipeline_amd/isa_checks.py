"""Static checks of the gfx950 assembly the build produced (host-side build tooling: no GPU, nothing of it ships in
libzudsmi).  ``build.py`` compiles every translation unit with ``-save-temps=obj`` and runs both checks on the device
assembly that was assembled into the object - the flags of the build, not a second compile - and fails the build on
a hit; ``tests/test_isa_lint.py`` runs them on the same files.

1. Counted LDS pipelines (``lgkm_lint``).  The fused coadd kernels read their filter windows with inline-asm ``ds_read``
   sequences and wait for them with COUNTED ``s_waitcnt lgkmcnt(N)``.  lgkmcnt also counts scalar loads and any LDS
   operation the compiler emits, and scalar loads return out of order: ONE such instruction between two counted waits
   lets a wait pass early and the pixel work on stale registers - wrong values on some waves of some launches, no
   fault (round 5 met exactly that).  The sources bracket each counted region with ``; ZM_LGKM_BEGIN`` / ``; ZM_LGKM_END``
   comments; between them only the kernels' own asm statements may touch the lgkm counter.

2. The DPP read hazard (``dpp_lint``).  The Cholesky kernels broadcast fp64 values with ``v_mov_b64_dpp`` /
   ``v_fmac_f64_dpp`` (row_newbcast) written as inline asm.  gfx950 does not interlock a DPP operand against a vector
   instruction that wrote it less than two wait states earlier: the broadcast would read the register's OLD value.
   The asm statements keep their own distance; what they cannot see is an instruction the compiler puts between them
   (a register copy, a re-materialised constant).  For every DPP instruction, no vector instruction among the
   preceding two wait states may write its DPP source.  Round 6: a basic-block label is a join - what the
   other predecessor executed last is not in the text above it - so the window restarts there as UNKNOWN: a DPP read
   within two wait states of a label is reported unless the instructions between cover the distance.
"""
import re

# what increments lgkmcnt: LDS / GDS, scalar memory, messages; flat accesses count in both counters
LGKM = re.compile(r'^(ds_|s_load|s_buffer_load|s_scratch_load|s_store|s_buffer_store|s_dcache|s_sendmsg|s_memtime|'
                  r's_memrealtime|s_atc_probe|flat_)')
VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
DPP_OPS = ('v_fmac_f64_dpp', 'v_mov_b64_dpp')


def lgkm_lint(asm_text):
    """([(kernel, line number, instruction)] of compiler-made lgkm operations inside a counted region, regions seen)."""
    bad, regions = [], 0
    kernel, inside, in_asm = None, False, False
    for n, line in enumerate(asm_text.split('\n'), 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            kernel = m.group(1)
        if 'ZM_LGKM_BEGIN' in t:
            assert not inside, f'nested ZM_LGKM_BEGIN at line {n}'
            inside = True
            regions += 1
            continue
        if 'ZM_LGKM_END' in t:
            assert inside, f'ZM_LGKM_END without a BEGIN at line {n}'
            inside = False
            continue
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if inside and not in_asm and LGKM.match(t):
            bad.append((kernel, n, t))
    assert not inside, 'ZM_LGKM_BEGIN without an END'
    return bad, regions


def vregs(operand):
    m = VREG.search(operand)
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def dpp_lint(asm_text):
    """([(kernel, line number, dpp instruction, offending writer or 'label')], number of DPP instructions seen)."""
    bad, ndpp = [], 0
    kernel = None
    window = []        # (wait states the instruction is worth, registers it writes if it is a VALU op - None: unknown -, text)
    for n, line in enumerate(asm_text.split('\n'), 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            kernel, window = m.group(1), []
            continue
        if re.match(r'^\.L\w+:', t):
            # a join: the other way in may have written anything with its last instruction
            window = [(0, None, f'label {t.split(":")[0]}')]
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        args = t[len(op):].split(';')[0]
        ops = [a.strip() for a in args.split(',')]
        if op in DPP_OPS:
            ndpp += 1
            src = vregs(ops[1].lstrip('-|'))
            need = 2
            for ws, wr, txt in reversed(window):
                if need <= 0:
                    break
                if wr is None or (wr & src):
                    bad.append((kernel, n, t, txt))
                    break
                need -= ws
        if op == 's_nop':
            window.append((int(ops[0]) + 1, set(), t))
        elif op.startswith('v_') and not op.startswith('v_cmp') and not op.startswith('v_readlane') \
                and not op.startswith('v_readfirstlane'):
            window.append((1, vregs(ops[0]), t))          # a VALU instruction: its destination is its first operand
        else:
            window.append((1, set(), t))
        window = window[-8:]
    return bad, ndpp


def check_file(path):
    """Both checks on one assembly file -> (list of findings as strings, dict of counts)."""
    with open(path) as f:
        text = f.read()
    out = []
    bad, regions = lgkm_lint(text)
    out += [f'{path}:{n}: {k}: compiler-made lgkm operation inside a counted pipeline: {t}' for k, n, t in bad]
    bad, ndpp = dpp_lint(text)
    out += [f'{path}:{n}: {k}: DPP source written {w!r} less than two wait states before: {t}' for k, n, t, w in bad]
    return out, dict(lgkm_regions=regions, dpp=ndpp)
