"""Static check of the hand-counted LDS pipelines of the fused coadd kernels (CPU: needs hipcc, no GPU).

k_coadd_fused_own / _dma / k_coadd_fused read their filter windows (and, in the register-staged form, the tap-table
nodes) with inline-asm ``ds_read`` sequences and wait for them with COUNTED ``s_waitcnt lgkmcnt(N)``: row r + 1 is
in flight while row r is applied.  lgkmcnt also counts scalar loads and any LDS operation the compiler emits, and
scalar loads return out of order: ONE such instruction between two counted waits lets a wait pass early and the
pixel work on stale registers - wrong values on some waves of some launches, no fault.  Round 5 met exactly that:
a C++ ``if`` on a kernel argument inside the loop made the compiler sink the argument's ``s_load`` there; the
k_resample comparison caught it (nondeterministic differences in the last pixel of every group).

The sources bracket each counted region with ``; ZM_LGKM_BEGIN`` / ``; ZM_LGKM_END`` comments; between them only the
kernels' own asm statements (``;;#ASMSTART`` ... ``;;#ASMEND``) may touch the lgkm counter."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'zuds-pipeline_amd', 'csrc', 'resample.hip')
# what increments lgkmcnt: LDS / GDS, scalar memory, messages; flat accesses count in both counters
LGKM = re.compile(r'^(ds_|s_load|s_buffer_load|s_scratch_load|s_store|s_buffer_store|s_dcache|s_sendmsg|s_memtime|'
                  r's_memrealtime|s_atc_probe|flat_)')


def lint(asm_text):
    """[(kernel, line number, instruction)] of compiler-made lgkm operations inside a counted region."""
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


def test_lint_sees_a_planted_scalar_load():
    txt = '\n'.join(['_Z1kv:', '\t; ZM_LGKM_BEGIN', '\t;;#ASMSTART', '\tds_read_b64 v[0:1], v2', '\t;;#ASMEND',
                     '\ts_load_dword s4, s[0:1], 0x10', '\tv_fma_f32 v0, v1, v2, v3', '\t; ZM_LGKM_END'])
    bad, regions = lint(txt)
    assert regions == 1 and [b[2].split()[0] for b in bad] == ['s_load_dword']


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_no_compiler_made_lgkm_operation_inside_the_counted_pipelines(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    out = tmp_path / 'resample.s'
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', SRC,
                           '-o', str(out)], stderr=subprocess.DEVNULL)
    bad, regions = lint(out.read_text())
    # every instance of the three kernels carries one region (the register-staged form wraps its tap loop too)
    assert regions >= 3 * 9, regions
    assert not bad, bad[:5]


# ---- the DPP read hazard of the Cholesky kernels (round 5) ---------------------------------------------------------
# csrc/chol_diag.h and the panel chains of csrc/hotpants.hip broadcast fp64 values with v_mov_b64_dpp / v_fmac_f64_dpp
# (row_newbcast) written as inline asm.  gfx950 does not interlock a DPP operand against a vector instruction that wrote
# it less than two wait states earlier: the broadcast would read the register's OLD value - no fault, a wrong factor
# on some lanes.  The asm statements keep their own distance; what they cannot see is an instruction the compiler
# puts between them (a register copy, a re-materialised constant).  This lint walks the assembly: for every DPP
# instruction, no vector instruction among the preceding two wait states may write its DPP source.
HOT = os.path.join(ROOT, 'zuds-pipeline_amd', 'csrc', 'hotpants.hip')
VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def vregs(operand):
    m = VREG.search(operand)
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def dpp_lint(asm_text):
    """[(kernel, line number, dpp instruction, offending writer)]; also the number of DPP instructions seen."""
    bad, ndpp = [], 0
    kernel = None
    window = []                  # (wait states this instruction is worth, registers it writes if it is a VALU op, text)
    for n, line in enumerate(asm_text.split('\n'), 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            kernel, window = m.group(1), []
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        args = t[len(op):].split(';')[0]
        ops = [a.strip() for a in args.split(',')]
        if op in ('v_fmac_f64_dpp', 'v_mov_b64_dpp'):
            ndpp += 1
            src = vregs(ops[1].lstrip('-|'))
            need = 2
            for ws, wr, txt in reversed(window):
                if need <= 0:
                    break
                if wr & src:
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


def test_dpp_lint_sees_a_planted_hazard_and_accepts_the_distance():
    head = ['_Z1kv:', '\tv_mul_f64 v[2:3], v[6:7], v[8:9]']
    dpp = '\tv_fmac_f64_dpp v[0:1], v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf'
    bad, n = dpp_lint('\n'.join(head + [dpp]))
    assert n == 1 and len(bad) == 1
    bad, n = dpp_lint('\n'.join(head + ['\tv_add_f32_e32 v9, v9, v9', dpp]))
    assert len(bad) == 1                                   # one instruction between: still too close
    for filler in (['\ts_nop 1'], ['\tv_add_f32_e32 v9, v9, v9', '\ts_waitcnt lgkmcnt(0)'], ['\ts_nop 0', '\ts_nop 0']):
        bad, n = dpp_lint('\n'.join(head + filler + [dpp]))
        assert not bad, filler


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_no_vector_write_within_two_wait_states_of_a_dpp_read_in_the_solver(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    out = tmp_path / 'hotpants.s'
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only',
                           '-I', os.path.join(ROOT, 'include'), '-S', HOT, '-o', str(out)], stderr=subprocess.DEVNULL)
    bad, ndpp = dpp_lint(out.read_text())
    # the diagonal factor (144 per inlined copy) and the panel chains of every form: thousands of them
    assert ndpp > 2000, ndpp
    assert not bad, bad[:5]
