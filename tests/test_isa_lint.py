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
