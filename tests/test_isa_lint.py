"""Static check of the hand-counted LDS pipelines of the fused coadd kernels (CPU: needs hipcc, no GPU).

k_coadd_fused_own / _dma read their filter windows with inline-asm ``ds_read`` sequences and wait for them with COUNTED ``s_waitcnt lgkmcnt(N)``: row r + 1 is
in flight while row r is applied.  lgkmcnt also counts scalar loads and any LDS operation the compiler emits, and
scalar loads return out of order: ONE such instruction between two counted waits lets a wait pass early and the
pixel work on stale registers - wrong values on some waves of some launches, no fault.  Round 5 met exactly that:
a C++ ``if`` on a kernel argument inside the loop made the compiler sink the argument's ``s_load`` there; the
k_resample comparison caught it (nondeterministic differences in the last pixel of every group).

The sources bracket each counted region with ``; ZM_LGKM_BEGIN`` / ``; ZM_LGKM_END`` comments; between them only the
kernels' own asm statements (``;;#ASMSTART`` ... ``;;#ASMEND``) may touch the lgkm counter."""
import importlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'zuds-pipeline_amd', 'csrc')
checks = importlib.import_module('zuds-pipeline_amd.isa_checks')
build = importlib.import_module('zuds-pipeline_amd.build')
lint, dpp_lint = checks.lgkm_lint, checks.dpp_lint
HAVE_HIPCC = shutil.which('hipcc') is not None or os.path.exists('/opt/rocm/bin/hipcc')


def built_asm(src, tmp_path):
    """The device assembly of a translation unit: the file the build kept next to the object when it is at least as
    new as the source (the instructions that are in libzudsmi.so), else a compile with the build's own flags."""
    objdir = build.LIBDIR / 'obj'
    asm = build.device_asm(objdir, src)
    path = os.path.join(CSRC, src)
    if asm.exists() and asm.stat().st_mtime >= os.path.getmtime(path):
        return asm.read_text()
    out = tmp_path / (src + '.s')
    subprocess.check_call([build._hipcc()] + build.FLAGS + ['--cuda-device-only', '-I', os.path.join(ROOT, 'include'),
                                                            '-S', path, '-o', str(out)], stderr=subprocess.DEVNULL)
    return out.read_text()


LGKM_SOURCES = ['fused_dma.hip', 'fused_own.hip']
MIN_LGKM_REGIONS = 2 * 9
DPP_SOURCES = ['hotpants.hip']


def test_lint_sees_a_planted_scalar_load():
    txt = '\n'.join(['_Z1kv:', '\t; ZM_LGKM_BEGIN', '\t;;#ASMSTART', '\tds_read_b64 v[0:1], v2', '\t;;#ASMEND',
                     '\ts_load_dword s4, s[0:1], 0x10', '\tv_fma_f32 v0, v1, v2, v3', '\t; ZM_LGKM_END'])
    bad, regions = lint(txt)
    assert regions == 1 and [b[2].split()[0] for b in bad] == ['s_load_dword']


@pytest.mark.skipif(not HAVE_HIPCC, reason='needs hipcc')
def test_no_compiler_made_lgkm_operation_inside_the_counted_pipelines(tmp_path):
    regions, bad = 0, []
    for src in LGKM_SOURCES:
        b, r = lint(built_asm(src, tmp_path))
        regions += r
        bad += b
    # every instance of the fused kernels carries one region
    assert regions >= MIN_LGKM_REGIONS, regions
    assert not bad, bad[:5]


# ---- the DPP read hazard of the Cholesky kernels (round 5) ---------------------------------------------------------
# csrc/chol_diag.h and the panel chains of csrc/hotpants.hip broadcast fp64 values with v_mov_b64_dpp / v_fmac_f64_dpp
# (row_newbcast) written as inline asm.  gfx950 does not interlock a DPP operand against a vector instruction that wrote
# it less than two wait states earlier: the broadcast would read the register's OLD value - no fault, a wrong factor
# on some lanes.  The asm statements keep their own distance; what they cannot see is an instruction the compiler
# puts between them (a register copy, a re-materialised constant).  This lint walks the assembly: for every DPP
# instruction, no vector instruction among the preceding two wait states may write its DPP source.
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


def test_dpp_lint_restarts_its_window_at_a_label():
    """a writer reached through a back-edge or a branch target is not in the text above the label."""
    dpp = '\tv_fmac_f64_dpp v[0:1], v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf'
    far = ['_Z1kv:', '\tv_mul_f64 v[2:3], v[6:7], v[8:9]', '\ts_nop 1']
    assert not dpp_lint('\n'.join(far + [dpp]))[0]
    bad, _ = dpp_lint('\n'.join(far + ['.LBB0_3:', dpp]))                 # straight behind a join: unknown
    assert len(bad) == 1 and 'label' in bad[0][3]
    bad, _ = dpp_lint('\n'.join(far + ['.LBB0_3:', '\tv_add_f32_e32 v9, v9, v9', dpp]))
    assert len(bad) == 1
    assert not dpp_lint('\n'.join(far + ['.LBB0_3:', '\ts_nop 1', dpp]))[0]
    assert not dpp_lint('\n'.join(far + ['.LBB0_3:', '\tv_add_f32_e32 v9, v9, v9', '\tv_add_f32_e32 v9, v9, v9', dpp]))[0]


@pytest.mark.skipif(not HAVE_HIPCC, reason='needs hipcc')
def test_no_vector_write_within_two_wait_states_of_a_dpp_read_in_the_solver(tmp_path):
    ndpp, bad = 0, []
    for src in DPP_SOURCES:
        b, n = dpp_lint(built_asm(src, tmp_path))
        ndpp += n
        bad += b
    # the diagonal factor (144 per inlined copy) and the panel chains of every form: thousands of them
    assert ndpp > 2000, ndpp
    assert not bad, bad[:5]


def test_the_build_runs_both_checks_on_what_it_compiled(tmp_path):
    """build.lint_built: a finding fails the build and removes the object (CPU: no compiler needed)."""
    objdir = tmp_path
    asm = build.device_asm(objdir, 'x.hip')
    (objdir / 'x.o').write_bytes(b'')
    asm.write_text('\n'.join(['_Z1kv:', '\tv_mul_f64 v[2:3], v[6:7], v[8:9]',
                               '\tv_fmac_f64_dpp v[0:1], v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf']))
    with pytest.raises(RuntimeError, match='isa_checks'):
        build.lint_built(objdir, ['x.hip'], verbose=False)
    assert not (objdir / 'x.o').exists()
    asm.write_text('\n'.join(['_Z1kv:', '\tv_mul_f64 v[2:3], v[6:7], v[8:9]', '\ts_nop 1',
                               '\tv_fmac_f64_dpp v[0:1], v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf']))
    build.lint_built(objdir, ['x.hip'], verbose=False)
