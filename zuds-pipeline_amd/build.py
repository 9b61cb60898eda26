"""Build libzudsmi.so (HIP kernels + C-ABI) for gfx950, in-tree.

``hipcc`` cross-compiles without a GPU; the shared object lands in
``zuds-pipeline_amd/lib/`` so it travels with the source snapshot.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / 'csrc'
LIBDIR = HERE / 'lib'
LIBNAME = 'libzudsmi.so'

SOURCES = ['ctx.hip', 'wcs_host.hip', 'resample.hip', 'resample_opts.hip', 'maskbox.hip', 'fused_host.hip', 'fused_dma.hip', 'fused_own.hip', 'combine.hip',
           'background.hip', 'api_coadd.hip', 'hotpants.hip', 'hp_vectors.hip', 'hp_apply.hip', 'api_subtract.hip', 'select_bracket.hip', 'elementwise.hip', 'photometry.hip', 'fitsio.hip', 'detect.hip', 'comm.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-Wno-unused-result']
# the device assembly of every translation unit stays next to its object (obj/<stem>-hip-amdgcn-amd-amdhsa-gfx950.s):
# it IS what was assembled into the object, and isa_checks lints it before the library is linked
TEMPS = ['-save-temps=obj']
EXTRA_FLAGS = {}     # per-source additions, e.g. {'x.hip': ['-mllvm', '...']}
# developer: ZM_HIPCC_FLAGS='-DFF_TALL=1' adds flags to every translation unit (use with --force)
FLAGS += os.environ.get('ZM_HIPCC_FLAGS', '').split()


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'),
                 '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found; a ROCm toolchain is required')


def _stale(target, deps):
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def device_asm(objdir, src):
    return objdir / (Path(src).stem + '-hip-amdgcn-amd-amdhsa-gfx950.s')


def lint_built(objdir, sources, verbose=True):
    """isa_checks on the device assembly of the translation units that were just compiled - the instructions that
    went into the objects, under the flags of this build (ZM_HIPCC_FLAGS included).  A finding fails the build: the
    object is removed so that the next build compiles and checks it again."""
    from . import isa_checks
    findings = []
    for src in sources:
        asm = device_asm(objdir, src)
        stem = Path(src).stem
        for junk in objdir.glob(stem + '-*'):             # preprocessed sources, bitcode, host assembly: tens of MB
            if junk != asm and junk.suffix in ('.hipi', '.bc', '.out', '.txt', '.hipfb') or junk.name.endswith('linux-gnu.s'):
                junk.unlink()
        for junk in objdir.glob(stem + '.hip-*'):
            junk.unlink()
        if not asm.exists():
            raise RuntimeError(f'{src}: no device assembly at {asm} (was -save-temps=obj dropped from the flags?)')
        out, counts = isa_checks.check_file(str(asm))
        if verbose:
            print(f'isa_checks {asm.name}: {counts["lgkm_regions"]} counted LDS pipelines, {counts["dpp"]} DPP reads, '
                  f'{len(out)} finding(s)', flush=True)
        if out:
            (objdir / (stem + '.o')).unlink(missing_ok=True)
            findings += out
    if findings:
        raise RuntimeError('isa_checks: ' + '; '.join(findings[:5]) + (f' ... ({len(findings)} in all)' if len(findings) > 5 else ''))


def build(force=False, verbose=True):
    """Compile every HIP translation unit and link libzudsmi.so."""
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / 'obj'
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    headers = list(CSRC.glob('*.h')) + [HERE.parent / 'include' / 'zudsmi.h']
    objs = []
    procs = []
    for src in SOURCES:
        s = CSRC / src
        o = objdir / (s.stem + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + TEMPS + EXTRA_FLAGS.get(src, []) + ['-c', str(s), '-o', str(o)]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
        elif verbose:
            print(f'up to date: {o.name} (newer than {src} and the headers)', flush=True)
    rc = {src: p.wait() for src, p in procs}
    failed = [src for src in rc if rc[src] != 0]
    # the units that did compile are checked (and their intermediate files removed) whether or not another one failed:
    # their objects are up to date from now on and would never be looked at again
    lint_built(objdir, [src for src in rc if rc[src] == 0], verbose)
    if failed:
        for src in failed:
            for junk in list(objdir.glob(Path(src).stem + '-*')) + list(objdir.glob(Path(src).stem + '.hip-*')):
                junk.unlink()
        raise RuntimeError(f'hipcc failed on {", ".join(failed)}')
    lib = LIBDIR / LIBNAME
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o',
               str(lib)] + [str(o) for o in objs] + ['-ldl']          # (dlopen of librccl: comm.hip)
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if verbose:
        print(f'libzudsmi: compiled {len(procs)} of {len(SOURCES)} translation units'
              f'{" (forced)" if force else ""}, '
              f'{"linked" if (force or procs) else "library reused"}', flush=True)
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv)
