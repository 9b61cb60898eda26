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

SOURCES = ['ctx.hip', 'wcs_host.hip', 'resample.hip', 'combine.hip',
           'background.hip', 'api_coadd.hip', 'hotpants.hip', 'api_subtract.hip', 'elementwise.hip', 'photometry.hip', 'fitsio.hip', 'detect.hip', 'comm.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-Wno-unused-result']
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
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', str(s), '-o', str(o)]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
        elif verbose:
            print(f'up to date: {o.name} (newer than {src} and the headers)', flush=True)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
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
