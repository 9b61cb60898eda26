"""ctypes front end of the C restatement (oracle; TEST INFRASTRUCTURE).

``load()`` builds ``_build/libzmoracle.so`` with gcc when it is missing (or stale) and
returns the wrapper functions; ``native=True`` rebuilds with ``-march=native`` into a
private directory for timing on the machine at hand (bench.py's cpu_baseline leg).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'zm_oracle.c')
SOURCES = [SRC, os.path.join(HERE, 'zm_hotpants.c')]      # resample -> coadd leg; subtraction leg (round 6)
NPV = 40


class zo_wcs(C.Structure):
    _fields_ = [('crpix', C.c_double * 2), ('crval', C.c_double * 2), ('cd', C.c_double * 4),
                ('pv1', C.c_double * NPV), ('pv2', C.c_double * NPV),
                ('naxis', C.c_int32 * 2), ('has_pv', C.c_int32), ('pad_', C.c_int32)]


def build(native=False):
    if native:
        out = os.path.join(tempfile.gettempdir(), f'libzmoracle_native_{os.getuid()}.so')
        flags = ['-O3', '-march=native', '-fopenmp', '-fPIC']
    else:
        os.makedirs(os.path.join(HERE, '_build'), exist_ok=True)
        out = os.path.join(HERE, '_build', 'libzmoracle.so')
        flags = ['-O3', '-fopenmp', '-fPIC']
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in SOURCES):
        subprocess.check_call([os.environ.get('CC', 'gcc')] + flags + ['-shared'] + SOURCES + ['-o', out, '-lm'])
    return out


def _wcs(w):
    s = zo_wcs()
    s.crpix[:] = list(w.crpix)
    s.crval[:] = list(w.crval)
    s.cd[:] = list(np.asarray(w.cd, dtype=np.float64).ravel())
    s.pv1[:] = list(w.pv1)
    s.pv2[:] = list(w.pv2)
    s.naxis[:] = [int(w.naxis[0]), int(w.naxis[1])]
    s.has_pv = int(bool(w.has_pv))
    return s


class zo_hp_params(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('tu', 'tl', 'iu', 'il', 'r', 'rss', 'fin', 'fi', 'ft', 'ks')] + \
               [(k, C.c_int32) for k in ('nsx', 'nsy', 'nrx', 'nry', 'ko', 'bgo', 'nss', 'normalize', 'ngauss')] + \
               [('deg', C.c_int32 * 4), ('sigma', C.c_double * 4)]


class zo_hp_region(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('solved', 'nstamps_total', 'nstamps_used', 'niter', 'ncoeff', 'pad_')] + \
               [('kernel_sum', C.c_double), ('chi2', C.c_double)]


class CPort(object):
    KIND = {'WEIGHTED': 0, 'MEDIAN': 1, 'CLIPPED': 2, 'AVERAGE': 3}

    def __init__(self, path):
        self.L = C.CDLL(path)
        self.L.zo_threads.restype = C.c_int
        P = C.c_void_p
        self.L.zo_positions.argtypes = [C.POINTER(zo_wcs), C.POINTER(zo_wcs), C.c_int, C.c_int, P, P]
        self.L.zo_resample.argtypes = [P, P, P, C.c_int, C.c_int, P, P, C.c_int, C.c_int, C.c_int,
                                       C.c_double, P, P, P]
        self.L.zo_combine.argtypes = [P, P, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_double, P, P]
        self.L.zo_background.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P]

    def threads(self):
        return int(self.L.zo_threads())

    def set_threads(self, n):
        self.L.zo_set_threads(int(n))

    def positions(self, wout, win, onx, ony):
        px = np.empty((ony, onx))
        py = np.empty((ony, onx))
        a, b = _wcs(wout), _wcs(win)
        self.L.zo_positions(C.byref(a), C.byref(b), onx, ony, px.ctypes.data, py.ctypes.data)
        return px, py

    def resample(self, img, wgt, px, py, kind=3, fscale=1.0, mask=None):
        img = np.ascontiguousarray(img, dtype=np.float32)
        wgt = None if wgt is None else np.ascontiguousarray(wgt, dtype=np.float32)
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        px = np.ascontiguousarray(px, dtype=np.float64)
        py = np.ascontiguousarray(py, dtype=np.float64)
        ny, nx = img.shape
        ony, onx = px.shape
        out = np.empty((ony, onx))
        outw = np.empty((ony, onx))
        outm = None if mask is None else np.empty((ony, onx), dtype=np.int64)
        self.L.zo_resample(img.ctypes.data, None if wgt is None else wgt.ctypes.data,
                           None if mask is None else mask.ctypes.data, nx, ny, px.ctypes.data,
                           py.ctypes.data, onx, ony, int(kind), float(fscale), out.ctypes.data,
                           outw.ctypes.data, None if outm is None else outm.ctypes.data)
        return out, outw, outm

    def background(self, img, wgt=None, mesh=128, fsize=3, want_images=True):
        """oracle.background.background: (bkg, rms, backmean, backsig, nodes_b, nodes_s)."""
        img = np.ascontiguousarray(img, dtype=np.float64)
        wgt = None if wgt is None else np.ascontiguousarray(wgt, dtype=np.float32)
        ny, nx = img.shape
        nbx, nby = (nx - 1) // mesh + 1, (ny - 1) // mesh + 1
        bkg = np.empty((ny, nx)) if want_images else None
        rms = np.empty((ny, nx)) if want_images else None
        stats = np.empty(2)
        nb, ns = np.empty((nby, nbx)), np.empty((nby, nbx))
        self.L.zo_background(img.ctypes.data, None if wgt is None else wgt.ctypes.data, nx, ny, int(mesh), int(fsize),
                             None if bkg is None else bkg.ctypes.data, None if rms is None else rms.ctypes.data,
                             stats.ctypes.data, nb.ctypes.data, ns.ctypes.data)
        return bkg, rms, float(stats[0]), float(stats[1]), nb, ns

    def hotpants(self, sci, ref, sci_rms, ref_rms, bpm=None, only_region=-1, **kw):
        """oracle.hotpants.subtract in C (zm_hotpants.c): (diff, noise, info) with info = dict(regions=[dict or
        None per region], nmasked).  ``only_region``: fit and apply that region alone."""
        from oracle import hotpants as ohp
        p = ohp.params(**kw)
        if len(p['deg']) > 4:
            raise ValueError('at most four Gaussians')
        P = zo_hp_params()
        for k in ('tu', 'tl', 'iu', 'il', 'r', 'rss', 'fin', 'fi', 'ft', 'ks'):
            setattr(P, k, float(p[k]))
        for k in ('nsx', 'nsy', 'nrx', 'nry', 'ko', 'bgo', 'nss', 'normalize'):
            setattr(P, k, int(p[k]))
        P.ngauss = len(p['deg'])
        for i, (d, sg) in enumerate(zip(p['deg'], p['sigma'])):
            P.deg[i], P.sigma[i] = int(d), float(sg)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (sci, ref, sci_rms, ref_rms)]
        ny, nx = a[0].shape
        b = None if bpm is None else np.ascontiguousarray(np.asarray(bpm) != 0, dtype=np.uint8)
        diff, noise = np.empty((ny, nx)), np.empty((ny, nx))
        nreg = P.nrx * P.nry
        regs = (zo_hp_region * nreg)()
        nm = C.c_int64()
        self.L.zo_hotpants.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.POINTER(zo_hp_params), C.c_int, C.c_void_p,
                                                        C.c_void_p, C.POINTER(zo_hp_region), C.POINTER(C.c_int64)]
        rc = self.L.zo_hotpants(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                None if b is None else b.ctypes.data, nx, ny, C.byref(P), int(only_region),
                                diff.ctypes.data, noise.ctypes.data, regs, C.byref(nm))
        if rc:
            raise RuntimeError(f'zo_hotpants failed ({rc})')
        infos = [dict(nstamps_total=r.nstamps_total, nstamps_used=r.nstamps_used, niter=r.niter, ncoeff=r.ncoeff,
                      kernel_sum=r.kernel_sum, chi2=r.chi2) if r.solved else None for r in regs]
        return diff, noise, dict(regions=infos, nmasked=int(nm.value))

    def combine(self, vals, wgts, kind='CLIPPED', clip_sigma=4.0, clip_ampfrac=0.3):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        wgts = np.ascontiguousarray(wgts, dtype=np.float64)
        n = vals.shape[0]
        npix = int(np.prod(vals.shape[1:]))
        out = np.empty(vals.shape[1:])
        outw = np.empty(vals.shape[1:])
        self.L.zo_combine(vals.ctypes.data, wgts.ctypes.data, n, npix, self.KIND[kind.upper()],
                          float(clip_sigma), float(clip_ampfrac), out.ctypes.data, outw.ctypes.data)
        return out, outw


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one (a GPU box
    hands a share of a big host to each GPU), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def load(native=False):
    return CPort(build(native))
