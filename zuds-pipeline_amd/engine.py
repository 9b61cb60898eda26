"""numpy-facing wrapper of the libzudsmi context.

One ``Engine`` per process/GPU.  Every method goes through the C-ABI
(``include/zudsmi.h``); inputs are borrowed numpy arrays, outputs are fresh
numpy arrays.  This is the edge that replaces ``subprocess.check_call`` on
SWarp / SExtractor / hotpants in the reference (``zuds/coadd.py:133,156``,
``zuds/swarp.py:175``, ``zuds/sextractor.py:128``, ``zuds/subtraction.py:162``).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (COMBINE, MASKCOMB, RESAMPLE, as_f32, as_i32, as_mask, check, ptr,
                   wcs_struct)

_default = None


def get_engine(device=0):
    """Process-wide engine (created on first use)."""
    global _default
    if _default is None:
        _default = Engine(device)
    return _default


def _enum(table, v, what):
    if isinstance(v, int):
        return v
    try:
        return table[str(v).upper()]
    except KeyError:
        raise ValueError(f'unknown {what} "{v}"; expected one of {list(table)}')


def coadd_params(combine='CLIPPED', mask_combine='AND', resample='LANCZOS3',
                 subtract_back=True, back_size=128, back_filtersize=3,
                 rescale_weights=True, clip_sigma=4.0, clip_ampfrac=0.3,
                 weight_thresh=1e-30):
    """zm_coadd_params with the values of default.swarp / mask.swarp."""
    p = _lib.zm_coadd_params()
    _lib.lib().zm_coadd_params_default(C.byref(p))
    p.combine = _enum(COMBINE, combine, 'COMBINE_TYPE')
    p.mask_combine = _enum(MASKCOMB, mask_combine, 'mask COMBINE_TYPE')
    p.resample = _enum(RESAMPLE, resample, 'RESAMPLING_TYPE')
    p.subtract_back = int(bool(subtract_back))
    p.back_size = int(back_size)
    p.back_filtersize = int(back_filtersize)
    p.rescale_weights = int(bool(rescale_weights))
    p.clip_sigma = float(clip_sigma)
    p.clip_ampfrac = float(clip_ampfrac)
    p.weight_thresh = float(weight_thresh)
    return p


class Engine(object):

    def __init__(self, device=0, stream=None):
        """``stream``: a HIP stream handle (``torch.cuda.Stream(...).cuda_stream``) the context works on from the start -
        it then never creates a stream of its own (``zm_ctx_create_on_stream``): what chains that run side by side on
        one GPU want."""
        self.L = _lib.lib()
        self._ctx = C.c_void_p()
        if stream:
            check(self.L.zm_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(self._ctx)),
                  'zm_ctx_create_on_stream')
        else:
            check(self.L.zm_ctx_create(int(device), C.byref(self._ctx)),
                  'zm_ctx_create')
        self.device = int(device)

    def close(self):
        if self._ctx:
            self.L.zm_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        return self._ctx

    def synchronize(self):
        check(self.L.zm_ctx_synchronize(self._ctx))

    def set_stream(self, stream_handle):
        check(self.L.zm_ctx_set_stream(self._ctx, C.c_void_p(stream_handle)))

    def set_share(self, nctx):
        """Declare that ``nctx`` engines subtract on this GPU at the same time (one host thread
        each): every engine then sizes its kernel-fit launches to 1 / nctx of the GPU."""
        check(self.L.zm_ctx_set_share(self._ctx, int(nctx)), 'zm_ctx_set_share')

    def set_conventions(self, edge='zero', mask_resample='or'):
        """SWarp's own edge / mask conventions for every resample and coadd of this engine
        (``zm_ctx_set_conventions``): ``edge`` 'zero' (default) or 'truncate' (kernel truncated at the frame edge),
        ``mask_resample`` 'or' (default) or 'lanczos_round' (masks interpolated and rounded).  INTEGRATION.md."""
        e = {'zero': 0, 'truncate': 1}
        m = {'or': 0, 'lanczos_round': 1}
        if str(edge).lower() not in e or str(mask_resample).lower() not in m:
            raise ValueError(f'edge must be one of {list(e)}, mask_resample one of {list(m)}')
        check(self.L.zm_ctx_set_conventions(self._ctx, e[str(edge).lower()], m[str(mask_resample).lower()]),
              'zm_ctx_set_conventions')

    def query(self, what):
        """``zm_ctx_query``: 'fused_form' (0 none, 1 k_coadd_fused_dma, 2 k_coadd_fused_own), 'dev_build'."""
        v = C.c_int64()
        check(self.L.zm_ctx_query(self._ctx, what.encode(), C.byref(v)), 'zm_ctx_query')
        return v.value

    # -- timing ----------------------------------------------------------------
    def timing(self, on=True, only=None):
        """HIP-event timers around the launches; ``only``: time just that scope."""
        check(self.L.zm_timing_filter(self._ctx, only.encode() if only else None))
        check(self.L.zm_timing_enable(self._ctx, int(on)))

    def timing_reset(self):
        check(self.L.zm_timing_reset(self._ctx))

    def timing_read(self, name):
        ms = C.c_double()
        n = C.c_int64()
        check(self.L.zm_timing_read(self._ctx, name.encode(), C.byref(ms),
                                    C.byref(n)))
        return ms.value, n.value

    # -- resample ---------------------------------------------------------------
    def resample(self, img, win, wout, wgt=None, mask=None, kernel='LANCZOS3',
                 fscale=1.0):
        """Resample ``img`` (+weight, +mask) from WCS ``win`` onto ``wout``.

        Returns (out_img, out_wgt, out_mask); entries are None when the
        corresponding input was None."""
        sin, sout = wcs_struct(win), wcs_struct(wout)
        ny, nx = (img if img is not None else mask).shape
        if (sin.naxis[0], sin.naxis[1]) != (nx, ny):
            raise ValueError(f'input WCS NAXIS {tuple(sin.naxis)} does not match '
                             f'data shape {(ny, nx)}')
        onx, ony = sout.naxis[0], sout.naxis[1]
        img = as_f32(img)
        wgt = as_f32(wgt)
        mask, mtype = as_mask(mask)
        oimg = owgt = omask = None
        if img is not None:
            oimg = np.empty((ony, onx), dtype=np.float32)
            owgt = np.empty((ony, onx), dtype=np.float32)
        if mask is not None:
            omask = np.empty((ony, onx), dtype=np.int32)
        fn = self.L.zm_resample_i16 if mtype == _lib.MASKTYPE_I16 else self.L.zm_resample
        check(fn(self._ctx, ptr(img), ptr(wgt), ptr(mask), C.byref(sin), C.byref(sout),
                 _enum(RESAMPLE, kernel, 'RESAMPLING_TYPE'), float(fscale), ptr(oimg), ptr(owgt),
                 ptr(omask)), 'zm_resample')
        return oimg, owgt, omask

    # -- coadd -------------------------------------------------------------------
    def coadd(self, frames, wout, params=None, want_mask=True):
        """frames: list of dicts {img, wgt (or None), mask (or None), wcs,
        flxscale}.  Returns (img, wgt, mask, mask_wgt)."""
        if params is None:
            params = coadd_params()
        n = len(frames)
        if n == 0:
            raise ValueError('coadd needs at least one frame')
        arr = (_lib.zm_frame * n)()
        keep = []
        any_mask = False
        for i, f in enumerate(frames):
            img = as_f32(f['img'])
            wgt = as_f32(f.get('wgt'))
            msk, mtype = as_mask(f.get('mask'))      # an int16 plane goes over as it is
            keep += [img, wgt, msk]
            s = wcs_struct(f['wcs'])
            if (s.naxis[1], s.naxis[0]) != img.shape:
                raise ValueError(f'frame {i}: WCS NAXIS {tuple(s.naxis)} does not '
                                 f'match data shape {img.shape}')
            arr[i].img = ptr(img)
            arr[i].wgt = ptr(wgt)
            arr[i].mask = ptr(msk)
            arr[i].mask_type = mtype
            arr[i].wcs = s
            arr[i].flxscale = float(f.get('flxscale', 1.0))
            any_mask |= msk is not None
        sout = wcs_struct(wout)
        onx, ony = sout.naxis[0], sout.naxis[1]
        oimg = np.empty((ony, onx), dtype=np.float32)
        owgt = np.empty((ony, onx), dtype=np.float32)
        omask = omw = None
        if want_mask and any_mask:
            omask = np.empty((ony, onx), dtype=np.int32)
            omw = np.empty((ony, onx), dtype=np.float32)
        check(self.L.zm_coadd(self._ctx, n, arr, C.byref(sout), C.byref(params),
                              ptr(oimg), ptr(owgt), ptr(omask), ptr(omw)),
              'zm_coadd')
        return oimg, owgt, omask, omw

    def resample_mask(self, mask, win, wout, kernel='LANCZOS3', combine='OR'):
        """Resample one integer mask; returns (None, None, mask, coverage) where
        coverage == 0 marks pixels the input does not reach (bit 16)."""
        mask, _ = as_mask(mask)
        p = coadd_params(combine='WEIGHTED', mask_combine=combine, resample=kernel,
                         subtract_back=False, rescale_weights=False)
        frame = dict(img=np.zeros(mask.shape, dtype=np.float32), wgt=None, mask=mask,
                     wcs=win, flxscale=1.0)
        _, _, omask, omw = self.coadd([frame], wout, p, want_mask=True)
        return None, None, omask, omw

    def autogrid(self, wcss):
        n = len(wcss)
        arr = (_lib.zm_wcs * n)(*[wcs_struct(w) for w in wcss])
        out = _lib.zm_wcs()
        check(self.L.zm_autogrid(n, arr, C.byref(out)), 'zm_autogrid')
        from .wcs import WCS
        return WCS.from_struct(out)

    def flux_scale(self, win, wout, flxscale=1.0):
        v = C.c_double()
        a, b = wcs_struct(win), wcs_struct(wout)
        check(self.L.zm_flux_scale(C.byref(a), C.byref(b), float(flxscale),
                                   C.byref(v)))
        return v.value

    # -- combine a host stack (tests, row-band exchange) ------------------------------
    def combine_stack(self, vals, wgts, params=None):
        """Combine host arrays vals, wgts of shape (n, ny, nx) on the device."""
        import ctypes as C
        from . import hipmem
        if params is None:
            params = coadd_params()
        vals = as_f32(vals)
        wgts = as_f32(wgts)
        n = vals.shape[0]
        npix = int(np.prod(vals.shape[1:]))
        pairs = np.ascontiguousarray(np.stack([vals.reshape(n, npix),
                                               wgts.reshape(n, npix)], axis=-1))
        d_stack = hipmem.DeviceBuffer(pairs.nbytes)
        d_img = hipmem.DeviceBuffer(npix * 4)
        d_wgt = hipmem.DeviceBuffer(npix * 4)
        d_stack.upload(pairs)
        check(self.L.zm_combine_stack_dev(self._ctx, n, d_stack.ptr, npix, npix,
                                          C.byref(params), d_img.ptr, d_wgt.ptr),
              'zm_combine_stack_dev')
        self.synchronize()
        oimg = d_img.download(np.float32, vals.shape[1:])
        owgt = d_wgt.download(np.float32, vals.shape[1:])
        return oimg, owgt

    # -- mesh background -----------------------------------------------------------------
    def background(self, img, wgt=None, mesh=128, filtersize=3, want=('bkg', 'rms', 'sub')):
        """(bkg, rms, sub, (backmean, backsig)); entries not in ``want`` are None."""
        img = as_f32(img)
        wgt = as_f32(wgt)
        ny, nx = img.shape
        out = {k: (np.empty((ny, nx), dtype=np.float32) if k in want else None)
               for k in ('bkg', 'rms', 'sub')}
        stats = (C.c_double * 2)()
        check(self.L.zm_background(self._ctx, ptr(img), ptr(wgt), nx, ny, int(mesh),
                                   int(filtersize), ptr(out['bkg']), ptr(out['rms']),
                                   ptr(out['sub']), stats), 'zm_background')
        return out['bkg'], out['rms'], out['sub'], (stats[0], stats[1])

    # -- subtraction -------------------------------------------------------------------------
    def subtract(self, sci, sci_rms, ref, ref_rms, bpm=None, params=None, **kw):
        """Alard-Lupton subtraction of ``ref`` from ``sci`` (same grid).

        Returns (diff, noise, info dict)."""
        if params is None:
            params = hp_params(**kw)
        sci, sci_rms = as_f32(sci), as_f32(sci_rms)
        ref, ref_rms = as_f32(ref), as_f32(ref_rms)
        ny, nx = sci.shape
        for a, nm in ((sci_rms, 'sci_rms'), (ref, 'ref'), (ref_rms, 'ref_rms')):
            if a.shape != (ny, nx):
                raise ValueError(f'{nm} has shape {a.shape}, expected {(ny, nx)}')
        if bpm is not None:
            bpm = np.ascontiguousarray(bpm).astype(np.uint8, copy=False)
            if bpm.shape != (ny, nx):
                raise ValueError(f'bpm has shape {bpm.shape}, expected {(ny, nx)}')
        diff = np.empty((ny, nx), dtype=np.float32)
        noise = np.empty((ny, nx), dtype=np.float32)
        info = _lib.zm_hp_info()
        check(self.L.zm_subtract(self._ctx, ptr(sci), ptr(sci_rms), ptr(ref), ptr(ref_rms),
                                 ptr(bpm), nx, ny, C.byref(params), ptr(diff), ptr(noise),
                                 C.byref(info)), 'zm_subtract')
        return diff, noise, {k: getattr(info, k) for k, _ in info._fields_}

    def subtract_batch(self, frames, params=None, **kw):
        """Many subtractions of one configuration in one call (``zm_subtract_batch``: the kernel fits of all
        frames as one chain of launches; same products as :meth:`subtract` frame by frame).

        ``frames``: sequence of ``(sci, sci_rms, ref, ref_rms, bpm or None)``, all of one shape; ``params``: one
        ``zm_hp_params`` for all, or a sequence (data limits and fill values may differ per frame), or
        keywords as for :meth:`subtract`.  Returns a list of (diff, noise, info dict)."""
        frames = list(frames)
        n = len(frames)
        if n == 0:
            return []
        if params is None:
            params = hp_params(**kw)
        plist = list(params) if isinstance(params, (list, tuple)) else [params] * n
        if len(plist) != n:
            raise ValueError('one zm_hp_params per frame, or one for all')
        arr = (_lib.zm_sub_job * n)()
        infos = (_lib.zm_hp_info * n)()
        keep, outs = [], []
        ny, nx = as_f32(frames[0][0]).shape
        for k, (sci, sci_rms, ref, ref_rms, bpm) in enumerate(frames):
            planes = [as_f32(a) for a in (sci, sci_rms, ref, ref_rms)]
            for a, nm in zip(planes, ('sci', 'sci_rms', 'ref', 'ref_rms')):
                if a.shape != (ny, nx):
                    raise ValueError(f'frame {k}: {nm} has shape {a.shape}, expected {(ny, nx)}')
            if bpm is not None:
                bpm = np.ascontiguousarray(bpm).astype(np.uint8, copy=False)
                if bpm.shape != (ny, nx):
                    raise ValueError(f'frame {k}: bpm has shape {bpm.shape}, expected {(ny, nx)}')
            diff = np.empty((ny, nx), dtype=np.float32)
            noise = np.empty((ny, nx), dtype=np.float32)
            keep.append((planes, bpm))
            outs.append((diff, noise))
            arr[k] = _lib.zm_sub_job(ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), ptr(planes[3]), ptr(bpm),
                                     C.pointer(plist[k]), ptr(diff), ptr(noise))
        check(self.L.zm_subtract_batch(self._ctx, n, arr, nx, ny, infos), 'zm_subtract_batch')
        return [(d, nz, {f: getattr(infos[k], f) for f, _ in _lib.zm_hp_info._fields_})
                for k, (d, nz) in enumerate(outs)]

    def median_mad(self, img, mask=None):
        """(median, 1.4826 MAD) of the pixels whose mask is 0
        (quick_background_estimate, zuds/utils.py:32-53)."""
        img = as_f32(img)
        mask = as_i32(mask)
        if mask is not None and mask.shape != img.shape:
            raise ValueError('mask shape does not match image shape')
        med = C.c_double()
        mad = C.c_double()
        check(self.L.zm_median_mad(self._ctx, ptr(img), ptr(mask), img.size, C.byref(med),
                                   C.byref(mad)), 'zm_median_mad')
        return med.value, mad.value


def _engine_aperture(self, img, x, y, rms=None, mask=None, radius=3.0):
    """Forced circular apertures at 0-based pixel positions: (flux, fluxerr, flags)."""
    img, rms, mask = as_f32(img), as_f32(rms), as_i32(mask)
    ny, nx = img.shape
    x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
    y = np.ascontiguousarray(np.atleast_1d(y), dtype=np.float64)
    if x.shape != y.shape:
        raise ValueError('x and y must have the same shape')
    n = x.size
    flux = np.zeros(n)
    err = np.zeros(n)
    flags = np.zeros(n, dtype=np.int32)
    check(self.L.zm_aperture_photometry(self._ctx, ptr(img), ptr(rms), ptr(mask), nx, ny, n,
                                        ptr(x), ptr(y), float(radius), ptr(flux), ptr(err),
                                        ptr(flags)), 'zm_aperture_photometry')
    return flux, err, flags


Engine.aperture_photometry = _engine_aperture


def hp_params(**kw):
    """zm_hp_params with hotpants' defaults, overridden by keyword (tu, tl, iu, il,
    r, rss, fin, fi, nsx, nsy, nrx, nry, ko, bgo, nss, normalize, ft, ks, deg,
    sigma; limits_dev / limits_nsigma: the lower data limits taken from background
    estimates that stayed on the device, ``zm_hp_params`` in include/zudsmi.h)."""
    p = _lib.zm_hp_params()
    _lib.lib().zm_hp_params_default(C.byref(p))
    for k, v in kw.items():
        if k == 'deg':
            p.ngauss = len(v)
            for i, d in enumerate(v):
                p.deg[i] = int(d)
        elif k == 'sigma':
            for i, s in enumerate(v):
                p.sigma[i] = float(s)
        elif k in ('nsx', 'nsy', 'nrx', 'nry', 'ko', 'bgo', 'nss', 'normalize'):
            setattr(p, k, int(v))
        elif k == 'limits_dev':
            p.limits_dev = int(v) if v else None       # device address of the six doubles of zm_median_mad2_async_dev
        elif k == 'flag_mask_dev':
            p.flag_mask_dev = int(v) if v else None    # device address of the int32 mask plane that takes flag_bit
        elif k in ('flag_bit', 'async_info'):
            setattr(p, k, int(v))
        elif hasattr(p, k):
            setattr(p, k, float(v))
        else:
            raise ValueError(f'unknown hotpants parameter "{k}"')
    return p
