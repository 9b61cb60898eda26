"""Seeing estimate without catalogs or network.

The reference's ``estimate_seeing`` (``zuds/seeing.py:10-118``) takes the median SExtractor
``FWHM_IMAGE`` of catalog sources matched to Gaia stars (Kowalski / astroquery) and writes it to
the header as ``SEEING``.  Here the stars are isolated, unsaturated, unmasked local maxima of
the background-subtracted image and their FWHM comes from adaptive Gaussian-weighted second
moments, both computed by libzudsmi (``zm_find_stars``, ``zm_star_fwhm``); the header card,
its comment and the ``save()`` are the reference's.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr
from .engine import get_engine

__all__ = ['estimate_seeing', 'measure_seeing', 'measure_seeing_dev']

NMAX = 300            # brightest stars used
ISOLATION = 5         # pixels: a star is the maximum of its 11 x 11 box
BORDER = 12
HALF = 10             # moments window: 21 x 21
NSIGMA = 30.0         # detection threshold above the background, in background sigmas


def measure_seeing(data, bad=None, saturate=None, engine=None):
    """(seeing in pixels, number of stars used) of a 2-D image; ``bad``: boolean bad-pixel
    map.  Raises RuntimeError when no usable star is found (as the reference does when no
    calibrator matches, ``zuds/seeing.py:103-105``)."""
    eng = engine or get_engine()
    data = np.ascontiguousarray(data, dtype=np.float32)
    ny, nx = data.shape
    wgt = None if bad is None else np.where(bad, 0.0, 1.0).astype(np.float32)
    _, _, sub, (bmean, bsig) = eng.background(data, wgt, want=('sub',))
    lo = float(NSIGMA * max(bsig, 1e-6))
    hi = float(0.5 * saturate - bmean) if saturate else 3.0e38
    bad8 = None if bad is None else np.ascontiguousarray(bad).astype(np.uint8)
    cap = 65536
    xs = np.empty(cap, np.int32)
    ys = np.empty(cap, np.int32)
    pk = np.empty(cap, np.float32)
    n = C.c_int(0)
    while True:
        check(eng.L.zm_find_stars(eng.ctx, ptr(sub), ptr(bad8), nx, ny, lo, hi, ISOLATION, BORDER, cap,
                                  ptr(xs), ptr(ys), ptr(pk), C.byref(n)), 'zm_find_stars')
        if n.value <= cap:
            break
        lo *= 2.0                       # crowded field: keep only the brighter half
    m = n.value
    if m == 0:
        raise RuntimeError('Unable to find any stars to estimate the seeing')
    order = np.lexsort((xs[:m], ys[:m], -pk[:m].astype(np.float64)))[:NMAX]
    sx = np.ascontiguousarray(xs[:m][order])
    sy = np.ascontiguousarray(ys[:m][order])
    k = len(sx)
    fw = np.empty(k)
    cx = np.empty(k)
    cy = np.empty(k)
    check(eng.L.zm_star_fwhm(eng.ctx, ptr(sub), nx, ny, k, ptr(sx), ptr(sy), HALF, ptr(fw), ptr(cx),
                             ptr(cy)), 'zm_star_fwhm')
    good = np.isfinite(fw)
    if not good.any():
        raise RuntimeError('Unable to measure the width of any star to estimate the seeing')
    return float(np.nanmedian(fw)), int(good.sum())


def measure_seeing_dev(img, bad=None, saturate=None, engine=None):
    """``measure_seeing`` on planes that are already in HBM: ``img`` a float32 torch tensor on the
    engine's device, ``bad`` a uint8 tensor (non-zero = unusable) or None.  The same launches with
    device pointers (``zm_background_dev``, ``zm_find_stars_dev``, ``zm_star_fwhm_dev``): the same
    stars, the same number.  The caller has the engine's stream current."""
    import torch
    eng = engine or get_engine()
    ny, nx = img.shape
    wgt = None if bad is None else (bad == 0).to(torch.float32)
    sub = torch.empty_like(img)
    stats = (C.c_double * 2)()
    check(eng.L.zm_background_dev(eng.ctx, img.data_ptr(), wgt.data_ptr() if wgt is not None else None, nx, ny,
                                  128, 3, None, None, sub.data_ptr(), stats), 'zm_background_dev')
    bmean, bsig = stats[0], stats[1]
    lo = float(NSIGMA * max(bsig, 1e-6))
    hi = float(0.5 * saturate - bmean) if saturate else 3.0e38
    cap = 65536
    xs = np.empty(cap, np.int32)
    ys = np.empty(cap, np.int32)
    pk = np.empty(cap, np.float32)
    n = C.c_int(0)
    while True:
        check(eng.L.zm_find_stars_dev(eng.ctx, sub.data_ptr(), bad.data_ptr() if bad is not None else None, nx, ny,
                                      lo, hi, ISOLATION, BORDER, cap, ptr(xs), ptr(ys), ptr(pk), C.byref(n)),
              'zm_find_stars_dev')
        if n.value <= cap:
            break
        lo *= 2.0
    m = n.value
    if m == 0:
        raise RuntimeError('Unable to find any stars to estimate the seeing')
    order = np.lexsort((xs[:m], ys[:m], -pk[:m].astype(np.float64)))[:NMAX]
    sx = np.ascontiguousarray(xs[:m][order])
    sy = np.ascontiguousarray(ys[:m][order])
    k = len(sx)
    fw = np.empty(k)
    cx = np.empty(k)
    cy = np.empty(k)
    check(eng.L.zm_star_fwhm_dev(eng.ctx, sub.data_ptr(), nx, ny, k, ptr(sx), ptr(sy), HALF, ptr(fw), ptr(cx),
                                 ptr(cy)), 'zm_star_fwhm_dev')
    good = np.isfinite(fw)
    if not good.any():
        raise RuntimeError('Unable to measure the width of any star to estimate the seeing')
    return float(np.nanmedian(fw)), int(good.sum())


def estimate_seeing(image):
    """Measure the seeing of ``image`` and record it as the reference does
    (``zuds/seeing.py:113-118``): header ``SEEING`` (pixels), its comment, ``save()``."""
    bad = None
    mask = getattr(image, '_mask_image', None) or getattr(image, 'mask_image', None)
    if mask is not None:
        bad = mask.boolean.data.astype(bool)
    sat = image.header.get('SATURATE')
    seeing, _ = measure_seeing(image.data, bad, float(sat) if sat else None)
    image.header['SEEING'] = float(seeing)
    image.header_comments['SEEING'] = 'FWHM of seeing in pixels (Goldstein)'
    if image.ismapped:          # a transaction copy held in memory has no file to update
        image.save()
    return seeing
