"""Pixel-level cuts of the reference's candidate filter.

``filter_sexcat`` (``zuds/filterobjects.py:57-195``) mixes SExtractor catalog columns with
three tests that need only pixels; those three are computed here on the GPU for any list of
positions (``X_IMAGE``, ``Y_IMAGE``: 1-based, as SExtractor reports them):

* ``BPMCUT``  exact-overlap aperture sum (r = 6 px) of the boolean bad-pixel map; must be 0;
* ``RMSCUT``  aperture sum of the rms map / (pi 6^2); must not exceed 1.1 x the median rms of
  the good pixels;
* negative-pixel cut: a pixel of the 11 x 11 cutout below -5 sigma with a 3 x 3 neighbour above
  +5 sigma, sigma = 1.48 x MAD of the image about its median (``filterobjects.py:155-162``).
"""
import ctypes as C

import numpy as np

from ._lib import check, ptr
from .engine import get_engine

__all__ = ['pixel_cuts', 'CUTSIZE']

CUTSIZE = 11          # pixels, zuds/filterobjects.py:12
CUT_RADIUS = 6.0      # zuds/filterobjects.py:102-104


def pixel_cuts(data, rms, bpm, x_image, y_image, engine=None):
    """dict(BPMCUT, RMSCUT, MEDCUT, NEGPIX, GOODCUT) for candidates at (x_image, y_image).

    ``GOODCUT`` holds 1 where all three pixel cuts pass (the catalog-column cuts of the
    reference are the caller's)."""
    eng = engine or get_engine()
    data = np.ascontiguousarray(data, dtype=np.float32)
    rms = np.ascontiguousarray(rms, dtype=np.float32)
    bpm = np.ascontiguousarray(bpm).astype(bool)
    x = np.ascontiguousarray(x_image, dtype=np.float64)
    y = np.ascontiguousarray(y_image, dtype=np.float64)
    n = x.size
    area = np.pi * CUT_RADIUS ** 2
    # photutils takes 0-based positions; the reference passes X_IMAGE / Y_IMAGE unchanged
    # (filterobjects.py:83-104), i.e. apertures sit one pixel high and right of the source.
    rmsbig, _, _ = eng.aperture_photometry(rms, x, y, radius=CUT_RADIUS)
    bpmbig, _, _ = eng.aperture_photometry(bpm.astype(np.float32), x, y, radius=CUT_RADIUS)
    med, _ = eng.median_mad(rms, bpm.astype(np.int32))
    medcut = 1.1 * med
    immed, immad = eng.median_mad(data)
    imsig = 1.48 * (immad / 1.4826)
    neg = np.zeros(n, np.int32)
    if n:
        ny, nx = data.shape
        check(eng.L.zm_negpix_test(eng.ctx, ptr(data), nx, ny, n, ptr(x), ptr(y), float(immed),
                                   float(imsig), ptr(neg)), 'zm_negpix_test')
    rmscut = rmsbig / area
    good = (bpmbig <= 0) & (rmscut <= medcut) & (neg == 0)
    return dict(BPMCUT=bpmbig, RMSCUT=rmscut, MEDCUT=medcut, NEGPIX=neg, GOODCUT=good.astype(np.uint8))
