"""Forced circular-aperture photometry restatement (oracle; test infrastructure).

Operator definition from the reference (``zuds/photometry.py:61-113,116-249``):
``photutils.aperture_photometry(data, CircularAperture(r = 3 px), error = rms)``
with the default ``method='exact'`` (``APERTURE_RADIUS``, ``zuds/constants.py:14``):
``flux = sum(data * frac)``, ``fluxerr = sqrt(sum(rms^2 * frac))`` where ``frac`` is
the exact fraction of each pixel inside the circle; ``flags`` = bitwise OR of the
mask over the aperture's bounding box (``to_mask(method='center').cutout(mask)``
returns the raw cutout, not the circle).  photutils itself is absent here; the
exact circle / pixel overlap below is the closed-form area (quarter-box
decomposition), pinned by analytic tests (sum of fractions = pi r^2).
"""
import numpy as np


def _quarter(x, y, r):
    """Area of circle(r) & [0, x] x [0, y] for x, y >= 0."""
    x = np.minimum(x, r)
    y = np.minimum(y, r)
    inside = x * x + y * y <= r * r
    xc = np.sqrt(np.maximum(r * r - y * y, 0.0))      # circle crosses height y at xc
    xm = np.minimum(x, xc)

    def P(u):
        return 0.5 * (u * np.sqrt(np.maximum(r * r - u * u, 0.0)) + r * r * np.arcsin(np.clip(u / r, -1, 1)))
    a = y * xm + P(x) - P(xm)
    return np.where(inside, x * y, a)


def _signed(x, y, r):
    return np.sign(x) * np.sign(y) * _quarter(np.abs(x), np.abs(y), r)


def overlap_fraction(x0, x1, y0, y1, r):
    """Exact area of circle(r, centre 0) & [x0, x1] x [y0, y1], per unit box area."""
    a = _signed(x1, y1, r) - _signed(x0, y1, r) - _signed(x1, y0, r) + _signed(x0, y0, r)
    return a / ((x1 - x0) * (y1 - y0))


def bbox(xc, yc, r):
    """photutils BoundingBox.from_float(x - r, x + r, y - r, y + r): [imin, imax)."""
    ixmin = int(np.floor(xc - r + 0.5))
    ixmax = int(np.ceil(xc + r + 0.5))
    iymin = int(np.floor(yc - r + 0.5))
    iymax = int(np.ceil(yc + r + 0.5))
    return ixmin, ixmax, iymin, iymax


def aperture_photometry(data, rms, mask, x, y, r=3.0):
    """x, y: 0-based pixel positions.  Returns flux, fluxerr, flags arrays."""
    data = np.asarray(data, dtype=np.float64)
    rms = np.asarray(rms, dtype=np.float64)
    ny, nx = data.shape
    flux = np.zeros(len(x))
    err = np.zeros(len(x))
    flags = np.zeros(len(x), dtype=np.int64)
    for k, (xc, yc) in enumerate(zip(x, y)):
        ixmin, ixmax, iymin, iymax = bbox(xc, yc, r)
        i0, i1 = max(ixmin, 0), min(ixmax, nx)
        j0, j1 = max(iymin, 0), min(iymax, ny)
        if i0 >= i1 or j0 >= j1:
            continue
        jj, ii = np.mgrid[j0:j1, i0:i1]
        frac = overlap_fraction(ii - 0.5 - xc, ii + 0.5 - xc, jj - 0.5 - yc, jj + 0.5 - yc, r)
        flux[k] = (data[j0:j1, i0:i1] * frac).sum()
        err[k] = np.sqrt((rms[j0:j1, i0:i1] ** 2 * frac).sum())
        if mask is not None:
            flags[k] = int(np.bitwise_or.reduce(np.asarray(mask)[j0:j1, i0:i1].astype(np.int64), axis=(0, 1)))
    return flux, err, flags
