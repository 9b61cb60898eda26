"""Stack combine restatement (oracle; test infrastructure).

Operator definition from the reference: ``COMBINE_TYPE CLIPPED`` with
``CLIP_SIGMA 4.0`` / ``CLIP_AMPFRAC 0.3`` for science coadds
(``zuds/astromatic/makecoadd/default.swarp:24-31``) and single-image aligns
(``zuds/swarp.py:141``); ``COMBINE_TYPE AND`` for coadd masks
(``zuds/astromatic/makecoadd/mask.swarp:25``); ``OR`` for mask aligns
(``zuds/swarp.py:141``); ``WEIGHTED``/``MEDIAN`` selectable through
``sci_swarp_kws`` (``zuds/swarp.py:76-78``).

Arithmetic (SWarp ``coadd.c`` as published; Gruen, Seitz & Bernstein 2014 for
CLIPPED): a sample is valid when its resampled weight is > 0.

* WEIGHTED: ``sum(w v) / sum(w)``, weight ``sum(w)``.
* MEDIAN: unweighted median of valid samples (even count: mean of the middle
  two); weight ``sum(w)`` of valid samples.
* CLIPPED: med = that median; reject ``|v - med| > CLIP_SIGMA / sqrt(w)
  + CLIP_AMPFRAC |med|``; result = inverse-variance weighted mean of the
  survivors, weight = their ``sum(w)``.
* no valid sample / no survivor: value 0, weight 0.
"""
import numpy as np


def _median_valid(v, valid):
    big = np.where(valid, v, np.inf)
    s = np.sort(big, axis=0)
    n = valid.sum(axis=0)
    hi = np.clip(n // 2, 0, v.shape[0] - 1)
    lo = np.clip((n - 1) // 2, 0, v.shape[0] - 1)
    a = np.take_along_axis(s, lo[None], axis=0)[0]
    b = np.take_along_axis(s, hi[None], axis=0)[0]
    med = 0.5 * (a + b)
    return np.where(n > 0, med, 0.0), n


def combine(vals, wgts, kind='CLIPPED', clip_sigma=4.0, clip_ampfrac=0.3):
    """vals, wgts: (N, ny, nx).  Returns (value, weight, nused)."""
    v = np.asarray(vals, dtype=np.float64)
    w = np.asarray(wgts, dtype=np.float64)
    valid = w > 0
    kind = kind.upper()
    if kind in ('WEIGHTED', 'AVERAGE'):
        ww = np.where(valid, w if kind == 'WEIGHTED' else 1.0, 0.0)
        s0 = ww.sum(axis=0)
        s1 = (ww * v).sum(axis=0)
        with np.errstate(invalid='ignore', divide='ignore'):
            out = np.where(s0 > 0, s1 / s0, 0.0)
        return out, np.where(valid, w, 0.0).sum(axis=0), valid.sum(axis=0)
    med, n = _median_valid(v, valid)
    if kind == 'MEDIAN':
        return med, np.where(valid, w, 0.0).sum(axis=0), n
    if kind != 'CLIPPED':
        raise ValueError(kind)
    with np.errstate(divide='ignore'):
        sig = np.where(valid, 1.0 / np.sqrt(np.where(valid, w, 1.0)), 0.0)
    keep = valid & (np.abs(v - med[None]) <= clip_sigma * sig
                    + clip_ampfrac * np.abs(med)[None])
    ww = np.where(keep, w, 0.0)
    s0 = ww.sum(axis=0)
    s1 = (ww * v).sum(axis=0)
    with np.errstate(invalid='ignore', divide='ignore'):
        out = np.where(s0 > 0, s1 / s0, 0.0)
    return out, s0, keep.sum(axis=0)


def combine_masks(masks, covered, kind='AND'):
    """masks: (N, ny, nx) int; covered: (N, ny, nx) bool (footprint in bounds).

    AND / OR over the frames that cover the pixel; uncovered everywhere -> 0
    and the returned coverage weight is 0 (the caller adds bit 16,
    ``zuds/mask.py:26-33``).
    """
    m = np.asarray(masks).astype(np.int64)
    c = np.asarray(covered, dtype=bool)
    ncov = c.sum(axis=0)
    if kind.upper() == 'OR':
        out = np.bitwise_or.reduce(np.where(c, m, 0), axis=0)
    elif kind.upper() == 'AND':
        out = np.bitwise_and.reduce(np.where(c, m, -1), axis=0)
        out = np.where(ncov > 0, out, 0)
    else:
        raise ValueError(kind)
    return out, (ncov > 0).astype(np.float64)
