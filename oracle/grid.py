"""Automatic output grid restatement (oracle; test infrastructure).

Operator definition from the reference: ``CENTER_TYPE ALL``,
``PIXELSCALE_TYPE MEDIAN``, ``IMAGE_SIZE 0``, ``PROJECTION_TYPE TPV`` (= TAN
when the output carries no PV terms), ``CELESTIAL_TYPE NATIVE``
(``zuds/astromatic/makecoadd/default.swarp:40-49``).

Adopted convention: CRVAL = midpoint of the RA and Dec ranges spanned by the
outer pixel borders of all inputs; CD = diag(-s, +s) with s the median input
pixel scale; NAXISn = ceil of the bounding-box extent, in output pixels, of all
input borders projected on that tangent plane; CRPIX centres the bounding box.
"""
import numpy as np

from .wcs import WCS


def border_points(w, step=64):
    nx, ny = w.naxis
    xs = np.unique(np.concatenate([np.arange(0.5, nx + 0.5, step), [nx + 0.5]]))
    ys = np.unique(np.concatenate([np.arange(0.5, ny + 0.5, step), [ny + 0.5]]))
    bx = np.concatenate([xs, xs, np.full_like(ys, 0.5), np.full_like(ys, nx + 0.5)])
    by = np.concatenate([np.full_like(xs, 0.5), np.full_like(xs, ny + 0.5), ys, ys])
    return bx, by


def autogrid(wcss):
    ras, decs, scales = [], [], []
    ra0 = wcss[0].crval[0]
    for w in wcss:
        bx, by = border_points(w)
        ra, dec = w.pix2sky(bx, by)
        ra = (ra - ra0 + 180.0) % 360.0 - 180.0 + ra0
        ras.append(ra)
        decs.append(dec)
        scales.append(w.pixel_scale())
    ras = np.concatenate(ras)
    decs = np.concatenate(decs)
    cra = 0.5 * (ras.min() + ras.max())
    cdec = 0.5 * (decs.min() + decs.max())
    s = float(np.median(scales))
    out = WCS((0.0, 0.0), (cra % 360.0, cdec), [-s, 0.0, 0.0, s])
    x, y = out.sky2pix(ras, decs)
    ex = x.max() - x.min()
    ey = y.max() - y.min()
    nx = max(int(np.ceil(ex - 1e-9)), 1)
    ny = max(int(np.ceil(ey - 1e-9)), 1)
    out.crpix = np.array([0.5 - x.min() + 0.5 * (nx - ex),
                          0.5 - y.min() + 0.5 * (ny - ey)])
    out.naxis = (nx, ny)
    return out
