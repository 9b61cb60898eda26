"""Header-level WCS object of the host layer.

Stands in for ``astropy.wcs.WCS(header)`` (``zuds/fitsfile.py:233-238``) for
the TAN / TPV headers this path sees (``zuds/tests/fixtures.py:196-245``).  All
projection arithmetic is done by libzudsmi (``zm_wcs_pix2sky`` ...); this
class only parses and emits header cards.
"""
import ctypes as C

import numpy as np

from . import _lib

NPV = _lib.NPV


class WCS(object):

    def __init__(self, crpix, crval, cd, pv1=None, pv2=None, naxis=(0, 0)):
        self.crpix = np.asarray(crpix, dtype=np.float64)
        self.crval = np.asarray(crval, dtype=np.float64)
        self.cd = np.asarray(cd, dtype=np.float64).reshape(2, 2)
        self.has_pv = pv1 is not None or pv2 is not None
        self.pv1 = np.zeros(NPV)
        self.pv2 = np.zeros(NPV)
        self.pv1[1] = self.pv2[1] = 1.0
        if pv1 is not None:
            self.pv1[:] = pv1
        if pv2 is not None:
            self.pv2[:] = pv2
        self.naxis = (int(naxis[0]), int(naxis[1]))

    @classmethod
    def from_header(cls, h):
        for k in ('CRPIX1', 'CRPIX2', 'CRVAL1', 'CRVAL2'):
            if k not in h:
                raise ValueError(f'header has no WCS solution (missing {k})')
        crpix = (float(h['CRPIX1']), float(h['CRPIX2']))
        crval = (float(h['CRVAL1']), float(h['CRVAL2']))
        if 'CD1_1' in h or 'CD2_2' in h:
            cd = [float(h.get('CD1_1', 0.0)), float(h.get('CD1_2', 0.0)),
                  float(h.get('CD2_1', 0.0)), float(h.get('CD2_2', 0.0))]
        else:
            d1 = float(h.get('CDELT1', 1.0))
            d2 = float(h.get('CDELT2', 1.0))
            cd = [d1 * float(h.get('PC1_1', 1.0)), d1 * float(h.get('PC1_2', 0.0)),
                  d2 * float(h.get('PC2_1', 0.0)), d2 * float(h.get('PC2_2', 1.0))]
        pv1 = pv2 = None
        if any(str(k).startswith(('PV1_', 'PV2_')) for k in h):
            pv1 = np.zeros(NPV)
            pv2 = np.zeros(NPV)
            pv1[1] = pv2[1] = 1.0
            for k in range(NPV):
                if f'PV1_{k}' in h:
                    pv1[k] = float(h[f'PV1_{k}'])
                if f'PV2_{k}' in h:
                    pv2[k] = float(h[f'PV2_{k}'])
        return cls(crpix, crval, cd, pv1, pv2,
                   (h.get('NAXIS1', 0), h.get('NAXIS2', 0)))

    @classmethod
    def from_struct(cls, s):
        pv1 = list(s.pv1) if s.flags & 1 else None
        pv2 = list(s.pv2) if s.flags & 1 else None
        return cls(list(s.crpix), list(s.crval), list(s.cd), pv1, pv2,
                   tuple(s.naxis))

    def to_header(self, relax=True):
        """WCS cards, the analogue of ``WCS(header).to_header(relax=True)``
        written to SWarp's ``.head`` file (``zuds/swarp.py:114-133``)."""
        h = {'WCSAXES': 2,
             'CTYPE1': 'RA---TPV' if self.has_pv else 'RA---TAN',
             'CTYPE2': 'DEC--TPV' if self.has_pv else 'DEC--TAN',
             'CRPIX1': float(self.crpix[0]), 'CRPIX2': float(self.crpix[1]),
             'CRVAL1': float(self.crval[0]), 'CRVAL2': float(self.crval[1]),
             'CUNIT1': 'deg', 'CUNIT2': 'deg',
             'CD1_1': float(self.cd[0, 0]), 'CD1_2': float(self.cd[0, 1]),
             'CD2_1': float(self.cd[1, 0]), 'CD2_2': float(self.cd[1, 1])}
        if self.has_pv:
            for k in range(NPV):
                if self.pv1[k] != 0.0:
                    h[f'PV1_{k}'] = float(self.pv1[k])
            for k in range(NPV):
                if self.pv2[k] != 0.0:
                    h[f'PV2_{k}'] = float(self.pv2[k])
        return h

    # -- projection through the C-ABI -------------------------------------
    def _call(self, fn, a, b):
        a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
        b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
        o1 = np.empty_like(a)
        o2 = np.empty_like(b)
        s = _lib.wcs_struct(self)
        _lib.check(fn(C.byref(s), a.size, _lib.ptr(a), _lib.ptr(b),
                      _lib.ptr(o1), _lib.ptr(o2)))
        return o1, o2

    def all_pix2world(self, x, y, origin=1):
        """(ra, dec) in degrees of pixel coordinates (``origin`` 0 or 1)."""
        x = np.asarray(x, dtype=np.float64) + (1 - origin)
        y = np.asarray(y, dtype=np.float64) + (1 - origin)
        shp = np.broadcast(x, y).shape
        x, y = np.broadcast_arrays(x, y)
        ra, dec = self._call(_lib.lib().zm_wcs_pix2sky, x.ravel(), y.ravel())
        return ra.reshape(shp), dec.reshape(shp)

    def all_world2pix(self, ra, dec, origin=1):
        ra = np.asarray(ra, dtype=np.float64)
        dec = np.asarray(dec, dtype=np.float64)
        shp = np.broadcast(ra, dec).shape
        ra, dec = np.broadcast_arrays(ra, dec)
        x, y = self._call(_lib.lib().zm_wcs_sky2pix, ra.ravel(), dec.ravel())
        return (x - (1 - origin)).reshape(shp), (y - (1 - origin)).reshape(shp)

    def calc_footprint(self):
        """Sky positions of the centres of the four corner pixels, in the order
        and with the ``center=True`` default of astropy's ``calc_footprint``
        (used by ``zuds/fitsfile.py:247``; pinned by tests/golden/astropy_wcs.json)."""
        nx, ny = self.naxis
        xs = np.array([1.0, 1.0, nx, nx], dtype=np.float64)
        ys = np.array([1.0, ny, ny, 1.0], dtype=np.float64)
        ra, dec = self.all_pix2world(xs, ys, 1)
        return np.stack([ra, dec], axis=1)

    def proj_plane_pixel_scales(self):
        """Degrees per pixel along x and y (``zuds/fitsfile.py:283-288``)."""
        return np.sqrt((self.cd ** 2).sum(axis=0))
