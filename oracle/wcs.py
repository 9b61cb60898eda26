"""TAN / TPV world coordinate systems in float64 (oracle; test infrastructure).

The reference builds ``astropy.wcs.WCS(header)`` (``zuds/fitsfile.py:233-238``)
and hands headers to SWarp, whose output projection is ``PROJECTION_TYPE TPV``
(``zuds/astromatic/makecoadd/default.swarp:43``).  ZTF science frames carry
``CTYPE RA---TPV / DEC--TPV`` with a CD matrix and ``PV1_0..PV2_16``
(``zuds/tests/fixtures.py:196-245``).  This module restates the FITS-WCS
TAN projection (Calabretta & Greisen 2002) and the TPV distortion polynomial
(the SCAMP/registry convention: PVi_1 defaults to 1, every other PV to 0).

Pixel coordinates are FITS 1-based (centre of the first pixel = 1.0).
"""
import numpy as np

D2R = np.pi / 180.0
NPV = 40

# exponents (px, py, pr) of each TPV term for axis 1 in (x, y, r); axis 2 swaps
# the roles of x and y.
_TPV_TERMS = [
    (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1),
    (2, 0, 0), (1, 1, 0), (0, 2, 0),
    (3, 0, 0), (2, 1, 0), (1, 2, 0), (0, 3, 0), (0, 0, 3),
    (4, 0, 0), (3, 1, 0), (2, 2, 0), (1, 3, 0), (0, 4, 0),
    (5, 0, 0), (4, 1, 0), (3, 2, 0), (2, 3, 0), (1, 4, 0), (0, 5, 0), (0, 0, 5),
    (6, 0, 0), (5, 1, 0), (4, 2, 0), (3, 3, 0), (2, 4, 0), (1, 5, 0), (0, 6, 0),
    (7, 0, 0), (6, 1, 0), (5, 2, 0), (4, 3, 0), (3, 4, 0), (2, 5, 0), (1, 6, 0),
    (0, 7, 0), (0, 0, 7),
]
assert len(_TPV_TERMS) == NPV


def tpv_eval(pv, x, y):
    """Value and partial derivatives (d/dx, d/dy) of one TPV polynomial."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    f = np.zeros(np.broadcast(x, y).shape)
    fx = np.zeros_like(f)
    fy = np.zeros_like(f)
    need_r = any(pv[k] != 0.0 for k in (3, 11, 23, 39))
    if need_r:
        r = np.sqrt(x * x + y * y)
        rs = np.where(r > 0, r, 1.0)
    for k, (a, b, c) in enumerate(_TPV_TERMS):
        p = pv[k]
        if p == 0.0:
            continue
        if c:
            f = f + p * r ** c
            g = p * c * r ** (c - 1) / rs
            fx = fx + g * x
            fy = fy + g * y
        else:
            f = f + p * x ** a * y ** b
            if a:
                fx = fx + p * a * x ** (a - 1) * y ** b
            if b:
                fy = fy + p * b * x ** a * y ** (b - 1)
    return f, fx, fy


class WCS(object):
    """TAN (flags=0) or TPV (flags=1) celestial WCS."""

    def __init__(self, crpix, crval, cd, pv1=None, pv2=None, naxis=(0, 0)):
        self.crpix = np.asarray(crpix, dtype=np.float64)
        self.crval = np.asarray(crval, dtype=np.float64)
        self.cd = np.asarray(cd, dtype=np.float64).reshape(2, 2)
        self.has_pv = pv1 is not None or pv2 is not None
        self.pv1 = np.zeros(NPV)
        self.pv2 = np.zeros(NPV)
        self.pv1[1] = 1.0
        self.pv2[1] = 1.0
        if pv1 is not None:
            self.pv1[:] = pv1
        if pv2 is not None:
            self.pv2[:] = pv2
        self.naxis = (int(naxis[0]), int(naxis[1]))

    # -- construction -----------------------------------------------------
    @classmethod
    def from_header(cls, header):
        """Build from a FITS header dict (CDi_j or CDELT/PC; PVi_k optional)."""
        h = header
        crpix = (float(h['CRPIX1']), float(h['CRPIX2']))
        crval = (float(h['CRVAL1']), float(h['CRVAL2']))
        if 'CD1_1' in h:
            cd = [h.get('CD1_1', 0.0), h.get('CD1_2', 0.0),
                  h.get('CD2_1', 0.0), h.get('CD2_2', 0.0)]
        else:
            d1 = float(h.get('CDELT1', 1.0))
            d2 = float(h.get('CDELT2', 1.0))
            cd = [d1 * h.get('PC1_1', 1.0), d1 * h.get('PC1_2', 0.0),
                  d2 * h.get('PC2_1', 0.0), d2 * h.get('PC2_2', 1.0)]
        pv1 = pv2 = None
        if any(k.startswith('PV1_') or k.startswith('PV2_') for k in h):
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

    def to_header(self):
        h = {'CTYPE1': 'RA---TPV' if self.has_pv else 'RA---TAN',
             'CTYPE2': 'DEC--TPV' if self.has_pv else 'DEC--TAN',
             'CRPIX1': float(self.crpix[0]), 'CRPIX2': float(self.crpix[1]),
             'CRVAL1': float(self.crval[0]), 'CRVAL2': float(self.crval[1]),
             'CD1_1': float(self.cd[0, 0]), 'CD1_2': float(self.cd[0, 1]),
             'CD2_1': float(self.cd[1, 0]), 'CD2_2': float(self.cd[1, 1]),
             'CUNIT1': 'deg', 'CUNIT2': 'deg'}
        if self.has_pv:
            for k in range(NPV):
                if self.pv1[k] != 0.0:
                    h[f'PV1_{k}'] = float(self.pv1[k])
                if self.pv2[k] != 0.0:
                    h[f'PV2_{k}'] = float(self.pv2[k])
        return h

    # -- tangent-plane frame ------------------------------------------------
    def frame(self):
        """Rows: east, north, pole unit vectors of the tangent frame at CRVAL."""
        a0 = self.crval[0] * D2R
        d0 = self.crval[1] * D2R
        sa, ca = np.sin(a0), np.cos(a0)
        sd, cd = np.sin(d0), np.cos(d0)
        east = np.array([-sa, ca, 0.0])
        north = np.array([-sd * ca, -sd * sa, cd])
        pole = np.array([cd * ca, cd * sa, sd])
        return np.stack([east, north, pole])

    # -- intermediate <-> projection-plane ---------------------------------
    def pix2plane(self, x, y):
        """Pixel -> projection-plane (xi, eta) in degrees."""
        dx = np.asarray(x, dtype=np.float64) - self.crpix[0]
        dy = np.asarray(y, dtype=np.float64) - self.crpix[1]
        u = self.cd[0, 0] * dx + self.cd[0, 1] * dy
        v = self.cd[1, 0] * dx + self.cd[1, 1] * dy
        if not self.has_pv:
            return u, v
        xi = tpv_eval(self.pv1, u, v)[0]
        eta = tpv_eval(self.pv2, v, u)[0]
        return xi, eta

    def plane2pix(self, xi, eta, niter=20, tol=1e-13):
        """Projection-plane (xi, eta) degrees -> pixel; Newton for TPV."""
        xi = np.asarray(xi, dtype=np.float64)
        eta = np.asarray(eta, dtype=np.float64)
        if self.has_pv:
            # start from the linear part
            u = (xi - self.pv1[0]) / self.pv1[1]
            v = (eta - self.pv2[0]) / self.pv2[1]
            for _ in range(niter):
                f, fu, fv = tpv_eval(self.pv1, u, v)
                g, gv, gu = tpv_eval(self.pv2, v, u)
                rf = f - xi
                rg = g - eta
                det = fu * gv - fv * gu
                du = (rf * gv - rg * fv) / det
                dv = (rg * fu - rf * gu) / det
                u = u - du
                v = v - dv
                if np.max(np.abs(du)) < tol and np.max(np.abs(dv)) < tol:
                    break
        else:
            u, v = xi, eta
        det = self.cd[0, 0] * self.cd[1, 1] - self.cd[0, 1] * self.cd[1, 0]
        x = (self.cd[1, 1] * u - self.cd[0, 1] * v) / det + self.crpix[0]
        y = (-self.cd[1, 0] * u + self.cd[0, 0] * v) / det + self.crpix[1]
        return x, y

    # -- sky ---------------------------------------------------------------
    def pix2vec(self, x, y):
        xi, eta = self.pix2plane(x, y)
        fr = self.frame()
        xr = xi * D2R
        er = eta * D2R
        vec = (xr[..., None] * fr[0] + er[..., None] * fr[1] + fr[2])
        return vec / np.linalg.norm(vec, axis=-1, keepdims=True)

    def vec2pix(self, vec):
        fr = self.frame()
        c = vec @ fr[2]
        xi = (vec @ fr[0]) / c / D2R
        eta = (vec @ fr[1]) / c / D2R
        return self.plane2pix(xi, eta)

    def pix2sky(self, x, y):
        v = self.pix2vec(np.asarray(x, dtype=np.float64),
                         np.asarray(y, dtype=np.float64))
        ra = np.arctan2(v[..., 1], v[..., 0]) / D2R
        ra = np.where(ra < 0, ra + 360.0, ra)
        dec = np.arcsin(np.clip(v[..., 2], -1, 1)) / D2R
        return ra, dec

    def sky2pix(self, ra, dec):
        a = np.asarray(ra, dtype=np.float64) * D2R
        d = np.asarray(dec, dtype=np.float64) * D2R
        v = np.stack([np.cos(d) * np.cos(a), np.cos(d) * np.sin(a), np.sin(d)],
                     axis=-1)
        return self.vec2pix(v)

    # -- metrics -----------------------------------------------------------
    def pixel_area(self, x=None, y=None):
        """|d(xi,eta)/d(x,y)| in deg^2 at pixel (x, y) (default: centre)."""
        if x is None:
            x = (self.naxis[0] + 1) / 2.0
            y = (self.naxis[1] + 1) / 2.0
        h = 0.5
        x1, e1 = self.pix2plane(x + h, y)
        x0, e0 = self.pix2plane(x - h, y)
        x3, e3 = self.pix2plane(x, y + h)
        x2, e2 = self.pix2plane(x, y - h)
        j11, j21 = (x1 - x0), (e1 - e0)
        j12, j22 = (x3 - x2), (e3 - e2)
        return float(abs(j11 * j22 - j12 * j21))

    def pixel_scale(self):
        """Pixel scale in degrees (sqrt of the central pixel area)."""
        return float(np.sqrt(self.pixel_area()))

    def footprint(self):
        """Sky positions of the four outer pixel corners (FITS convention)."""
        nx, ny = self.naxis
        xs = np.array([0.5, 0.5, nx + 0.5, nx + 0.5])
        ys = np.array([0.5, ny + 0.5, ny + 0.5, 0.5])
        return self.pix2sky(xs, ys)


def map_out_to_in(wout, win, xo, yo):
    """Input-frame pixel position (1-based) of output pixels (1-based).

    This is the inverse mapping SWarp evaluates for every output pixel
    (call sites ``zuds/coadd.py:133``, ``zuds/swarp.py:175``); the sky is never
    materialised: the two tangent frames are linked by one 3x3 rotation.
    """
    xi, eta = wout.pix2plane(xo, yo)
    m = win.frame() @ wout.frame().T      # rows: in-frame axes in out-frame basis
    xr = xi * D2R
    er = eta * D2R
    a = m[0, 0] * xr + m[0, 1] * er + m[0, 2]
    b = m[1, 0] * xr + m[1, 1] * er + m[1, 2]
    c = m[2, 0] * xr + m[2, 1] * er + m[2, 2]
    return win.plane2pix(a / c / D2R, b / c / D2R)
