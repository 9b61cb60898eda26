"""Seeded synthetic ZTF-like frames for tests and bench.

Headers are cloned from the real ZTF science header the reference ships as a
test fixture (``zuds/tests/fixtures.py:196-245``: TPV, 1.012 arcsec / px,
CD ~ -2.81e-4 deg / px) with a dithered CRPIX and a small rotation per frame.
"""
import numpy as np

from .wcs import WCS

# CD matrix and PV terms of zuds/tests/fixtures.py:203-245
ZTF_CD = np.array([[-0.0002812466181043, 1.366648840419e-06],
                   [-1.336245321655e-06, -0.0002812619006425]])
ZTF_CRVAL = (23.34894444544, 30.91859533121)
ZTF_PV1 = {0: 2.82931510214e-05, 1: 1.000000886805, 2: -1.583849820628e-05,
           4: -0.0004551962017747, 5: -8.448987011491e-05,
           6: -0.0002590727599212, 7: 0.0001271446860683,
           8: -2.218410001277e-05, 9: -0.0002238379281277,
           10: -8.6789023318149e-05, 12: 0.0007544816031555,
           13: -0.0006509589247359, 14: 0.0001397116876056,
           15: 0.0001571286145113, 16: 0.0006416466661674}
ZTF_PV2 = {0: 4.320809994094e-05, 1: 1.000088198292, 2: -2.199903872155e-05,
           4: -0.0005366569130607, 5: -0.0002406676575668,
           6: -0.000207199127297, 7: -0.0002189047536949,
           8: -0.0001038595866183, 9: -0.0002698874623035,
           10: 9.987399186654e-05, 12: 0.0007652771860817,
           13: 0.0003851918002, 14: -0.0001589248794534,
           15: -0.0003523303703531, 16: 0.0002730145790782}


def ztf_wcs(nx, ny, dx=0.0, dy=0.0, rot_deg=0.0, tpv=True, crval=ZTF_CRVAL):
    """ZTF-quadrant-like WCS of an nx x ny frame, dithered by (dx, dy) pixels
    and rotated by rot_deg about its centre."""
    c, s = np.cos(np.deg2rad(rot_deg)), np.sin(np.deg2rad(rot_deg))
    cd = ZTF_CD @ np.array([[c, -s], [s, c]])
    pv1 = pv2 = None
    if tpv:
        pv1 = np.zeros(40)
        pv2 = np.zeros(40)
        for k, v in ZTF_PV1.items():
            pv1[k] = v
        for k, v in ZTF_PV2.items():
            pv2[k] = v
    return WCS(((nx + 1) / 2.0 + dx, (ny + 1) / 2.0 + dy), crval, cd, pv1, pv2,
               (nx, ny))


def tan_wcs(nx, ny, scale=2.81e-4, crval=ZTF_CRVAL, dx=0.0, dy=0.0):
    """Config-1 style shared TAN WCS (CRPIX at the centre, CD diag(-s, s))."""
    return WCS(((nx + 1) / 2.0 + dx, (ny + 1) / 2.0 + dy), crval,
               [-scale, 0.0, 0.0, scale], None, None, (nx, ny))


def add_stars(img, x, y, flux, fwhm):
    """Add circular Gaussians (pixel-centre sampled) in place."""
    ny, nx = img.shape
    sig = np.broadcast_to(np.asarray(fwhm, dtype=np.float64) / 2.3548200450309493,
                          np.shape(x))
    for xs, ys, f, s in zip(x, y, flux, sig):
        r = int(np.ceil(5 * s)) + 1
        x0, x1 = max(int(xs) - r, 0), min(int(xs) + r + 1, nx)
        y0, y1 = max(int(ys) - r, 0), min(int(ys) + r + 1, ny)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        g = np.exp(-0.5 * ((xx - xs) ** 2 + (yy - ys) ** 2) / (s * s))
        img[y0:y1, x0:x1] += (f / (2 * np.pi * s * s)) * g
    return img


def make_frame(nx, ny, seed, wcs, sky=150.0, noise=5.0, nstars=40, fwhm=2.0,
               flux_range=(1e3, 1e5), magzp=25.0, star_sky=None, nbad=0,
               bad_block=None, dtype=np.float32):
    """One synthetic science frame: dict(img, wgt, mask, wcs, header, flxscale).

    ``star_sky``: optional (ra, dec, flux) shared by a set of dithered frames so
    that the same stars appear in each of them through its own WCS.
    """
    rng = np.random.default_rng(seed)
    img = np.full((ny, nx), float(sky))
    if star_sky is not None:
        ra, dec, flux = star_sky
        xs, ys = wcs.all_world2pix(ra, dec, 0)
        fl = np.asarray(flux) * 10 ** (0.4 * (magzp - 25.0))
    else:
        xs = rng.uniform(0, nx - 1, nstars)
        ys = rng.uniform(0, ny - 1, nstars)
        fl = np.exp(rng.uniform(np.log(flux_range[0]), np.log(flux_range[1]),
                                nstars))
    add_stars(img, xs, ys, fl, fwhm)
    img += rng.normal(0.0, noise, img.shape)
    var = np.full((ny, nx), float(noise) ** 2)
    mask = np.zeros((ny, nx), dtype=np.int32)
    if bad_block is not None:
        bx, by, bs = bad_block
        mask[by:by + bs, bx:bx + bs] |= 1 << 8
    if nbad:
        bxs = rng.integers(0, nx, nbad)
        bys = rng.integers(0, ny, nbad)
        bits = rng.choice([1 << 0, 1 << 8], nbad)
        mask[bys, bxs] |= bits.astype(np.int32)
    bad = (mask & 198589) > 0     # BAD_SUM, zuds/constants.py:45-46
    wgt = np.where(bad, 0.0, 1.0 / var)
    header = {'SIMPLE': True, 'BITPIX': -32, 'NAXIS': 2, 'NAXIS1': nx,
              'NAXIS2': ny, 'MAGZP': float(magzp), 'SEEING': float(fwhm),
              'GAIN': 6.2, 'SATURATE': 48059.879, 'APCOR4': -0.096003,
              'OBSMJD': 58000.0 + seed * 1e-3, 'FIELD': 651, 'CCDID': 3,
              'QID': 1, 'FID': 1}
    header.update(wcs.to_header())
    return dict(img=img.astype(dtype), wgt=wgt.astype(dtype), mask=mask,
                wcs=wcs, header=header,
                flxscale=10 ** (-0.4 * (magzp - 25.0)))


def config1(n=4, nx=512, ny=512):
    """BASELINE config 1: n frames, shared TAN WCS, sky 150 + N(0, 5^2), 40
    stars FWHM 2.0 px, MAGZP 25, one 5x5 bad block (bit 8) per frame."""
    w = tan_wcs(nx, ny)
    rng = np.random.default_rng(1234)
    xs = rng.uniform(10, nx - 10, 40)
    ys = rng.uniform(10, ny - 10, 40)
    fl = np.exp(rng.uniform(np.log(1e3), np.log(1e5), 40))
    ra, dec = w.all_pix2world(xs, ys, 0)
    frames = []
    for i in range(n):
        frames.append(make_frame(nx, ny, 1234 + i, w, star_sky=(ra, dec, fl),
                                 bad_block=(50 + 60 * i, 80 + 40 * i, 5)))
    return frames


def config2(n=32, nx=3072, ny=3072, nstars=3000, seed0=2000, tpv=True,
            dither=15.0, rot=0.1):
    """BASELINE config 2: n dithered / rotated TPV frames of one star field."""
    base = ztf_wcs(nx, ny, tpv=tpv)
    rng = np.random.default_rng(seed0 - 1)
    xs = rng.uniform(-20, nx + 20, nstars)
    ys = rng.uniform(-20, ny + 20, nstars)
    fl = np.exp(rng.uniform(np.log(1e3), np.log(1e5), nstars))
    ra, dec = base.all_pix2world(xs, ys, 0)
    frames = []
    for i in range(n):
        r = np.random.default_rng(seed0 + i)
        w = ztf_wcs(nx, ny, dx=r.uniform(-dither, dither),
                    dy=r.uniform(-dither, dither), rot_deg=r.uniform(-rot, rot),
                    tpv=tpv)
        sky = r.uniform(100, 300)
        frames.append(make_frame(nx, ny, seed0 + i, w, sky=sky,
                                 noise=np.sqrt(sky / 6.2),
                                 fwhm=r.uniform(1.8, 2.6),
                                 magzp=r.uniform(25.8, 26.6),
                                 star_sky=(ra, dec, fl),
                                 nbad=int(1e-3 * nx * ny)))
    return frames
