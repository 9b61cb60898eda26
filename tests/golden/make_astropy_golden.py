#!/opt/conda/bin/python3.9
"""Golden vectors from astropy 4.3.1 (wcslib + astropy.io.fits), the library the
reference uses for ``HasWCS.wcs`` (``zuds/fitsfile.py:233-238``), for the
``.head`` files SWarp aligns to (``zuds/swarp.py:114-133``) and for FITS I/O
(``zuds/fitsfile.py:69-206``).

Run in the build container only (astropy lives in a stale conda tree that does
not travel):

    env -u PYTHONHOME -u PYTHONPATH /opt/conda/bin/python3.9 tests/golden/make_astropy_golden.py

Outputs (committed, read by tests/test_golden.py):
    astropy_wcs.json     headers + pixel <-> sky vectors (TPV and TAN)
    astropy_f32.fits     float32 image + ZTF-like header written by astropy.io.fits
    astropy_i16.fits     int16 mask image (BITPIX 16)
    astropy_u8.fits      uint8 image (BITPIX 8), how the reference stores boolean
                         bad-pixel maps (zuds/fitsfile.py:175-185)
    astropy_fits.json    the pixel values / header values astropy reads back, and
                         the result of astropy reading files written by OUR writer

The header values are those of the real ZTF science header the reference ships
as a test fixture (zuds/tests/fixtures.py:196-245): data, not code.
"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

if not hasattr(np, 'asscalar'):
    np.asscalar = lambda a: a.item()
if not hasattr(np, 'alen'):
    np.alen = len

from astropy.io import fits          # noqa: E402
from astropy.wcs import WCS          # noqa: E402
import astropy                       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

ZTF = {
    'CTYPE1': 'RA---TPV', 'CTYPE2': 'DEC--TPV',
    'CRPIX1': 1536.5, 'CRPIX2': 1540.5,
    'CRVAL1': 23.34894444544, 'CRVAL2': 30.91859533121,
    'CUNIT1': 'deg', 'CUNIT2': 'deg',
    'CD1_1': -0.0002812466181043, 'CD1_2': 1.366648840419e-06,
    'CD2_1': -1.336245321655e-06, 'CD2_2': -0.0002812619006425,
    'PV1_0': 2.82931510214e-05, 'PV1_1': 1.000000886805, 'PV1_2': -1.583849820628e-05,
    'PV1_4': -0.0004551962017747, 'PV1_5': -8.448987011491e-05,
    'PV1_6': -0.0002590727599212, 'PV1_7': 0.0001271446860683,
    'PV1_8': -2.218410001277e-05, 'PV1_9': -0.0002238379281277,
    'PV1_10': -8.6789023318149e-05, 'PV1_12': 0.0007544816031555,
    'PV1_13': -0.0006509589247359, 'PV1_14': 0.0001397116876056,
    'PV1_15': 0.0001571286145113, 'PV1_16': 0.0006416466661674,
    'PV2_0': 4.320809994094e-05, 'PV2_1': 1.000088198292, 'PV2_2': -2.199903872155e-05,
    'PV2_4': -0.0005366569130607, 'PV2_5': -0.0002406676575668,
    'PV2_6': -0.000207199127297, 'PV2_7': -0.0002189047536949,
    'PV2_8': -0.0001038595866183, 'PV2_9': -0.0002698874623035,
    'PV2_10': 9.987399186654e-05, 'PV2_12': 0.0007652771860817,
    'PV2_13': 0.0003851918002, 'PV2_14': -0.0001589248794534,
    'PV2_15': -0.0003523303703531, 'PV2_16': 0.0002730145790782,
}
NX, NY = 3072, 3080


def variants():
    """The fixture header, dithered / rotated clones of it (what config 2 uses),
    a high-declination clone and a plain TAN header."""
    out = [('ztf_fixture', dict(ZTF, NAXIS1=NX, NAXIS2=NY))]
    rng = np.random.default_rng(20261003)
    for i in range(3):
        h = dict(ZTF, NAXIS1=NX, NAXIS2=NY)
        h['CRPIX1'] += rng.uniform(-15, 15)
        h['CRPIX2'] += rng.uniform(-15, 15)
        a = np.deg2rad(rng.uniform(-0.1, 0.1))
        cd = np.array([[h['CD1_1'], h['CD1_2']], [h['CD2_1'], h['CD2_2']]]) @ \
            np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        h['CD1_1'], h['CD1_2'], h['CD2_1'], h['CD2_2'] = [float(v) for v in cd.ravel()]
        out.append((f'ztf_dither{i}', h))
    hi = dict(ZTF, NAXIS1=NX, NAXIS2=NY, CRVAL1=359.98, CRVAL2=78.25)
    out.append(('ztf_highdec_wrap', hi))
    tan = {k: v for k, v in ZTF.items() if not k.startswith('PV')}
    tan.update(CTYPE1='RA---TAN', CTYPE2='DEC--TAN', NAXIS1=512, NAXIS2=512,
               CRPIX1=256.5, CRPIX2=256.5, CD1_1=-2.81e-4, CD1_2=0.0, CD2_1=0.0, CD2_2=2.81e-4)
    out.append(('tan_config1', tan))
    return out


def wcs_vectors():
    rng = np.random.default_rng(7)
    recs = []
    for name, h in variants():
        hdr = fits.Header()
        for k, v in h.items():
            hdr[k] = v
        w = WCS(hdr)
        nx, ny = h['NAXIS1'], h['NAXIS2']
        x = np.concatenate([[1.0, nx, 1.0, nx, 0.5, nx + 0.5, (nx + 1) / 2.0],
                            rng.uniform(-40, nx + 40, 57)])
        y = np.concatenate([[1.0, 1.0, ny, ny, 0.5, ny + 0.5, (ny + 1) / 2.0],
                            rng.uniform(-40, ny + 40, 57)])
        ra, dec = w.all_pix2world(x, y, 1)
        # inverse: wcslib inverts TPV iteratively; keep its answer and how well it closes
        xb, yb = w.all_world2pix(ra, dec, 1)
        recs.append(dict(name=name, header=h, x=x.tolist(), y=y.tolist(),
                         ra=np.asarray(ra).tolist(), dec=np.asarray(dec).tolist(),
                         x_back=np.asarray(xb).tolist(), y_back=np.asarray(yb).tolist(),
                         footprint=np.asarray(w.calc_footprint()).tolist(),
                         pixel_scales=np.asarray(
                             astropy.wcs.utils.proj_plane_pixel_scales(w)).tolist()))
    return recs


def load_our_fits():
    spec = importlib.util.spec_from_file_location(
        'zm_fits', os.path.join(ROOT, 'zuds-pipeline_amd', 'fits.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def fits_vectors():
    rng = np.random.default_rng(11)
    img = rng.normal(150.0, 5.0, (37, 53)).astype(np.float32)
    img[3, 4] = np.nan
    img[5, 6] = 1e-30
    i16 = rng.integers(-300, 3000, (37, 53)).astype(np.int16)
    u8 = (rng.uniform(size=(37, 53)) < 0.2).astype(np.uint8)
    hdr = fits.Header()
    for k, v in ZTF.items():
        hdr[k] = v
    hdr['MAGZP'] = (26.123456789, 'zero point')
    hdr['SEEING'] = 2.25
    hdr['SATURATE'] = 55000.0
    hdr['FIELD'] = 651
    hdr['FILTER'] = 'ZTF_r'
    hdr['PHOTLINK'] = False
    hdr['LONGSTR'] = 'gaia_000651_c03_q1.fits'
    hdr['OBSJD'] = 2458383.7790278
    hdr['NEGEXP'] = -1.5e-07
    fits.PrimaryHDU(img, hdr).writeto(os.path.join(HERE, 'astropy_f32.fits'), overwrite=True)
    fits.PrimaryHDU(i16).writeto(os.path.join(HERE, 'astropy_i16.fits'), overwrite=True)
    fits.PrimaryHDU(u8).writeto(os.path.join(HERE, 'astropy_u8.fits'), overwrite=True)
    out = {'f32_sum': float(np.nansum(img.astype(np.float64))), 'f32_shape': list(img.shape),
           'f32_sample': img[:4, :5].astype(np.float64).tolist(),
           'i16_sample': i16[:4, :5].tolist(), 'i16_sum': int(i16.astype(np.int64).sum()),
           'u8_sum': int(u8.sum()),
           'header': {k: (v if not isinstance(v, (np.floating, np.integer)) else v.item())
                      for k, v in hdr.items()}}
    # the reverse direction: astropy reads what OUR writer produces
    zf = load_our_fits()
    rev = {}
    with tempfile.TemporaryDirectory() as d:
        for nm, arr in (('f32', img), ('i16', i16), ('u8', u8),
                        ('i32', (i16.astype(np.int32) * 70000)), ('f64', img.astype(np.float64))):
            p = os.path.join(d, nm + '.fits')
            h = {k: v for k, v in out['header'].items()}
            zf.write(p, arr, h)
            with fits.open(p) as hd:
                hd.verify('exception')
                got = hd[0].data
                same = np.array_equal(np.asarray(got), arr, equal_nan=True) if arr.dtype.kind == 'f' \
                    else np.array_equal(np.asarray(got), arr)
                hsame = all((hd[0].header[k] == v) for k, v in h.items())
                rev[nm] = dict(data_identical=bool(same), header_identical=bool(hsame),
                               dtype=str(np.asarray(got).dtype.newbyteorder('=')),
                               size_multiple_of_2880=os.path.getsize(p) % 2880 == 0)
    out['astropy_reads_our_writer'] = rev
    return out


def main():
    meta = dict(astropy=astropy.__version__, numpy=np.__version__, python=sys.version.split()[0])
    with open(os.path.join(HERE, 'astropy_wcs.json'), 'w') as f:
        json.dump(dict(meta=meta, cases=wcs_vectors()), f)
    fv = fits_vectors()
    with open(os.path.join(HERE, 'astropy_fits.json'), 'w') as f:
        json.dump(dict(meta=meta, **fv), f, indent=1)
    print(json.dumps(fv['astropy_reads_our_writer'], indent=1))


if __name__ == '__main__':
    main()
