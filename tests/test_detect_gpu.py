"""Pixel-only seeing estimate and detection cuts against the oracle
and against synthetic truth."""
import ctypes as C

import numpy as np
import pytest

from oracle import detect as odet
from util import pkg, synth

pytestmark = pytest.mark.gpu


def star_field(fwhm, seed=1, nx=400, ny=360, nstars=60, noise=3.0):
    s = synth()
    rng = np.random.default_rng(seed)
    img = np.zeros((ny, nx))
    xs, ys = rng.uniform(15, nx - 15, nstars), rng.uniform(15, ny - 15, nstars)
    s.add_stars(img, xs, ys, np.exp(rng.uniform(np.log(3e3), np.log(3e5), nstars)), fwhm)
    img += rng.normal(0, noise, img.shape)
    return img.astype(np.float32)


def gpu_stars(engine, img, bad, lo, hi, iso=5, border=12, cap=4096):
    z = pkg()
    ny, nx = img.shape
    xs, ys, pk = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
    n = C.c_int(0)
    b8 = None if bad is None else bad.astype(np.uint8)
    z._lib.check(engine.L.zm_find_stars(engine.ctx, z._lib.ptr(img), z._lib.ptr(b8), nx, ny, lo, hi, iso,
                                        border, cap, z._lib.ptr(xs), z._lib.ptr(ys), z._lib.ptr(pk),
                                        C.byref(n)))
    m = n.value
    order = np.lexsort((xs[:m], ys[:m], -pk[:m].astype(np.float64)))
    return xs[:m][order], ys[:m][order], pk[:m][order]


def test_star_list_is_the_oracles(engine):
    img = star_field(2.3, seed=2)
    img[100, 100] = np.nan
    bad = np.zeros(img.shape, bool)
    bad[200:220, 150:170] = True
    xs, ys, pk = gpu_stars(engine, img, bad, 40.0, 9000.0)
    ref, nref = odet.find_stars(img, bad, 40.0, 9000.0, nmax=100000)
    assert len(xs) == nref > 20
    assert [(int(a), int(b)) for a, b in zip(xs, ys)] == [(x, y) for x, y, _ in ref]
    assert np.array_equal(pk, np.array([v for _, _, v in ref], np.float32))


def test_fwhm_matches_oracle_and_truth(engine):
    z = pkg()
    for fwhm in (1.8, 2.5, 4.0):
        img = star_field(fwhm, seed=int(fwhm * 10), noise=1.0)
        xs, ys, _ = gpu_stars(engine, img, None, 200.0, 3e38)
        xs, ys = np.ascontiguousarray(xs[:40]), np.ascontiguousarray(ys[:40])
        k = len(xs)
        fw, cx, cy = np.empty(k), np.empty(k), np.empty(k)
        z._lib.check(engine.L.zm_star_fwhm(engine.ctx, z._lib.ptr(img), img.shape[1], img.shape[0], k,
                                           z._lib.ptr(xs), z._lib.ptr(ys), 10, z._lib.ptr(fw),
                                           z._lib.ptr(cx), z._lib.ptr(cy)))
        ref = np.array([odet.star_fwhm(img, int(x), int(y), 10) for x, y in zip(xs, ys)])
        np.testing.assert_allclose(fw, ref[:, 0], rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(cx, ref[:, 1], rtol=0, atol=1e-8)
        assert abs(np.nanmedian(fw) / fwhm - 1) < 0.03


def test_measure_seeing_on_a_sky_with_background(engine):
    z = pkg()
    img = star_field(2.2, seed=9, nx=700, ny=650, nstars=150, noise=4.0) + 180.0
    yy, xx = np.mgrid[0:650, 0:700]
    img = (img + 0.02 * xx).astype(np.float32)
    img[300:305, 300:305] = 60000.0                      # a saturated blob: not a star
    bad = np.zeros(img.shape, bool)
    seeing, nused = z.seeing.measure_seeing(img, bad, saturate=50000.0, engine=engine)
    assert nused > 30 and abs(seeing / 2.2 - 1) < 0.03
    with pytest.raises(RuntimeError):
        z.seeing.measure_seeing(np.random.default_rng(0).normal(100, 3, (200, 200)).astype(np.float32),
                                engine=engine)


def test_estimate_seeing_writes_the_header_like_the_reference(engine, tmp_path):
    z = pkg()
    s = synth()
    img = star_field(2.6, seed=4) + 150.0
    hdr = dict(s.ztf_wcs(400, 360).to_header(), SATURATE=50000.0, MAGZP=26.0)
    p = str(tmp_path / 'sci.fits')
    z.fits.write(p, img, hdr)
    im = z.ScienceImage.from_file(p)
    seeing = z.estimate_seeing(im)
    assert abs(seeing / 2.6 - 1) < 0.03
    again = z.fits.read(p)
    assert again[1]['SEEING'] == pytest.approx(seeing) and 'Goldstein' in again[2]['SEEING']


def test_negpix_and_pixel_cuts(engine):
    z = pkg()
    rng = np.random.default_rng(6)
    ny, nx = 300, 320
    img = rng.normal(0, 2.0, (ny, nx)).astype(np.float32)
    rms = np.full((ny, nx), 2.0, np.float32)
    bpm = np.zeros((ny, nx), bool)
    # candidates (1-based X_IMAGE, Y_IMAGE): clean, dipole, near a bad pixel, noisy region, edge
    x = np.array([50.3, 120.0, 200.5, 260.0, 3.0, 318.6])
    y = np.array([60.7, 130.5, 210.0, 40.0, 4.0, 297.2])
    img[129, 121] = -40.0
    img[129, 122] = 45.0                                  # dipole inside candidate 2's cutout
    bpm[212, 203] = True
    rms[30:52, 250:272] = 5.0
    img[2, 1] = -50.0
    img[2, 2] = 60.0                                      # dipole at the frame corner
    cuts = z.pixel_cuts(img, rms, bpm, x, y, engine=engine)
    med = float(np.median(img))
    sig = 1.48 * float(np.median(np.abs(img - med)))
    ref_neg = odet.negpix(img, x, y, med, sig)
    assert np.array_equal(cuts['NEGPIX'], ref_neg)
    assert cuts['NEGPIX'].tolist() == [0, 1, 0, 0, 1, 0]
    assert cuts['BPMCUT'][2] > 0 and cuts['BPMCUT'][[0, 1, 3]].max() == 0
    assert cuts['RMSCUT'][3] > cuts['MEDCUT'] and cuts['RMSCUT'][0] == pytest.approx(2.0, rel=1e-5)
    assert cuts['GOODCUT'].tolist() == [1, 0, 0, 0, 0, 1]
