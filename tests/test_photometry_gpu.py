"""Forced aperture photometry against the oracle."""
import os

import numpy as np
import pytest

from oracle import photometry as ophot
from util import pkg, synth

pytestmark = pytest.mark.gpu


def test_matches_oracle_including_edges(engine):
    rng = np.random.default_rng(1)
    ny, nx = 120, 150
    img = rng.normal(5, 2, (ny, nx)).astype(np.float32)
    rms = rng.uniform(1, 3, (ny, nx)).astype(np.float32)
    mask = (rng.uniform(size=(ny, nx)) < 0.02).astype(np.int32) * rng.choice([1, 256, 2048], (ny, nx))
    x = np.concatenate([rng.uniform(-5, nx + 5, 300), [0.0, nx - 1.0, 10.5, 40.0, -50.0]])
    y = np.concatenate([rng.uniform(-5, ny + 5, 300), [0.0, ny - 1.0, 20.5, 40.0, 10.0]])
    for r in (3.0, 1.2, 7.5):
        f, e, fl = engine.aperture_photometry(img, x, y, rms=rms, mask=mask, radius=r)
        rf, re, rfl = ophot.aperture_photometry(img, rms, mask, x, y, r)
        np.testing.assert_allclose(f, rf, rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(e, re, rtol=1e-10, atol=1e-9)
        assert np.array_equal(fl, rfl)            # integer work: bit exact


def test_constant_image_gives_the_circle_area(engine):
    img = np.full((64, 64), 2.0, np.float32)
    f, e, fl = engine.aperture_photometry(img, [31.3, 20.0], [30.7, 20.5], rms=np.ones_like(img))
    np.testing.assert_allclose(f, 2.0 * np.pi * 9.0, rtol=1e-12)
    np.testing.assert_allclose(e, np.sqrt(np.pi * 9.0), rtol=1e-12)
    assert not fl.any()


def test_object_api_on_a_subtraction_like_image(engine, tmp_path):
    z = pkg()
    s = synth()
    f = s.make_frame(200, 180, 5, s.ztf_wcs(200, 180), nstars=0, sky=0.0, noise=1.0, nbad=30)
    f['header'].update({'OBSJD': 2458000.5, 'FILTER': 'ZTF_g'})
    img = np.zeros((180, 200))
    s.add_stars(img, [100.3], [90.6], [5000.0], 2.0)
    p = str(tmp_path / 'sub.fits')
    z.fits.write(p, img.astype(np.float32), f['header'])
    z.fits.write(p.replace('.fits', '.rms.fits'), np.ones((180, 200), np.float32), f['header'])
    z.fits.write(p.replace('.fits', '.mask.fits'), f['mask'], f['header'])
    ra, dec = f['wcs'].all_pix2world([100.3], [90.6], 0)
    t = z.raw_aperture_photometry(p, p.replace('.fits', '.rms.fits'), p.replace('.fits', '.mask.fits'),
                                  ra, dec)
    assert set(['flux', 'fluxerr', 'flags', 'zp', 'obsjd', 'filtercode']) <= set(t.colnames)
    # a Gaussian of FWHM 2 px: 98.6 % of the flux falls within r = 3 px
    assert abs(t['flux'][0] / 5000.0 - (1 - np.exp(-0.5 * (3 / (2 / 2.3548)) ** 2))) < 2e-3
    assert abs(t['fluxerr'][0] - np.sqrt(np.pi * 9)) < 1e-6
    assert t['zp'][0] == f['header']['MAGZP'] + f['header']['APCOR4']
    assert t['filtercode'][0] == 'zg' and t['obsjd'][0] == 2458000.5
    im = z.ScienceImage.from_file(p)
    im.mask_image = z.MaskImage.from_file(p.replace('.fits', '.mask.fits'))
    t2 = z.aperture_photometry(im, ra, dec, assume_background_subtracted=True, apply_calibration=True)
    assert t2['flux'][0] == t['flux'][0] and np.isfinite(t2['mag'][0])
