"""Analytic known answers for the pixel-only seeing estimate / detection cuts of the oracle
(oracle/detect.py); the GPU versions are compared with it in tests/test_detect_gpu.py."""
import numpy as np

from oracle import detect as odet
from util import synth


def test_adaptive_moments_recover_a_gaussian():
    # sums over pixels instead of integrals: exact for a well sampled star, 2 % low at
    # sigma = 0.7 px (FWHM 1.7), where the pixel grid undersamples the profile
    s = synth()
    for fwhm, (x, y), tol, ctol in ((1.7, (30.3, 28.8), 2e-2, 5e-2), (2.5, (31.0, 30.0), 1e-3, 1e-6),
                                    (4.2, (29.6, 31.4), 1e-6, 1e-3)):
        img = np.zeros((61, 61))
        s.add_stars(img, [x], [y], [1e5], fwhm)
        f, cx, cy = odet.star_fwhm(img, int(round(x)), int(round(y)), half=14)
        assert abs(f / fwhm - 1) < tol and abs(cx - x) < ctol and abs(cy - y) < ctol


def test_star_finder_rules():
    img = np.zeros((60, 60), np.float32)
    img[20, 20] = 100          # isolated star
    img[40, 40] = img[40, 41] = 80      # tie: raster-first pixel wins
    img[30, 30] = 5000         # "saturated": above hi
    img[5, 5] = 100            # inside the border
    img[45, 20] = 100
    bad = np.zeros(img.shape, bool)
    bad[47, 22] = True         # a bad pixel in the box of (20, 45)
    stars, n = odet.find_stars(img, bad, 10.0, 1000.0, iso=5, border=12)
    assert n == 2 and [(x, y) for x, y, _ in stars] == [(20, 20), (40, 40)]


def test_negative_pixel_cut_semantics():
    img = np.zeros((40, 40), np.float32)
    img[10, 10], img[10, 11] = -10, 10          # dipole
    img[25, 25] = -10                           # lone negative pixel: fine
    img[30, 30], img[30, 32] = -10, 10          # positive pixel two columns away: fine
    x = np.array([11.0, 26.0, 31.0, 18.0])      # 1-based X_IMAGE
    y = np.array([11.0, 26.0, 31.0, 11.0])
    out = odet.negpix(img, x, y, 0.0, 1.0)
    assert out.tolist() == [1, 0, 0, 0]          # the 4th cutout (cols 12..22) misses the dipole
