"""Known-answer tests of the background oracle (no GPU)."""
import numpy as np
from scipy.interpolate import CubicSpline

from oracle import background as oback


def test_natural_spline_matches_scipy():
    rng = np.random.default_rng(1)
    nodes = rng.normal(100, 3, (7, 5))
    up = oback.expand(nodes, 5 * 32, 7 * 32, 32)
    # reference: natural cubic splines along y then x through the mesh centres
    yc = (np.arange(7) + 0.5) * 32 - 0.5
    xc = (np.arange(5) + 0.5) * 32 - 0.5
    rows = CubicSpline(yc, nodes, axis=0, bc_type='natural', extrapolate=True)(np.arange(7 * 32))
    ref = CubicSpline(xc, rows, axis=1, bc_type='natural', extrapolate=True)(np.arange(5 * 32))
    np.testing.assert_allclose(up, ref, rtol=0, atol=1e-9)


def test_flat_noise_field_recovers_level_and_sigma():
    rng = np.random.default_rng(2)
    img = rng.normal(150.0, 5.0, (512, 512))
    bkg, rms, mean, sig, bo, so = oback.background(img, None, 128)
    assert abs(mean - 150.0) < 0.1 and abs(sig - 5.0) < 0.1
    assert np.abs(bkg - 150.0).max() < 0.5
    assert np.abs(rms - 5.0).max() < 0.3


def test_stars_do_not_bias_the_mode():
    rng = np.random.default_rng(3)
    img = rng.normal(100.0, 4.0, (256, 256))
    img[rng.uniform(size=img.shape) < 0.05] += rng.uniform(20, 2000)   # 5 % bright pixels
    _, _, mean, sig, _, _ = oback.background(img, None, 128)
    assert abs(mean - 100.0) < 0.5 and abs(sig - 4.0) < 0.5


def test_masked_mesh_is_filled():
    rng = np.random.default_rng(4)
    img = rng.normal(50.0, 2.0, (384, 384))
    w = np.ones_like(img)
    w[128:256, 128:256] = 0
    back, sigm = oback.mesh_maps(img, w, 128)
    assert back[1, 1] == -oback.BIG
    bo, so = oback.filter_maps(back, sigm, 3)
    assert abs(bo[1, 1] - 50.0) < 0.5


def test_histogram_median_merge_path_equals_sequential_walk():
    # host-side check of the search the kernel uses in place of backguess's walk
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(1, 200))
        h = rng.integers(0, 6, n) * (rng.uniform(size=n) < 0.7)
        lcut = int(rng.integers(0, n))
        hcut = int(rng.integers(lcut, n))
        ref = oback.histogram_median_walk(h, lcut, hcut)
        P = np.concatenate([[0], np.cumsum(h)])   # P[i+1] = inclusive prefix at i
        p0 = lambda i: 0 if i < 0 else int(P[i + 1])
        T = hcut - lcut + 1
        lo, hi = 0, T
        while lo < hi:
            a = (lo + hi + 1) >> 1
            La = p0(lcut + a - 2) - p0(lcut - 1)
            Hb = p0(hcut) - p0(hcut - (T - a))
            if La < Hb:
                lo = a
            else:
                hi = a - 1
        a, b = lo, T - lo
        lowsum = p0(lcut + a - 1) - p0(lcut - 1)
        highsum = p0(hcut) - p0(hcut - b)
        ihigh, ilow = hcut - b, lcut + a
        if ihigh >= 0:
            den = 2.0 * max(int(h[ilow]), int(h[ihigh]))
            med = ihigh + 0.5 + ((highsum - lowsum) / den if den > 0 else 0.0)
        else:
            med = 0.0
        assert med == ref, (trial, med, ref)
