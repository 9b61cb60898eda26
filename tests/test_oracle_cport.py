"""The C / OpenMP restatement (oracle/cport) against the numpy oracle: same arithmetic,
fp64, so they agree to rounding.  The C port is what bench.py times as cpu_baseline."""
import numpy as np
import pytest

from oracle import combine as ocombine
from oracle import cport
from oracle import resample as ores
from util import synth, to_oracle_wcs


@pytest.fixture(scope='module')
def c():
    return cport.load()


def test_positions_match_numpy(c):
    s = synth()
    for tpv in (True, False):
        wout = to_oracle_wcs(s.ztf_wcs(150, 130, tpv=tpv))
        win = to_oracle_wcs(s.ztf_wcs(160, 120, dx=3.3, dy=-2.2, rot_deg=0.3, tpv=tpv))
        px, py = c.positions(wout, win, 150, 130)
        rx, ry = ores.positions(wout, win, 150, 130)
        assert np.abs(px - rx).max() < 1e-9 and np.abs(py - ry).max() < 1e-9


@pytest.mark.parametrize('kind', [ores.LANCZOS3, ores.BILINEAR])
def test_resample_matches_numpy(c, kind):
    s = synth()
    win = s.ztf_wcs(140, 110, tpv=True)
    wout = s.ztf_wcs(150, 120, dx=2.37, dy=-1.61, rot_deg=0.35, tpv=True)
    f = s.make_frame(140, 110, 11, win, nstars=15, nbad=60)
    px, py = ores.positions(to_oracle_wcs(wout), to_oracle_wcs(win), 150, 120)
    # an exactly aligned column and row exercise the delta kernels
    px[:, 7] = np.round(px[:, 7])
    py[9, :] = np.round(py[9, :])
    o, w, m = c.resample(f['img'], f['wgt'], px, py, kind, 0.37, f['mask'])
    ro, rw, rm = ores.resample(f['img'], f['wgt'], px, py, kind, 0.37, f['mask'])
    assert np.array_equal(w > 0, rw > 0)
    np.testing.assert_allclose(o, ro, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(w, rw, rtol=1e-12, atol=0)
    assert np.array_equal(m, rm)
    o2, w2, _ = c.resample(f['img'], None, px, py, kind, 1.0)
    ro2, rw2, _ = ores.resample(f['img'], None, px, py, kind, 1.0)
    np.testing.assert_allclose(o2, ro2, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(w2, rw2, rtol=1e-12)


@pytest.mark.parametrize('kind', ['WEIGHTED', 'CLIPPED', 'MEDIAN', 'AVERAGE'])
def test_combine_matches_numpy(c, kind):
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 32):
        vals = rng.normal(100, 5, (n, 20, 30))
        wgts = rng.uniform(0.01, 0.1, (n, 20, 30))
        wgts[rng.uniform(size=wgts.shape) < 0.2] = 0
        if n > 2:
            vals[1][rng.uniform(size=(20, 30)) < 0.1] += 500
        o, w = c.combine(vals, wgts, kind)
        ro, rw, _ = ocombine.combine(vals, wgts, kind)
        np.testing.assert_allclose(o, ro, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(w, rw, rtol=1e-12, atol=0)


def test_background_matches_numpy(c):
    """Mesh background / RMS maps of the C port against oracle.background (what bench.py's
    cpu_baseline times for SWarp's SUBTRACT_BACK and the variance rescale): ragged frame, a
    star field, a masked region that empties some meshes, flat weights (the variance map)."""
    from oracle import background as obk
    s = synth()
    rng = np.random.default_rng(12)
    ny, nx = 300, 410
    img = rng.normal(180.0, 5.0, (ny, nx)) + 0.02 * np.arange(nx)[None, :]
    s.add_stars(img, rng.uniform(0, nx, 40), rng.uniform(0, ny, 40), np.exp(rng.uniform(7, 10, 40)), 2.2)
    wgt = np.full((ny, nx), 1 / 25.0, np.float32)
    wgt[rng.uniform(size=wgt.shape) < 0.01] = 0
    wgt[:70, 130:260] = 0                                   # a whole mesh without pixels
    for im, wg, mesh in ((img, wgt, 64), (img, None, 128), (1.0 / np.maximum(wgt, 1e-3).astype(np.float64), wgt, 64)):
        b, r, bm, bs, nb, ns = c.background(im, wg, mesh=mesh, fsize=3)
        rb, rr, rbm, rbs, rnb, rns = obk.background(im, wg, mesh=mesh, fsize=3)
        np.testing.assert_allclose(nb, rnb, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ns, rns, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(b, rb, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r, rr, rtol=1e-9, atol=1e-9)
        assert abs(bm - rbm) < 1e-9 * max(1, abs(rbm)) and abs(bs - rbs) < 1e-9 * max(1, abs(rbs))
