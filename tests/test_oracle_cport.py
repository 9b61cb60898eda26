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


# ---- the subtraction leg (oracle/cport/zm_hotpants.c) against oracle/hotpants.py ------------------------------------
# The C restatement builds the basis vectors with the separable passes the basis allows (as hotpants' xy_conv_stamp and
# csrc/hp_vectors.hip do) where the numpy oracle correlates 49 two-dimensional kernels: the same definition, sums that
# differ in rounding.  Discrete outcomes (stamps, rounds, fill pattern, counts) must be identical; pixels agree to the
# conditioning of the normal equations (Jacobi scaling + 1e-10 ridge).
def _hp_scene(nx=384, ny=352, seed=1, nstars=120, ksig=0.9, scale=1.3, bg=20.0, gradient=0.0, nbad=6):
    from scipy.ndimage import gaussian_filter
    s = synth()
    rng = np.random.default_rng(seed)
    ref = np.full((ny, nx), 150.0)
    s.add_stars(ref, rng.uniform(10, nx - 10, nstars), rng.uniform(10, ny - 10, nstars),
                np.exp(rng.uniform(np.log(3e3), np.log(8e4), nstars)), 2.0)
    yy, xx = np.mgrid[0:ny, 0:nx]
    if gradient:
        a = gaussian_filter(ref, ksig * (1 - gradient / 2), mode='nearest')
        b = gaussian_filter(ref, ksig * (1 + gradient / 2), mode='nearest')
        t = xx / (nx - 1.0)
        sci = scale * ((1 - t) * a + t * b) + bg
    else:
        sci = scale * gaussian_filter(ref, ksig, mode='nearest') + bg
    ref = ref + rng.normal(0, 0.5, ref.shape)
    sci = sci + rng.normal(0, 3.0, sci.shape)
    bpm = np.zeros((ny, nx), np.uint8)
    for _ in range(nbad):
        bx, by = rng.integers(20, nx - 20), rng.integers(20, ny - 20)
        bpm[by:by + 3, bx:bx + 3] = 1
    return (sci.astype(np.float32), ref.astype(np.float32), np.full((ny, nx), 3.0, np.float32),
            np.full((ny, nx), 0.5, np.float32), bpm)


@pytest.mark.parametrize('case', [
    dict(sc=dict(), kw=dict(r=5.0, rss=12.0, nsx=4, nsy=4, nrx=2, nry=1, ko=1, bgo=1), tol=1e-7),
    dict(sc=dict(nx=330, ny=300, seed=7, gradient=0.3), kw=dict(r=4.0, rss=9.0, nsx=3, nsy=3, nrx=3, nry=3, ko=1, bgo=0), tol=1e-6),
    dict(sc=dict(nx=448, ny=416, seed=5, nstars=220, gradient=0.2), kw=dict(r=6.0, rss=11.0, nsx=5, nsy=5, ko=2, bgo=1), tol=1e-7),
    dict(sc=dict(seed=3), kw=dict(r=4.0, rss=8.0, nsx=6, nsy=6, ko=4, bgo=0, normalize=1), tol=5e-6),   # the reference's orders: 722 unknowns
])
def test_hotpants_port_matches_numpy(c, case):
    from oracle import hotpants as ohp
    data = _hp_scene(**case['sc'])
    kw = dict(case['kw'], tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
    d0, n0, i0 = ohp.subtract(*data, **kw)
    d1, n1, i1 = c.hotpants(*data, **kw)
    assert np.array_equal(d0 == 1e-30, d1 == 1e-30) and i0['nmasked'] == i1['nmasked']
    assert np.array_equal(n0 == kw.get('fin', ohp.DEFAULTS['fin']), n1 == kw.get('fin', ohp.DEFAULTS['fin']))
    good = d0 != 1e-30
    assert good.mean() > 0.5
    assert np.abs(d0 - d1)[good].max() <= case['tol'] * np.abs(d0[good]).max()
    np.testing.assert_allclose(n1[good], n0[good], rtol=1e-8)
    assert len(i0['regions']) == len(i1['regions'])
    for a, b in zip(i0['regions'], i1['regions']):
        assert (a is None) == (b is None)
        if a is not None:
            for k in ('nstamps_total', 'nstamps_used', 'niter', 'ncoeff'):
                assert a[k] == b[k], k
            assert b['kernel_sum'] == pytest.approx(a['kernel_sum'], rel=1e-8)
            assert b['chi2'] == pytest.approx(a['chi2'], rel=1e-6)


def test_hotpants_port_one_region_and_an_unsolvable_one(c):
    from oracle import hotpants as ohp
    data = list(_hp_scene(nx=330, ny=300, seed=7, gradient=0.3))
    kw = dict(r=4.0, rss=9.0, nsx=3, nsy=3, nrx=3, nry=3, ko=1, bgo=0, tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
    full = c.hotpants(*data, **kw)
    one = c.hotpants(*data, only_region=4, **kw)
    x0, x1, y0, y1 = ohp.regions(330, 300, 3, 3)[4]
    assert np.array_equal(one[0][y0:y1, x0:x1], full[0][y0:y1, x0:x1]) and one[2]['regions'][4] == full[2]['regions'][4]
    assert (one[0][:y0] == 1e-30).all() and [r is None for r in one[2]['regions']].count(False) == 1
    # a region without a usable stamp (everything masked there): None in both, fill values under it
    bpm = data[4].copy()
    bpm[:100, :110] = 1
    d0, _, i0 = ohp.subtract(data[0], data[1], data[2], data[3], bpm, **kw)
    d1, _, i1 = c.hotpants(data[0], data[1], data[2], data[3], bpm, **kw)
    assert i0['regions'][0] is None and i1['regions'][0] is None and np.array_equal(d0 == 1e-30, d1 == 1e-30)
