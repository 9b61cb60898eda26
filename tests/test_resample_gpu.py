"""Parity of the HIP resampler (through the C-ABI) with the oracle.

Tolerances: the kernel interpolates fp64 lattice nodes in fp32 and evaluates the
taps in fp32; pixel values agree with the fp64 oracle to 2e-5 of the local
pixel scale (rtol 2e-5 on values, atol 2e-5 x frame rms).  Validity (weight > 0)
must agree exactly except on pixels whose fractional offset sits within the
lattice error of the snap threshold next to a bad pixel (budget 1e-5).
"""
import numpy as np
import pytest

from oracle import resample as oresample
from util import assert_close_masked, pkg, synth, to_oracle_wcs

pytestmark = pytest.mark.gpu


def oracle_resample(img, wgt, mask, win, wout, kind, fscale=1.0):
    onx, ony = wout.naxis
    px, py = oresample.positions(to_oracle_wcs(wout), to_oracle_wcs(win), onx, ony)
    return oresample.resample(img, wgt, px, py, kind, fscale, mask)


def check(engine, img, wgt, mask, win, wout, kernel='LANCZOS3', fscale=1.0,
          max_flip=1e-5):
    kind = {'LANCZOS3': oresample.LANCZOS3, 'BILINEAR': oresample.BILINEAR,
            'NEAREST': oresample.NEAREST}[kernel]
    g_img, g_wgt, g_msk = engine.resample(img, win, wout, wgt=wgt, mask=mask,
                                          kernel=kernel, fscale=fscale)
    r_img, r_wgt, r_msk = oracle_resample(img, wgt, mask, win, wout, kind, fscale)
    gv, rv = g_wgt > 0, r_wgt > 0
    flips = (gv != rv).mean()
    assert flips <= max_flip, f'validity differs on {flips:.2e} of the pixels'
    both = gv & rv
    scale = float(np.std(img)) * abs(fscale)
    vb = max_flip if kernel == 'NEAREST' else 0.0
    assert_close_masked(g_img[both], r_img[both], 2e-5, 2e-5 * scale, 'values', vb)
    assert_close_masked(g_wgt[both], r_wgt[both], 5e-5, 0.0, 'weights', vb)
    assert np.all(g_img[~gv] == 0)
    if mask is not None:
        mm = (g_msk != r_msk).mean()
        assert mm <= max_flip, f'mask differs on {mm:.2e} of the pixels'
    return g_img, g_wgt, g_msk


def test_identity_is_exact_up_to_the_border(engine):
    """Delta kernels have one non-zero tap per axis: an identity alignment keeps every pixel,
    border included (the 6 x 6 footprint rule only applies to axes with six live taps)."""
    s = synth()
    f = s.make_frame(200, 150, 7, s.tan_wcs(200, 150), nbad=40)
    g_img, g_wgt, g_msk = engine.resample(f['img'], f['wcs'], f['wcs'],
                                          wgt=f['wgt'], mask=f['mask'])
    good = f['wgt'] > 0
    assert np.array_equal(g_img[good], f['img'][good])
    assert np.array_equal(g_wgt > 0, good)              # delta taps: bad stays a single pixel
    assert np.array_equal(g_msk, f['mask'])
    assert g_wgt[0, 0] > 0 and g_wgt[-1, -1] > 0


def test_half_pixel_shift_along_one_axis_keeps_the_border_of_the_other(engine):
    """dx = 0.5, dy = 0: six live taps along x (2 / 3 columns lost), a delta along y (no row lost)."""
    s = synth()
    f = s.make_frame(160, 120, 9, s.tan_wcs(160, 120), nbad=0)
    wout = s.tan_wcs(160, 120, dx=0.5)                  # out (x, y) = in (x - 0.5, y)
    g_img, g_wgt, _ = engine.resample(f['img'], f['wcs'], wout, wgt=f['wgt'])
    cols = np.nonzero((g_wgt > 0).any(axis=0))[0]
    assert (g_wgt > 0)[:, cols[0]:cols[-1] + 1].all()   # every row of the covered columns
    assert cols[0] == 3 and cols[-1] == 160 - 3         # floor(x - 0.5) - 2 >= 0, floor(x - 0.5) + 3 <= 159


def test_integer_shift_is_exact(engine):
    s = synth()
    f = s.make_frame(160, 120, 8, s.tan_wcs(160, 120))
    wout = s.tan_wcs(160, 120, dx=-7.0, dy=4.0)   # out pixel (x, y) = in pixel (x + 7, y - 4)
    g_img, g_wgt, _ = engine.resample(f['img'], f['wcs'], wout)
    ys, xs = np.nonzero(g_wgt > 0)
    assert ys.size > 10000
    assert np.array_equal(g_img[ys, xs], f['img'][ys - 4, xs + 7])


def test_constant_image_stays_constant(engine):
    s = synth()
    img = np.full((300, 280), 137.25, dtype=np.float32)
    win = s.ztf_wcs(280, 300, tpv=True)
    wout = s.ztf_wcs(280, 300, dx=3.37, dy=-2.81, rot_deg=0.4, tpv=True)
    g_img, g_wgt, _ = engine.resample(img, win, wout)
    ok = g_wgt > 0
    assert ok.mean() > 0.9
    np.testing.assert_allclose(g_img[ok], 137.25, rtol=3e-6)
    np.testing.assert_allclose(g_wgt[ok], 1.0, rtol=3e-6)


@pytest.mark.parametrize('kernel', ['LANCZOS3', 'BILINEAR', 'NEAREST'])
def test_tpv_dither_rotation_matches_oracle(engine, kernel):
    s = synth()
    win = s.ztf_wcs(400, 360, dx=5.3, dy=-8.7, rot_deg=0.1, tpv=True)
    wout = s.ztf_wcs(420, 380, tpv=True)
    f = s.make_frame(400, 360, 11, win, nbad=150, nstars=60)
    # nearest neighbour: a position within fp32 rounding of x.5 picks the other pixel
    check(engine, f['img'], f['wgt'], f['mask'], win, wout, kernel, fscale=0.37,
          max_flip=1e-4 if kernel == 'NEAREST' else 1e-5)


def test_large_rotation_and_scale_fall_back_correctly(engine):
    # 30 degrees and a 1.7x coarser output grid: the tile footprint no longer
    # fits the planned LDS tile everywhere, so both code paths run
    s = synth()
    win = s.ztf_wcs(300, 300, rot_deg=30.0, tpv=False)
    wout = s.tan_wcs(220, 200, scale=1.7 * 2.8125e-4)
    f = s.make_frame(300, 300, 12, win, nbad=100)
    check(engine, f['img'], f['wgt'], f['mask'], win, wout, 'LANCZOS3')


def test_disjoint_frames_give_no_data(engine):
    s = synth()
    win = s.tan_wcs(128, 128)
    wout = s.tan_wcs(128, 128, crval=(200.0, -10.0))
    img = np.ones((128, 128), dtype=np.float32)
    g_img, g_wgt, g_msk = engine.resample(img, win, wout, mask=np.ones((128, 128), np.int32))
    assert not g_img.any() and not g_wgt.any() and not g_msk.any()


def test_ragged_sizes(engine):
    # odd widths, sizes that are not multiples of the 64 x 16 tile or the lattice
    s = synth()
    for (nx, ny, onx, ony) in [(33, 17, 31, 19), (65, 129, 67, 15), (1, 1, 5, 5), (7, 7, 1, 1)]:
        win = s.tan_wcs(nx, ny, dx=0.3, dy=-0.2)
        wout = s.tan_wcs(onx, ony)
        rng = np.random.default_rng(nx * 1000 + ny)
        img = rng.normal(100, 5, (ny, nx)).astype(np.float32)
        msk = (rng.uniform(size=(ny, nx)) < 0.05).astype(np.int32) * 4
        check(engine, img, None, msk, win, wout, 'LANCZOS3', max_flip=0.0)


def test_flux_is_conserved_for_a_star(engine):
    s = synth()
    win = s.tan_wcs(129, 129)
    img = np.zeros((129, 129))
    s.add_stars(img, [64.3], [63.6], [5e4], 2.2)
    # pure sub-pixel shift: unit-sum taps conserve the flux exactly
    wout = s.tan_wcs(129, 129, dx=0.37, dy=-0.41)
    g_img, g_wgt, _ = engine.resample(img.astype(np.float32), win, wout)
    assert abs(g_img.sum() / img.sum() - 1.0) < 1e-5
    # 5 % coarser output pixels: with the fixed area ratio the flux of a
    # well-sampled star (FWHM 3.5 px) is kept to the accuracy of the Lanczos-3
    # interpolant itself (0.3 % of the peak per axis; the oracle agrees)
    img = np.zeros((129, 129))
    s.add_stars(img, [64.3], [63.6], [5e4], 3.5)
    wout = s.tan_wcs(129, 129, dx=0.37, dy=-0.41, scale=2.81e-4 * 1.05)
    fs = engine.flux_scale(win, wout)
    assert abs(fs - 1.05 ** 2) < 1e-6
    g_img, g_wgt, _ = engine.resample(img.astype(np.float32), win, wout, fscale=fs)
    assert abs(g_img.sum() / img.sum() - 1.0) < 1e-2
    r_img = oracle_resample(img, None, None, win, wout, oresample.LANCZOS3, fs)[0]
    assert abs(g_img.sum() / r_img.sum() - 1.0) < 1e-5


def test_wrong_shape_raises(engine):
    s = synth()
    with pytest.raises(ValueError):
        engine.resample(np.zeros((10, 12), np.float32), s.tan_wcs(10, 12), s.tan_wcs(5, 5))
    z = pkg()
    with pytest.raises(z.ZMError):
        engine.resample(np.zeros((12, 10), np.float32), s.tan_wcs(10, 12),
                        s.tan_wcs(5, 5), kernel=7)
