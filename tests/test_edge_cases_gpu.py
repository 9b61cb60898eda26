"""Degenerate inputs through the C-ABI: frames off the grid, frames smaller than the kernel
footprint, fully masked frames, single-frame stacks, ragged tiny sizes.  Expected values come
from the oracle; nothing may crash and 'no data' must come out as weight 0 / mask coverage 0."""
import numpy as np
import pytest

from oracle import combine as ocombine
from oracle import resample as ores
from util import assert_close_masked, pkg, synth, to_oracle_wcs

pytestmark = pytest.mark.gpu


def oracle_resample(f, wout, kind=ores.LANCZOS3):
    onx, ony = wout.naxis
    px, py = ores.positions(to_oracle_wcs(wout), to_oracle_wcs(f['wcs']), onx, ony)
    return ores.resample(f['img'], f['wgt'], px, py, kind, f['flxscale'], f['mask'])


def test_frame_entirely_off_the_output_grid(engine):
    z = pkg()
    s = synth()
    base = s.tan_wcs(120, 100)
    far = s.make_frame(120, 100, 1, s.tan_wcs(120, 100, dx=5000.0, dy=-4000.0), nstars=5)
    near = s.make_frame(120, 100, 2, s.tan_wcs(120, 100, dx=1.5, dy=0.5), nstars=5)
    o, w, m = engine.resample(far['img'], far['wcs'], base, wgt=far['wgt'], mask=far['mask'])
    assert not w.any() and not o.any() and not m.any()
    p = z.coadd_params(combine='CLIPPED', subtract_back=False, rescale_weights=False)
    img, wgt, msk, mw = engine.coadd([far, near], base, p)
    ro, rw, rm = oracle_resample(near, base)
    assert np.array_equal(wgt > 0, rw > 0)
    assert_close_masked(img[rw > 0], ro[rw > 0], 2e-5, 1e-3, 'single covering frame')
    # only off-grid frames: a valid, empty coadd
    img, wgt, msk, mw = engine.coadd([far, far], base, p)
    assert not wgt.any() and not img.any() and not msk.any() and not mw.any()


@pytest.mark.parametrize('shape', [(5, 5), (6, 6), (7, 9), (1, 40), (40, 3)])
def test_frames_around_the_size_of_the_kernel_footprint(engine, shape):
    s = synth()
    nx, ny = shape
    rng = np.random.default_rng(nx * 100 + ny)
    w_in = s.tan_wcs(nx, ny)
    f = dict(img=rng.normal(10, 1, (ny, nx)).astype(np.float32),
             wgt=np.full((ny, nx), 0.5, np.float32), mask=np.full((ny, nx), 4, np.int32), wcs=w_in,
             flxscale=1.0)
    base = s.tan_wcs(24, 20, dx=(24 - nx) / 2.0 + 0.3, dy=(20 - ny) / 2.0 - 0.2)   # frame near the centre
    for kernel, kind in (('LANCZOS3', ores.LANCZOS3), ('BILINEAR', ores.BILINEAR), ('NEAREST', ores.NEAREST)):
        o, w, m = engine.resample(f['img'], w_in, base, wgt=f['wgt'], mask=f['mask'], kernel=kernel)
        ro, rw, rm = oracle_resample(f, base, kind)
        assert np.array_equal(w > 0, rw > 0), kernel
        assert np.array_equal(m, rm), kernel
        if (rw > 0).any():
            assert_close_masked(o[rw > 0], ro[rw > 0], 2e-5, 1e-4, kernel)
        if kernel == 'LANCZOS3' and (nx < 6 or ny < 6):
            assert not w.any()                     # no 6 x 6 footprint fits


def test_fully_masked_and_single_frame_stacks(engine):
    z = pkg()
    s = synth()
    base = s.tan_wcs(90, 70)
    good = s.make_frame(90, 70, 3, s.tan_wcs(90, 70, dx=0.4, dy=-0.6), nstars=8, nbad=20)
    dead = dict(good, wgt=np.zeros_like(good['wgt']), mask=np.full_like(good['mask'], 256))
    for kind in ('CLIPPED', 'MEDIAN', 'WEIGHTED'):
        p = z.coadd_params(combine=kind, subtract_back=True, rescale_weights=True, back_size=32)
        # a dead frame adds nothing (its background statistics find no valid mesh)
        a = engine.coadd([good], base, p)
        b = engine.coadd([good, dead], base, p)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), kind
        # AND of {m, 256-everywhere} over the pixels both cover
        both = (b[3] > 0)
        assert np.array_equal(b[2][both], (a[2] & 256)[both])
        c = engine.coadd([dead], base, p)
        assert not c[1].any() and not c[0].any()


def test_background_of_tiny_and_odd_frames(engine):
    from oracle import background as oback
    rng = np.random.default_rng(4)
    for ny, nx, mesh in ((9, 13, 8), (33, 17, 16), (130, 64, 64), (64, 1, 32)):
        img = (rng.normal(50, 2, (ny, nx)) + 0.1 * np.arange(nx)[None, :]).astype(np.float32)
        bkg, rms, sub, stats = engine.background(img, None, mesh=mesh)
        rb, rr, rmean, rsig, _, _ = oback.background(img.astype(np.float64), None, mesh)
        assert_close_masked(bkg, rb, 2e-5, 1e-3, f'bkg {ny}x{nx}')
        assert_close_masked(rms, rr, 1e-4, 1e-4, f'rms {ny}x{nx}')


def test_median_and_subtract_reject_empty_inputs(engine):
    z = pkg()
    with pytest.raises(z.ZMError):
        engine.median_mad(np.zeros((0,), np.float32))
    with pytest.raises(ValueError):
        engine.subtract(np.zeros((8, 8), np.float32), np.ones((8, 8), np.float32),
                        np.zeros((9, 8), np.float32), np.ones((8, 8), np.float32))
    # a featureless frame is a degenerate fit (the ridge keeps it solvable): same stamps and
    # fill pattern as the oracle, no crash
    from oracle import hotpants as ohp
    flat = np.full((96, 96), 100.0, np.float32)
    kw = dict(r=3.0, rss=6.0, nsx=3, nsy=3, ko=0, bgo=0)
    d, n, info = engine.subtract(flat, np.ones_like(flat), flat, np.ones_like(flat), None, **kw)
    rd, rn, rinfo = ohp.subtract(flat, flat, np.ones_like(flat), np.ones_like(flat), None, **kw)
    assert info['nstamps_used'] == rinfo['regions'][0]['nstamps_used']
    assert np.array_equal(d == np.float32(1e-30), rd == 1e-30)
    # every pixel above the upper threshold: no stamp at all -> status != 0, everything filled
    d, n, info = engine.subtract(flat, np.ones_like(flat), flat, np.ones_like(flat), None, tu=50.0, iu=50.0,
                                 **kw)
    assert info['status'] != 0 and info['nstamps_used'] == 0
    assert np.all(d == np.float32(1e-30)) and np.all(n == np.float32(np.sqrt(50000.0)))
