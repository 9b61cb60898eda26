"""Size-independent properties at BASELINE.json's full frame size (3072 x 3072 and
the real ZTF quadrant 3072 x 3080), where the oracle is too slow to run: exact
identities, linearity, idempotence, round trips and fill patterns."""
import numpy as np
import pytest
from scipy.ndimage import maximum_filter

from util import pkg, synth

pytestmark = pytest.mark.gpu

NX, NY = 3072, 3080


@pytest.fixture(scope='module')
def frame():
    s = synth()
    rng = np.random.default_rng(42)
    w = s.ztf_wcs(NX, NY, tpv=True)
    img = rng.normal(200.0, 6.0, (NY, NX)).astype(np.float32)
    xs, ys = rng.uniform(20, NX - 20, 1500), rng.uniform(20, NY - 20, 1500)
    fl = np.exp(rng.uniform(np.log(2e3), np.log(4e4), 1500))
    s.add_stars(img, xs, ys, fl, 2.1)
    mask = np.zeros((NY, NX), np.int32)
    bad = rng.integers(0, NX * NY, 9000)
    mask.ravel()[bad] = rng.choice([1, 256, 2, 2048], bad.size)
    wgt = np.where((mask & 198589) > 0, 0.0, 1.0 / 36.0).astype(np.float32)
    return dict(img=img, wgt=wgt, mask=mask, wcs=w, flxscale=1.0, stars=(xs, ys, fl))


def test_identity_resample_is_bit_exact(engine, frame):
    o, ow, om = engine.resample(frame['img'], frame['wcs'], frame['wcs'], wgt=frame['wgt'],
                                mask=frame['mask'])
    good = frame['wgt'] > 0
    assert np.array_equal(o[good], frame['img'][good])
    assert np.array_equal(ow > 0, good)                 # delta kernels: the border is kept
    assert np.array_equal(om, frame['mask'])


def test_integer_shift_is_bit_exact_through_tpv(engine, frame):
    s = synth()
    wout = s.ztf_wcs(NX, NY, dx=-11.0, dy=6.0, tpv=True)     # out (x, y) = in (x + 11, y - 6)
    o, ow, _ = engine.resample(frame['img'], frame['wcs'], wout)
    ys, xs = np.nonzero(ow > 0)
    assert ys.size > 9_000_000
    assert np.array_equal(o[ys, xs], frame['img'][ys - 6, xs + 11])


def test_mask_resample_is_the_or_under_the_footprint(engine, frame):
    """A sub-pixel translation: every output mask value is the OR of the 6 x 6 input pixels under
    the Lanczos-3 footprint - also where bits above 15 make the 16-bit box-OR plane defer to the
    raw mask (a block with bit 16, a block with every low bit and bit 17)."""
    s = synth()
    mask = frame['mask'].copy()
    mask[1000:1100, 500:600] |= 1 << 16
    mask[2000:2010, 2000:2010] |= (1 << 17) | 0xffff
    mask[300:303, 700:703] = 0xffff                      # all sixteen low bits, no high bit
    wout = s.ztf_wcs(NX, NY, dx=-11.4, dy=6.3, tpv=True)   # out (x, y) = in (x + 11.4, y - 6.3)
    _, ow, om = engine.resample(frame['img'], frame['wcs'], wout, mask=mask)
    want = np.zeros_like(mask)
    ys, xs = np.arange(NY)[:, None], np.arange(NX)[None, :]
    inside = np.ones((NY, NX), bool)
    for r in range(-9, -3):                              # rows floor(y - 6.3) - 2 .. + 3
        for c in range(9, 15):                           # columns floor(x + 11.4) - 2 .. + 3
            yy, xx = ys + r, xs + c
            ok = (yy >= 0) & (yy < NY) & (xx >= 0) & (xx < NX)
            inside &= ok
            want |= np.where(ok, mask[np.clip(yy, 0, NY - 1), np.clip(xx, 0, NX - 1)], 0)
    assert np.array_equal(ow > 0, inside)
    assert np.array_equal(om[inside], want[inside])
    assert (om[inside] >> 16).any() and (om[inside] == 0xffff).any()


def test_resample_is_linear(engine, frame):
    s = synth()
    wout = s.ztf_wcs(NX, NY, dx=7.3, dy=-4.6, rot_deg=0.08, tpv=True)
    rng = np.random.default_rng(3)
    other = rng.normal(0, 50, (NY, NX)).astype(np.float32)
    ra, wa, _ = engine.resample(frame['img'], frame['wcs'], wout)
    rb, _, _ = engine.resample(other, frame['wcs'], wout)
    rc, _, _ = engine.resample((2.0 * frame['img'] - 0.5 * other).astype(np.float32), frame['wcs'], wout)
    ok = wa > 0
    err = np.abs(rc - (2.0 * ra - 0.5 * rb))[ok]
    assert err.max() < 2e-5 * (np.abs(ra[ok]).max() + 200.0)
    # constant in, constant out (unit-sum taps)
    one, _, _ = engine.resample(np.ones((NY, NX), np.float32), frame['wcs'], wout)
    np.testing.assert_allclose(one[ok], 1.0, rtol=3e-6)


def test_coadd_of_identical_frames_is_the_frame(engine, frame):
    z = pkg()
    p = z.coadd_params(combine='CLIPPED', subtract_back=False, rescale_weights=False)
    frames = [frame] * 6
    o, ow, om, omw = engine.coadd(frames, frame['wcs'], p)
    inner = (slice(2, -3), slice(2, -3))
    good = frame['wgt'][inner] > 0
    np.testing.assert_allclose(o[inner][good], frame['img'][inner][good], rtol=3e-6)
    np.testing.assert_allclose(ow[inner][good], 6 * frame['wgt'][inner][good], rtol=3e-6)
    assert np.array_equal(om[inner], frame['mask'][inner])       # AND of identical masks
    assert (omw[inner] == 1).all()


def test_clipped_coadd_is_immune_to_single_frame_outliers(engine, frame):
    z = pkg()
    s = synth()
    rng = np.random.default_rng(8)
    frames, dirty = [], []
    for i in range(5):
        w = s.ztf_wcs(NX, NY, dx=rng.uniform(-8, 8), dy=rng.uniform(-8, 8), rot_deg=rng.uniform(-0.05, 0.05))
        f = dict(frame, wcs=w)
        frames.append(f)
    hit = dict(frames[2])
    im = hit['img'].copy()
    yy, xx = rng.integers(50, NY - 50, 400), rng.integers(50, NX - 50, 400)
    im[yy, xx] += 30000.0                      # 400 cosmic-ray like hits in one frame
    hit['img'] = im
    dirty = frames[:2] + [hit] + frames[3:]
    p = z.coadd_params(combine='CLIPPED', subtract_back=False, rescale_weights=False)
    clean = engine.coadd(frames, frame['wcs'], p, want_mask=False)[0]
    got = engine.coadd(dirty, frame['wcs'], p, want_mask=False)[0]
    # the five frames show different sky (same pixels through different WCS), so the
    # outlier is 30000 over a spread of a few sigma: always clipped
    assert np.abs(got - clean).max() < 60.0
    pw = z.coadd_params(combine='WEIGHTED', subtract_back=False, rescale_weights=False)
    assert np.abs(engine.coadd(dirty, frame['wcs'], pw, want_mask=False)[0] - clean).max() > 1000.0


def test_background_plus_residual_is_the_image(engine, frame):
    bkg, rms, sub, stats = engine.background(frame['img'], frame['wgt'], mesh=128)
    np.testing.assert_allclose(bkg + sub, frame['img'], rtol=2e-7, atol=2e-4)   # fp32 round trip
    assert abs(stats[0] - 200.0) < 0.3 and abs(stats[1] - 6.0) < 0.2
    assert np.abs(bkg - 200.0).max() < 2.5 and np.abs(rms - 6.0).max() < 1.0


def test_median_mad_is_exact_on_nine_megapixels(engine, frame):
    med, mad = engine.median_mad(frame['img'], frame['mask'])
    pix = frame['img'][frame['mask'] == 0]
    rmed = np.median(pix)
    assert med == float(rmed)
    assert abs(mad - 1.4826 * np.median(np.abs(pix - rmed))) < 1e-6


def test_subtracting_a_convolved_scaled_copy_leaves_nothing(engine, frame):
    # sci = 1.7 (ref (x) Gaussian) + 25, both noise free: the kernel basis can represent
    # it, so the residual is small everywhere, the kernel sum is the flux ratio and the
    # fill pattern is the bad-pixel map grown by the kernel half width
    from scipy.ndimage import gaussian_filter
    s = synth()
    xs, ys, fl = frame['stars']
    ref = np.full((NY, NX), 200.0)
    s.add_stars(ref, xs, ys, fl, 2.1)
    sci = (1.7 * gaussian_filter(ref, 0.9, mode='nearest') + 25.0).astype(np.float32)
    ref = ref.astype(np.float32)
    bpm = ((frame['mask'] & 198589) > 0).astype(np.uint8)
    rms = np.full((NY, NX), 6.0, np.float32)
    kw = dict(r=5.0, rss=12.0, nsx=10, nsy=10, nrx=3, nry=3, ko=2, bgo=0, tu=1e6, iu=1e6,
              tl=-1e3, il=-1e3)
    d, n, info = engine.subtract(sci, rms, ref, rms, bpm, **kw)
    assert info['status'] == 0 and info['nstamps_used'] > 100
    assert abs(info['kernel_sum'] - 1.7) < 1e-3
    hw = 5
    grown = maximum_filter(bpm, size=2 * hw + 1, mode='constant', cval=0).astype(bool)
    grown[:hw] = grown[-hw:] = True
    grown[:, :hw] = grown[:, -hw:] = True
    assert np.array_equal(d == np.float32(1e-30), grown)
    good = ~grown
    assert np.abs(d[good]).max() < 5e-3 * np.abs(sci).max()
    assert np.abs(d[good]).mean() < 0.05
    # noise = sqrt(6^2 + 6^2 sum K^2), K = 1.7 x Gaussian(sigma 0.9): sum K^2 = 1.7^2 / (4 pi 0.81)
    expect = 6.0 * np.sqrt(1 + 1.7 ** 2 / (4 * np.pi * 0.81))
    assert abs(np.median(n[good]) - expect) < 0.15
    assert np.all(n[grown] == np.float32(np.sqrt(50000.0)))


def blob_bpm(seed, n=300):
    """Detector defects as bench.py draws them: n clustered 3 x 3 blobs (a 49 x 49 substamp and
    its 21 x 21 kernel margin must be clean to be usable)."""
    rng = np.random.default_rng(seed)
    bpm = np.zeros((NY, NX), np.uint8)
    for bx, by in zip(rng.integers(2, NX - 2, n), rng.integers(2, NY - 2, n)):
        bpm[by - 1:by + 2, bx - 1:bx + 2] = 1
    return bpm


def test_config2_subtraction_at_the_reference_parameters(engine, frame):
    """BASELINE config[2] as prepare_hotpants emits it for SEEING = 4 px
    (zuds/hotpants.py:44-93): r = 10, rss = 24, 3 x 3 regions of 10 x 10 stamps, ko = 4,
    bgo = 0 -> nine 722-unknown systems and the 21 x 21 convolution, on a real ZTF quadrant
    size.  sci = 1.7 (ref (x) G(0.9)) + 25, noise free: kernel sum = flux ratio, residual ~ 0,
    fill pattern = bad-pixel map grown by the kernel half width, noise map analytic; the
    ko = 2 fit (SURVEY 8(d) primary) of the same constant kernel gives the same difference."""
    from scipy.ndimage import gaussian_filter
    s = synth()
    xs, ys, fl = frame['stars']
    ref = np.full((NY, NX), 200.0)
    s.add_stars(ref, xs, ys, fl, 2.1)
    sci = (1.7 * gaussian_filter(ref, 0.9, mode='nearest') + 25.0).astype(np.float32)
    ref = ref.astype(np.float32)
    bpm = blob_bpm(77)
    rms = np.full((NY, NX), 6.0, np.float32)
    kw = dict(r=10.0, rss=24.0, nsx=10, nsy=10, nrx=3, nry=3, bgo=0, tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
    d, n, info = engine.subtract(sci, rms, ref, rms, bpm, ko=4, **kw)
    assert info['status'] == 0 and info['ncoeff'] == 722
    assert info['nstamps_total'] > 600 and info['nstamps_used'] > 0.8 * info['nstamps_total']
    assert abs(info['kernel_sum'] - 1.7) < 1e-3
    hw = 10
    grown = maximum_filter(bpm, size=2 * hw + 1, mode='constant', cval=0).astype(bool)
    grown[:hw] = grown[-hw:] = True
    grown[:, :hw] = grown[:, -hw:] = True
    assert np.array_equal(d == np.float32(1e-30), grown)
    assert info['nmasked'] == int(grown.sum())
    good = ~grown
    assert np.abs(d[good]).max() < 5e-3 * np.abs(sci).max()
    assert np.abs(d[good]).mean() < 0.05
    expect = 6.0 * np.sqrt(1 + 1.7 ** 2 / (4 * np.pi * 0.81))
    assert abs(np.median(n[good]) - expect) < 0.15
    assert np.all(n[grown] == np.float32(np.sqrt(50000.0)))
    d2, n2, info2 = engine.subtract(sci, rms, ref, rms, bpm, ko=2, **kw)
    assert info2['status'] == 0 and info2['ncoeff'] == 1 + 48 * 6 + 1
    assert abs(info2['kernel_sum'] - 1.7) < 1e-3
    assert np.array_equal(d2 == np.float32(1e-30), grown)
    assert np.abs(d2[good] - d[good]).max() < 5e-3 * np.abs(sci).max()
    np.testing.assert_allclose(n2[good], n[good], rtol=2e-3)


def test_fullsize_coadd_removes_the_background_and_rescales_the_weights(engine, frame):
    """The science SWarp run with its defaults at full size (default.swarp: SUBTRACT_BACK Y,
    BACK_SIZE 128 from zuds/swarp.py:69, RESCALE_WEIGHTS Y): a smooth sky gradient of 60
    counts across the frame is gone from the coadd, and the output weight is N / (measured
    variance) whatever scale the input weight maps claim - the batched mesh statistics +
    rescale path bench.py times."""
    z = pkg()
    s = synth()
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:NY, 0:NX].astype(np.float32)
    sky = (30.0 * (xx / NX) + 20.0 * (yy / NY) ** 2 + 10.0 * np.sin(2.5 * xx / NX) * (yy / NY)).astype(np.float32)
    frames = []
    for i in range(3):
        w = s.ztf_wcs(NX, NY, dx=rng.uniform(-8, 8), dy=rng.uniform(-8, 8), rot_deg=rng.uniform(-0.05, 0.05))
        frames.append(dict(frame, wcs=w, img=frame['img'] + sky * (1.0 + 0.3 * i)))
    flat = [dict(f, img=frame['img']) for f in frames]
    p = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True, back_size=128)
    o, ow, _, _ = engine.coadd(frames, frame['wcs'], p, want_mask=False)
    f0, fw, _, _ = engine.coadd(flat, frame['wcs'], p, want_mask=False)
    good = (ow > 0) & (fw > 0)
    assert good.mean() > 0.98
    # the gradient (30 .. 78 counts peak to peak over the three frames) is gone: block medians
    # of the coadd sit at zero, and the coadd equals the one made from the flat-sky frames
    for by in range(0, NY - 512, 512):
        for bx in range(0, NX, 512):
            blk = (slice(by, by + 512), slice(bx, bx + 512))
            assert abs(np.median(o[blk][good[blk]])) < 0.35, (by, bx)
    assert np.abs(np.median((o - f0)[good])) < 0.05
    # (a 128-pixel mesh with a 3 x 3 median filter follows the curved part of this sky to ~1.5
    # counts in the corners: 2 % of the 78-count gradient)
    assert np.percentile(np.abs(o - f0)[good], 99) < 2.5
    # weights: 3 frames of sigma 6 -> 3 / 36, from weight maps that claim 1 / 36
    assert abs(np.median(ow[good]) * 36.0 / 3.0 - 1.0) < 0.05
    # ... and from maps that claim four times / a quarter of it: RESCALE_WEIGHTS removes the claim
    for c in (4.0, 0.25):
        scaled = [dict(f, wgt=(f['wgt'] * np.float32(c))) for f in frames]
        o2, ow2, _, _ = engine.coadd(scaled, frame['wcs'], p, want_mask=False)
        assert np.array_equal(ow2 > 0, ow > 0)
        np.testing.assert_allclose(ow2[good], ow[good], rtol=2e-5)
        np.testing.assert_allclose(o2[good], o[good], rtol=1e-5, atol=2e-3)
    # without the rescale the claim goes straight through
    pn = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=False, back_size=128)
    _, own, _, _ = engine.coadd([dict(f, wgt=f['wgt'] * np.float32(4.0)) for f in frames], frame['wcs'], pn,
                                want_mask=False)
    assert abs(np.median(own[good]) * 36.0 / 12.0 - 1.0) < 0.01


def test_256_frame_clipped_stack_of_baseline_config_4(engine):
    """BASELINE config[3] on one GPU: 256 frames 3072 x 3072 (TPV, dithered), CLIP_SIGMA 4 / CLIP_AMPFRAC
    0.3, frames resident in HBM, through the path every rank of the 8-GPU run takes for its row band
    (`ShardedCoadd.exact` with the collectives of a one-rank RCCL group: resampled stack, band exchange,
    k_combine_wide<4> over all 256 samples of a pixel, all-gather).  Properties: the clipped mean of 256
    noise frames has their noise / 16; the output weight is the sum of the input weights; a 40 sigma
    outlier block in one frame leaves no trace; equal to the single-call coadd bit for bit."""
    import importlib
    import os
    import torch
    import torch.distributed as dist
    z = pkg()
    s = synth()
    par = importlib.import_module('zuds-pipeline_amd.parallel')
    dmod = importlib.import_module('zuds-pipeline_amd.device')
    n, nx, ny = 256, 3072, 3072
    dev_ = torch.device('cuda', 0)
    base = s.ztf_wcs(nx, ny, tpv=True)
    rng = np.random.default_rng(4000)
    g = torch.Generator(device=dev_)
    frames = []
    for i in range(n):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-15, 15), dy=rng.uniform(-15, 15), rot_deg=rng.uniform(-0.1, 0.1), tpv=True)
        g.manual_seed(4000 + i)
        img = 200.0 + 6.0 * torch.randn((ny, nx), generator=g, device=dev_)
        frames.append(dict(img=img, wgt=torch.full((ny, nx), 1.0 / 36.0, device=dev_), wcs=w, flxscale=1.0))
    frames[7]['img'][1500:1520, 1500:1520] += 240.0                  # 40 sigma
    p = z.coadd_params(combine='CLIPPED', subtract_back=False, rescale_weights=False)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29537')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    par.FORCE_COLLECTIVES = True
    try:
        dfr = dmod.DeviceFrames(frames, dev_)
        del frames
        be = par.HipBackend(base, p, device=0, engine=engine)
        img, wgt = par.ShardedCoadd(be).exact(dfr)
        be.stream.synchronize()
        img, wgt = img.cpu().numpy(), wgt.cpu().numpy()
        del be
        dc = dmod.DeviceCoadd(base, p, device=0, engine=engine)
        dc.run(dfr)
        dc.stream.synchronize()
        assert np.array_equal(dc.img.cpu().numpy(), img) and np.array_equal(dc.wgt.cpu().numpy(), wgt)
    finally:
        par.FORCE_COLLECTIVES = False
        dist.destroy_process_group()
        engine.set_stream(0)
    inner = (slice(40, -40), slice(40, -40))
    # flux scale of the TPV frames onto the base grid is 1 within 1e-3; Lanczos-3 of white noise keeps
    # ~0.8 of its rms: the coadd's scatter is that / sqrt(256), the weight the sum of the resampled weights
    assert abs(np.median(img[inner]) - 200.0) < 0.3
    assert 0.2 < img[inner].std() < 0.42
    assert np.all(wgt[inner] > 0)
    assert abs(np.median(wgt[inner]) * img[inner].var() - 1.0) < 0.35
    blk = img[1495:1525, 1495:1525]
    assert abs(blk.mean() - np.median(img[inner])) < 0.15          # 240 / 256 = 0.94 if the block were not clipped
