"""The sample-bracketed select (``csrc/select_bracket.hip``, round 6): ``zm_median_mad*`` on frames of a megapixel
and more stream the frame twice instead of six times.  The answers must stay what ``quick_background_estimate``
(``zuds/utils.py:32-53``) gives - ``np.median`` of the unmasked float32 pixels and 1.4826 x the median of their
absolute float32 deviations - bit for bit, on sky-like frames and on the frames a bracket is bad at: ties, two
populations, few valid pixels, and a frame built so that the sample says nothing about the rest (the select then
runs again in its one-workgroup three-pass form, on the device, without the host hearing of it)."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAMPLE4 = 1024                     # RS2_SAMPLE4


def pkg():
    return importlib.import_module('zuds-pipeline_amd')


@pytest.fixture(scope='module')
def engine():
    return pkg().Engine(0)


def expect(img, mask):
    pix = img.ravel() if mask is None else img.ravel()[mask.ravel() == 0]
    pix = pix[~np.isnan(pix)]
    med = np.median(pix)
    mad = 1.4826 * float(np.median(np.abs(pix - med)))
    return float(med), mad, pix.size


def check(engine, img, mask, what):
    med, mad = engine.median_mad(img, mask)
    rmed, rmad, _ = expect(img, mask)
    assert med == rmed, (what, med, rmed)
    assert mad == pytest.approx(rmad, rel=1e-12, abs=0), (what, mad, rmad)


def sample_positions(n):
    """``rs2_sample_pos`` of the kernel: the float4 group of each of the 1 024 sampled groups."""
    stride = (n // 4) // SAMPLE4
    j = np.arange(SAMPLE4, dtype=np.uint64)
    h = ((j * np.uint64(2654435761)) & np.uint64(0xffffffff)) ^ (((j * np.uint64(40503)) & np.uint64(0xffffffff)) >> np.uint64(3))
    return (j * np.uint64(stride) + h % np.uint64(stride)).astype(np.int64)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_sky_frames_with_stars_masks_and_nans(engine, seed):
    rng = np.random.default_rng(seed)
    shape = [(3072, 3072), (1025, 1027), (2048, 1500)][seed]          # (1025 x 1027: n % 4 = 3, the scalar tail)
    n = shape[0] * shape[1]
    img = rng.normal(150 + 30 * seed, 3 + 4 * seed, n).astype(np.float32)
    stars = rng.random(n) < 0.02
    img[stars] += rng.uniform(100, 60000, int(stars.sum())).astype(np.float32)
    img += np.linspace(-3, 3, n, dtype=np.float32)                     # a gradient down the frame
    img[rng.random(n) < 1e-4] = np.nan
    mask = (rng.random(n) < 0.07).astype(np.int32) * rng.choice([1, 2, 256, 65536], n).astype(np.int32)
    check(engine, img.reshape(shape), mask.reshape(shape), ('sky', seed))
    check(engine, img.reshape(shape), None, ('sky, no mask', seed))


def test_ties_flat_frames_and_two_populations(engine):
    rng = np.random.default_rng(7)
    n = 1536 * 2048
    # a flat frame: every key is the median
    check(engine, np.full(n, 150.0, np.float32), None, 'flat')
    # integers: ties everywhere, the bracket ends on bins of one key
    check(engine, rng.integers(140, 160, n).astype(np.float32), (rng.random(n) < 0.1).astype(np.int32), 'integers')
    # a tie mass at the median inside a continuous population (a wide bracket full of equal keys: the segments
    # overflow and the select is repeated on the device)
    a = rng.normal(0, 1, n).astype(np.float32)
    a[rng.random(n) < 0.3] = 0.0
    check(engine, a, None, 'tie mass at the median')
    # two populations with nothing at the median (the bracket spans the gap) - and the even count's two middle
    # ranks on either side of it
    b = np.concatenate([rng.normal(-100, 1, n // 2), rng.normal(1e4, 50, n - n // 2)]).astype(np.float32)
    rng.shuffle(b)
    check(engine, b, None, 'two populations')
    # both signs, tiny and huge magnitudes
    c = (rng.standard_cauchy(n) * 1e-3).astype(np.float32)
    c[::1000] = 3e38
    c[1::1000] = -3e38
    check(engine, c, None, 'cauchy')
    # zeros of both signs around the median
    d = rng.normal(0, 1e-3, n).astype(np.float32)
    d[::3] = 0.0
    d[1::3] = -0.0
    check(engine, d, None, 'signed zeros')


def test_few_valid_pixels_and_a_valid_corner(engine):
    rng = np.random.default_rng(3)
    n = 2048 * 2048
    img = rng.normal(150, 5, n).astype(np.float32)
    mask = np.ones(n, np.int32)
    mask[rng.random(n) < 1e-3] = 0                                   # ~4 000 valid pixels, ~4 of them in the sample
    check(engine, img, mask, 'one pixel in a thousand')
    mask = np.ones(n, np.int32)
    mask[:5000] = 0                                                    # valid pixels the sample all but misses
    check(engine, img, mask, 'a valid corner')
    mask = np.ones(n, np.int32)
    mask[12345] = 0
    check(engine, img, mask, 'one valid pixel')
    z = pkg()
    with pytest.raises(z.ZMError):
        engine.median_mad(img, np.ones(n, np.int32))


def test_a_frame_whose_sample_lies(engine):
    """Every sampled pixel is 0 and the rest of the frame is sky: the bracket [0, 0] cannot hold the median; the
    select notices (rank outside the bracket) and runs again in its three-pass form on the device."""
    rng = np.random.default_rng(5)
    n = 2048 * 2048
    img = rng.normal(150, 5, n).astype(np.float32)
    q = sample_positions(n)
    for c in range(4):
        img[4 * q + c] = 0.0
    rmed, _, _ = expect(img, None)
    assert rmed > 100                                                  # (the sample is 0.1 % of the frame)
    check(engine, img, None, 'lying sample, median')
    # the sample is right about the median and wrong about the deviations: sampled pixels sit AT the median
    img2 = rng.normal(150, 5, n).astype(np.float32)
    med = np.float32(np.median(img2))
    for c in range(4):
        img2[4 * q + c] = med
    check(engine, img2, None, 'lying sample, MAD')


def test_two_frames_in_one_call_on_the_device(engine):
    import torch
    z = pkg()
    rng = np.random.default_rng(11)
    n = 3072 * 3072
    a = rng.normal(150, 5, n).astype(np.float32)
    b = rng.normal(-3, 40, n).astype(np.float32)
    b[rng.random(n) < 0.2] = 0.0                                       # uncovered pixels of an aligned reference
    ma = (rng.random(n) < 0.1).astype(np.int32) * 256
    mb = (rng.random(n) < 0.3).astype(np.int32) * 2
    dev = lambda x: torch.from_numpy(x).cuda()                       # noqa: E731
    da, db, dma, dmb = dev(a), dev(b), dev(ma), dev(mb)
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    out = (C.c_double * 4)()
    z._lib.check(engine.L.zm_median_mad2_dev(engine.ctx, da.data_ptr(), dma.data_ptr(), db.data_ptr(), dmb.data_ptr(), n, out))
    out6 = torch.zeros(6, dtype=torch.float64, device='cuda')
    for _ in range(2):                                                 # (twice: the state of the first call is not in the way)
        z._lib.check(engine.L.zm_median_mad2_async_dev(engine.ctx, da.data_ptr(), dma.data_ptr(), db.data_ptr(),
                                                       dmb.data_ptr(), n, out6.data_ptr()))
    torch.cuda.synchronize()
    o6 = out6.cpu().numpy()
    for k, (x, m) in enumerate(((a, ma), (b, mb))):
        rmed, rmad, cnt = expect(x, m)
        assert out[2 * k] == rmed and out[2 * k + 1] == pytest.approx(rmad, rel=1e-12, abs=0)
        assert o6[3 * k] == rmed and o6[3 * k + 1] == pytest.approx(rmad, rel=1e-12, abs=0) and o6[3 * k + 2] == cnt
