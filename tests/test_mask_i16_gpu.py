"""int16 mask planes end to end (round 4).  ZTF masks are BITPIX 16
(`zuds/mask.py:26-72` only ever tests bits); the C-ABI takes such a plane as it is
(`zm_frame.mask_type = ZM_MASKTYPE_I16`, `zm_resample_i16`) instead of an int32 copy widened on the
host.  Every product must equal, bit for bit, what the same call gives on `mask.astype(int32)`:
the streaming box-OR kernel on int16 and int32 input (and the LDS-tiled kernel of round 3 it
replaces), negative words (bit 15: their sign extension carries bits above 15, the raw-mask
fallback), frames that cannot be staged raw (rows not a multiple of four pixels, BACK_SIZE not a
multiple of 8: the prepped path widens on the device), the materialised path, the host-pointer and
the device-pointer entry points, the device FITS decoder."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from test_fused_coadd_gpu import assert_same, stack
from util import pkg, synth

pytestmark = pytest.mark.gpu


def as16(frames):
    out = []
    for f in frames:
        g = dict(f)
        if g.get('mask') is not None:
            assert np.abs(g['mask']).max() < 2 ** 15
            g['mask'] = g['mask'].astype(np.int16)
        out.append(g)
    return out


def with_env(key, val, fn):
    old = os.environ.get(key)
    os.environ[key] = val
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = old


@pytest.mark.parametrize('mask_kind', ['AND', 'OR'])
@pytest.mark.parametrize('combine', ['WEIGHTED', 'CLIPPED'])
def test_int16_masks_give_the_bits_of_their_int32_copies(engine, mask_kind, combine):
    z = pkg()
    frames, wout = stack(5, 700, 650, 1100, nbad=600)
    rng = np.random.default_rng(5)
    for f in frames:                                   # every low bit somewhere, bit 14 in blocks
        f['mask'][rng.integers(0, 650, 300), rng.integers(0, 700, 300)] = rng.integers(1, 2 ** 15, 300)
        f['mask'][40:47, 100:109] |= 1 << 14
    p = z.coadd_params(combine=combine, mask_combine=mask_kind, subtract_back=True, rescale_weights=True,
                       back_size=128)
    a = engine.coadd(frames, wout, p)
    b = engine.coadd(as16(frames), wout, p)
    assert_same(a, b)
    assert (b[2] != 0).any()
    # the materialised path (k_resample frame by frame, its box-OR plane from the LDS-tiled k_mask_box) agrees as well
    d = with_env('ZM_COADD_FUSED', '0', lambda: engine.coadd(as16(frames), wout, p))
    assert_same(a, d)


def test_negative_int16_words_mean_their_sign_extension(engine):
    """numpy's astype(int32) sign-extends: a word with bit 15 carries bits 16 .. 31 afterwards, the
    16-bit box-OR plane defers to the raw mask there (ZM_BOX_RAW) and the raw int16 words are read."""
    z = pkg()
    frames, wout = stack(4, 450, 430, 1200)
    m16 = []
    for i, f in enumerate(frames):
        m = f['mask'].astype(np.int16)
        m[100 + 20 * i:140, 200:260] |= np.int16(-32768)            # bit 15 alone
        m[300:310, 50 + 5 * i:90] = np.int16(-1)                     # every bit
        m[20:23, 400:403] = np.int16(0x7fff)
        m16.append(m)
    f16 = [dict(f, mask=m) for f, m in zip(frames, m16)]
    f32 = [dict(f, mask=m.astype(np.int32)) for f, m in zip(frames, m16)]
    for kind in ('AND', 'OR'):
        p = z.coadd_params(combine='WEIGHTED', mask_combine=kind, subtract_back=False, rescale_weights=False)
        a = engine.coadd(f32, wout, p)
        b = engine.coadd(f16, wout, p)
        assert_same(a, b)
        # (bit 31 is the coadd's own "never covered" marker: a negative word poisons its pixel the same way
        # in both representations - mask flags live in bits 0 .. 30)
        if kind == 'OR':
            assert (b[2] & 0x4000).any()


def test_frames_that_cannot_be_staged_raw_widen_on_the_device(engine):
    """nx not a multiple of four (no 16-byte rows) and BACK_SIZE 100 (not a multiple of 8): such frames
    are prepped into a plane by k_prep_box, which reads int32 - their int16 masks are widened on the
    device first.  Mixed stacks: int16 and int32 masks, masked and unmasked frames side by side."""
    z = pkg()
    frames, wout = stack(4, 451, 433, 1300)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='OR', subtract_back=True, rescale_weights=True, back_size=100)
    a = engine.coadd(frames, wout, p)
    assert_same(a, engine.coadd(as16(frames), wout, p))
    frames, wout = stack(5, 640, 600, 1400)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='AND', subtract_back=True, rescale_weights=True)
    mixed = as16(frames)
    mixed[1]['mask'] = frames[1]['mask']                # int32
    mixed[3]['mask'] = None
    ref = [dict(f) for f in frames]
    ref[3]['mask'] = None
    assert_same(engine.coadd(ref, wout, p), engine.coadd(mixed, wout, p))


def test_zm_resample_i16_and_the_device_entry_points(engine):
    import torch
    z = pkg()
    s = synth()
    dmod = importlib.import_module('zuds-pipeline_amd.device')
    nx, ny = 600, 560
    rng = np.random.default_rng(9)
    w = s.ztf_wcs(nx, ny, tpv=True)
    wout = s.ztf_wcs(nx, ny, dx=3.4, dy=-2.7, rot_deg=0.05, tpv=True)
    f = s.make_frame(nx, ny, 77, w, nbad=300)
    m16 = f['mask'].astype(np.int16)
    m16[50:60, 70:90] = np.int16(-32768)
    a = engine.resample(f['img'], w, wout, wgt=f['wgt'], mask=m16.astype(np.int32))
    b = engine.resample(f['img'], w, wout, wgt=f['wgt'], mask=m16)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # mask alone (run_align on a MaskImage, zuds/swarp.py:186-191)
    a = engine.resample_mask(m16.astype(np.int32), w, wout)
    b = engine.resample_mask(m16, w, wout)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    # zm_mask_widen_dev == astype(int32); odd lengths, unaligned starts
    L = engine.L
    for n, off in ((nx * ny, 0), (1001, 1), (7, 3), (4, 0), (3, 0)):
        src = torch.from_numpy(m16.ravel()[: n + off].copy()).to('cuda')[off:]
        dst = torch.empty(n + 1, dtype=torch.int32, device='cuda')[1:]
        z._lib.check(L.zm_mask_widen_dev(engine.ctx, src.data_ptr(), n, dst.data_ptr()))
        engine.synchronize()
        assert np.array_equal(dst.cpu().numpy(), m16.ravel()[off: n + off].astype(np.int32))
    # DeviceFrames keeps an int16 tensor at 16 bits and says so in the descriptor
    fr = dmod.DeviceFrames([dict(f, mask=m16)], torch.device('cuda', 0))
    assert fr.tensors[0][2].dtype == torch.int16 and fr.arr[0].mask_type == z._lib.MASKTYPE_I16
    p = z.coadd_params(combine='WEIGHTED', mask_combine='OR', subtract_back=False, rescale_weights=False)
    co = dmod.DeviceCoadd(wout, p, device=0, engine=engine, want_mask=True)
    co.run(fr)
    engine.synchronize()
    want = engine.coadd([dict(f, mask=m16.astype(np.int32))], wout, p)
    assert np.array_equal(co.mask.cpu().numpy(), want[2]) and np.array_equal(co.img.cpu().numpy(), want[0])


def test_fullsize_int16_mask_resample_is_the_or_under_the_footprint(engine):
    """tests/test_fullsize_gpu.py::test_mask_resample_is_the_or_under_the_footprint on an int16 plane at
    3072 x 3080, through the coadd entry point (box-OR planes from the streaming kernel)."""
    s = synth()
    z = pkg()
    NX, NY = 3072, 3080
    rng = np.random.default_rng(42)
    mask = np.zeros((NY, NX), np.int16)
    bad = rng.integers(0, NX * NY, 9000)
    mask.ravel()[bad] = rng.choice([1, 256, 2, 2048, 0x4000], bad.size)
    mask[300:303, 700:703] = np.int16(0x7fff)              # (no negative words here: bit 31 is the fold's marker)
    mask[NY - 3:, :] |= 4                                  # the last rows and columns: band / strip edges
    mask[:, NX - 2:] |= 8
    mask[120, :] |= 16
    mask[121 * 7 - 1: 121 * 7 + 1, 247:250] |= 32          # across a band and a strip boundary of the kernel
    w = s.ztf_wcs(NX, NY, tpv=True)
    wout = s.ztf_wcs(NX, NY, dx=-11.4, dy=6.3, tpv=True)   # out (x, y) = in (x + 11.4, y - 6.3)
    _, _, om, ow = engine.resample_mask(mask, w, wout)
    m32 = mask.astype(np.int32)
    want = np.zeros_like(m32)
    ys, xs = np.arange(NY)[:, None], np.arange(NX)[None, :]
    inside = np.ones((NY, NX), bool)
    for r in range(-9, -3):
        for c in range(9, 15):
            yy, xx = ys + r, xs + c
            ok = (yy >= 0) & (yy < NY) & (xx >= 0) & (xx < NX)
            inside &= ok
            want |= np.where(ok, m32[np.clip(yy, 0, NY - 1), np.clip(xx, 0, NX - 1)], 0)
    assert np.array_equal(ow > 0, inside)
    assert np.array_equal(om[inside], want[inside])
    assert (om[inside] == 0x7fff).any() and (om[inside] & 32).any()
    _, _, om32, ow32 = engine.resample_mask(m32, w, wout)
    assert np.array_equal(om32, om) and np.array_equal(ow32, ow)


def test_device_fits_decoder_keeps_a_bitpix16_mask_at_16_bits(engine, tmp_path):
    import torch
    z = pkg()
    dmod = importlib.import_module('zuds-pipeline_amd.device')
    rng = np.random.default_rng(2)
    m = rng.integers(-2 ** 15, 2 ** 15, (37, 52)).astype(np.int16)
    p = tmp_path / 'mskimg.fits'
    z.fits.write(p, m, {'OBJECT': 'x'})
    io = dmod.FITSDeviceIO(device=0, engine=engine)
    t, _ = io.load(str(p), 'mask')
    torch.cuda.synchronize()
    assert t.dtype == torch.int16 and np.array_equal(t.cpu().numpy(), m)
    t32, _ = io.load(str(p), 'i32')
    torch.cuda.synchronize()
    assert np.array_equal(t32.cpu().numpy(), m.astype(np.int32))
    # a scaled 16-bit file (unsigned convention) is not a plain mask: int32
    u = rng.integers(0, 2 ** 16, (9, 11)).astype(np.uint16)
    p2 = tmp_path / 'u16.fits'
    z.fits.write(p2, u, {})
    t, _ = io.load(str(p2), 'mask')
    torch.cuda.synchronize()
    assert t.dtype == torch.int32 and np.array_equal(t.cpu().numpy(), u.astype(np.int32))
