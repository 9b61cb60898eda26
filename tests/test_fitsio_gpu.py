"""FITS data blocks decoded / encoded on the device against the host
reader / writer, which tests/test_golden.py pins against astropy."""
import importlib
import os

import numpy as np
import pytest

from util import pkg, synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def io(engine):
    dev = importlib.import_module('zuds-pipeline_amd.device')
    o = dev.FITSDeviceIO(0, engine=engine)
    yield o
    engine.set_stream(None)


def test_decode_matches_host_reader_on_astropy_files(io):
    z = pkg()
    for name, kind, dt in (('astropy_f32.fits', 'f32', np.float32), ('astropy_i16.fits', 'i32', np.int32),
                           ('astropy_u8.fits', 'u8', np.uint8), ('astropy_i16.fits', 'f32', np.float32)):
        p = os.path.join(GOLD, name)
        t, hdr = io.load(p, kind)
        ref, rhdr, _ = z.fits.read(p)
        got = t.cpu().numpy()
        assert got.dtype == dt and got.shape == ref.shape
        assert np.array_equal(got, ref.astype(dt), equal_nan=True)
        assert hdr == rhdr


def test_every_bitpix_and_unsigned_conventions(io, tmp_path):
    z = pkg()
    rng = np.random.default_rng(3)
    cases = [rng.normal(0, 1e3, (33, 47)).astype(np.float32),
             rng.normal(0, 1e3, (33, 47)).astype(np.float64),
             rng.integers(-2 ** 31, 2 ** 31 - 1, (33, 47)).astype(np.int32),
             rng.integers(0, 65535, (33, 47)).astype(np.uint16),       # BZERO 32768
             rng.integers(-32768, 32767, (33, 47)).astype(np.int16),
             rng.integers(0, 255, (33, 47)).astype(np.uint8)]
    for k, a in enumerate(cases):
        p = str(tmp_path / f'c{k}.fits')
        z.fits.write(p, a, {'OBJECT': 'x'})
        ref = z.fits.read(p)[0]
        kind = 'f32' if a.dtype.kind == 'f' else 'i32'
        got = io.load(p, kind)[0].cpu().numpy()
        want = ref.astype(np.float32) if kind == 'f32' else ref.astype(np.int32)
        assert np.array_equal(got, want), a.dtype


def test_encode_writes_the_same_bytes_as_the_host_writer(io, tmp_path):
    import torch
    z = pkg()
    rng = np.random.default_rng(4)
    hdr = {'MAGZP': 26.5, 'SEEING': 2.1, 'FILTER': 'ZTF_r', 'FLAG': True}
    for a, bp in ((rng.normal(0, 50, (40, 56)).astype(np.float32), None),
                  (rng.integers(0, 70000, (40, 56)).astype(np.int32), None),
                  (rng.integers(0, 3000, (40, 56)).astype(np.int32), 16),
                  ((rng.uniform(size=(40, 56)) < 0.3).astype(np.uint8), None)):
        p1, p2 = str(tmp_path / 'dev.fits'), str(tmp_path / 'host.fits')
        io.save(p1, torch.from_numpy(a).cuda(), hdr, bitpix=bp)
        z.fits.write(p2, a.astype(np.int16) if bp == 16 else a, hdr)
        assert open(p1, 'rb').read() == open(p2, 'rb').read()


def test_coadd_from_files_equals_coadd_from_arrays(io, engine, tmp_path):
    z = pkg()
    s = synth()
    dev = importlib.import_module('zuds-pipeline_amd.device')
    base = s.ztf_wcs(220, 180, tpv=True)
    frames, sci, wgt, msk = [], [], [], []
    for i in range(3):
        w = s.ztf_wcs(220, 180, dx=1.7 * i, dy=-1.1 * i, rot_deg=0.05 * i, tpv=True)
        f = s.make_frame(220, 180, 30 + i, w, nstars=12, nbad=40, magzp=25.6 + 0.2 * i)
        frames.append(f)
        for lst, key, suffix in ((sci, 'img', 'sci'), (wgt, 'wgt', 'weight'), (msk, 'mask', 'mask')):
            p = str(tmp_path / f'f{i}.{suffix}.fits')
            z.fits.write(p, f[key] if key != 'mask' else f[key].astype(np.int16), f['header'])
            lst.append(p)
    p = z.coadd_params(combine='CLIPPED', subtract_back=True, rescale_weights=True, back_size=64)
    h_img, h_wgt, h_msk, _ = engine.coadd(frames, base, p, want_mask=True)
    df, _ = io.load_frames(sci, wgt, msk)
    dc = dev.DeviceCoadd(base, p, device=0, engine=engine, want_mask=True)
    dc.run(df)
    import torch
    torch.cuda.synchronize()
    assert np.array_equal(dc.img.cpu().numpy(), h_img) and np.array_equal(dc.wgt.cpu().numpy(), h_wgt)
    assert np.array_equal(dc.mask.cpu().numpy(), h_msk)
