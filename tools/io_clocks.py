"""Developer tool: the three clocks of SURVEY.md 8(d) for a coadd of NF 3072^2 frames whose
inputs are FITS files on local disk (page cache warm): kernels only (device resident),
device time with PCIe, wall clock with FITS I/O - host decode path vs device decode path."""
import importlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

z = importlib.import_module('zuds-pipeline_amd')
s = importlib.import_module('zuds-pipeline_amd.synth')
dev = importlib.import_module('zuds-pipeline_amd.device')

N, NF = 3072, int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = tempfile.mkdtemp(prefix='zmio_')
try:
    rng = np.random.default_rng(0)
    base = s.ztf_wcs(N, N, tpv=True)
    sci, wgt, msk = [], [], []
    for i in range(NF):
        w = s.ztf_wcs(N, N, dx=rng.uniform(-15, 15), dy=rng.uniform(-15, 15), rot_deg=rng.uniform(-0.1, 0.1))
        hdr = dict(w.to_header(), NAXIS1=N, NAXIS2=N, MAGZP=26.0 + 0.01 * i, SEEING=2.0)
        img = rng.normal(200, 6, (N, N)).astype(np.float32)
        m = ((rng.uniform(size=(N, N)) < 1e-3) * 256).astype(np.int16)
        for lst, arr, suf in ((sci, img, 'sci'), (wgt, np.where(m > 0, 0, 1 / 36.0).astype(np.float32), 'weight'),
                              (msk, m, 'mask')):
            p = os.path.join(d, f'f{i:02d}.{suf}.fits')
            z.fits.write(p, arr, hdr)
            lst.append(p)
    eng = z.get_engine(0)
    p = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True)
    mpix = NF * N * N / 1e6

    def host_path():
        frames = []
        for a, b, c in zip(sci, wgt, msk):
            img, hdr, _ = z.fits.read(a)
            frames.append(dict(img=img, wgt=z.fits.read(b)[0], mask=z.fits.read(c)[0].astype(np.int32),
                               wcs=z.wcs.WCS.from_header(hdr), flxscale=10 ** (-0.4 * (hdr['MAGZP'] - 25))))
        t1 = time.perf_counter()
        out = eng.coadd(frames, base, p)
        t2 = time.perf_counter()
        for arr, suf in zip(out[:3], ('coadd', 'coadd.weight', 'coadd.mask')):
            z.fits.write(os.path.join(d, f'host.{suf}.fits'), arr, base.to_header())
        return t1, t2

    io = dev.FITSDeviceIO(0, engine=eng)
    dc = dev.DeviceCoadd(base, p, device=0, engine=eng, want_mask=True)

    def device_path():
        df, _ = io.load_frames(sci, wgt, msk)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dc.run(df)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for t, suf in ((dc.img, 'coadd'), (dc.wgt, 'coadd.weight'), (dc.mask, 'coadd.mask')):
            io.save(os.path.join(d, f'dev.{suf}.fits'), t, base.to_header())
        return t1, t2

    for name, fn in (('host decode (numpy) + host-pointer C-ABI', host_path),
                     ('device decode + device-resident C-ABI', device_path)):
        fn()                                     # warm: page cache, allocations
        t0 = time.perf_counter()
        t1, t2 = fn()
        t3 = time.perf_counter()
        print(f'{name}: read {1e3 * (t1 - t0):.0f} ms, coadd {1e3 * (t2 - t1):.1f} ms, write '
              f'{1e3 * (t3 - t2):.0f} ms; wall {1e3 * (t3 - t0):.0f} ms = {mpix / (t3 - t0):.0f} Mpix/s')
    a = z.fits.read(os.path.join(d, 'host.coadd.fits'))[0]
    b = z.fits.read(os.path.join(d, 'dev.coadd.fits'))[0]
    print('products identical:', np.array_equal(a, b))
finally:
    shutil.rmtree(d, ignore_errors=True)
