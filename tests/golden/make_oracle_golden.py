#!/usr/bin/env python
"""Small committed input/output vectors of the numpy oracle (oracle/), one file per
stage of the path.  They are regression pins (the oracle must keep producing them
bit for bit in fp64) and the fixtures the GPU parity tests compare the HIP path
with when no oracle run is wanted.  NOT reference outputs: SWarp / hotpants /
SExtractor cannot run here (oracle/__init__.py).

    python tests/golden/make_oracle_golden.py

Inputs are stored as float32 / int32 (what the C-ABI takes), expected outputs as
float64 (float32 where noted) so the files stay below ~300 KB each.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import background as oback          # noqa: E402
from oracle import combine as ocombine          # noqa: E402
from oracle import hotpants as ohp              # noqa: E402
from oracle import photometry as ophot          # noqa: E402
from oracle import resample as ores             # noqa: E402
from oracle.wcs import WCS as OWCS              # noqa: E402

synth = importlib.import_module('zuds-pipeline_amd.synth')


def ow(w):
    return OWCS(w.crpix, w.crval, w.cd, w.pv1 if w.has_pv else None,
                w.pv2 if w.has_pv else None, w.naxis)


def hdr(w):
    """Plain dict of header cards (json / npz friendly)."""
    return {k: v for k, v in w.to_header().items()}


def save(name, **arrs):
    p = os.path.join(HERE, name)
    np.savez_compressed(p, **arrs)
    print(f'{name}: {os.path.getsize(p) / 1024:.0f} KiB')


def g_resample():
    nx, ny = 96, 80
    win = synth.ztf_wcs(nx, ny, tpv=True)
    wout = synth.ztf_wcs(nx, ny, dx=2.37, dy=-1.61, rot_deg=0.35, tpv=True)
    f = synth.make_frame(nx, ny, 101, win, nstars=12, nbad=25)
    px, py = ores.positions(ow(wout), ow(win), nx, ny)
    fs = ores.flux_scale(ow(win), ow(wout), 0.37)
    out = {}
    for kind, nm in ((ores.LANCZOS3, 'lanczos3'), (ores.BILINEAR, 'bilinear'), (ores.NEAREST, 'nearest')):
        o, w, m = ores.resample(f['img'], f['wgt'], px, py, kind, fs, f['mask'])
        out[nm + '_img'], out[nm + '_wgt'], out[nm + '_mask'] = o, w, m.astype(np.int32)
    save('oracle_resample.npz', img=f['img'], wgt=f['wgt'], mask=f['mask'].astype(np.int32),
         win=np.array(list(hdr(win).items()), dtype=object), wout=np.array(list(hdr(wout).items()), dtype=object),
         naxis=np.array([nx, ny]), flxscale=np.float64(0.37), fscale=np.float64(fs), px=px, py=py, **out)


def g_background():
    rng = np.random.default_rng(202)
    ny, nx = 200, 264           # ragged: 264 = 4 x 64 + 8, 200 = 3 x 64 + 8
    yy, xx = np.mgrid[0:ny, 0:nx]
    img = 120.0 + 0.05 * xx - 0.03 * yy + rng.normal(0, 4.0, (ny, nx))
    synth.add_stars(img, rng.uniform(5, nx - 5, 25), rng.uniform(5, ny - 5, 25),
                    np.exp(rng.uniform(np.log(1e3), np.log(5e4), 25)), 2.2)
    img = img.astype(np.float32)
    wgt = np.full((ny, nx), 1 / 16.0, np.float32)
    wgt[40:60, 100:180] = 0
    wgt.ravel()[rng.integers(0, nx * ny, 300)] = 0
    bkg, rms, bmean, bsig, back, sigm = oback.background(img, wgt, 64)
    save("oracle_background.npz", img=img, wgt=wgt, mesh=np.int64(64), bkg=bkg.astype(np.float32),
         rms=rms.astype(np.float32),
         backmean=np.float64(bmean), backsig=np.float64(bsig), nodes_back=back, nodes_sigma=sigm)


def g_combine():
    rng = np.random.default_rng(303)
    n, ny, nx = 9, 24, 40
    vals = rng.normal(100, 5, (n, ny, nx))
    wgts = rng.uniform(0.02, 0.08, (n, ny, nx))
    wgts[rng.uniform(size=wgts.shape) < 0.15] = 0.0
    wgts[:, 3, 5] = 0.0                       # a pixel nobody covers
    wgts[1:, 4, 6] = 0.0                      # a pixel only one frame covers
    vals[2][rng.uniform(size=(ny, nx)) < 0.05] += 400.0      # outliers in one frame
    vals = vals.astype(np.float32)
    wgts = wgts.astype(np.float32)
    out = {}
    for kind in ('WEIGHTED', 'CLIPPED', 'MEDIAN', 'AVERAGE'):
        img, wgt, nused = ocombine.combine(vals.astype(np.float64), wgts.astype(np.float64), kind)
        out[kind + '_img'], out[kind + '_wgt'] = img, wgt
    masks = rng.choice([0, 0, 0, 1, 2, 256, 2048, 257], size=(n, ny, nx)).astype(np.int32)
    cov = wgts > 0
    for kind in ('AND', 'OR'):
        m, c = ocombine.combine_masks(masks, cov, kind)
        out['mask_' + kind], out['cov_' + kind] = m.astype(np.int32), c.astype(np.int32)
    save('oracle_combine.npz', vals=vals, wgts=wgts, masks=masks, **out)


def g_hotpants():
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(404)
    nx, ny = 160, 144
    ref = np.full((ny, nx), 150.0)
    synth.add_stars(ref, rng.uniform(8, nx - 8, 45), rng.uniform(8, ny - 8, 45),
                    np.exp(rng.uniform(np.log(3e3), np.log(6e4), 45)), 2.0)
    sci = 1.25 * gaussian_filter(ref, 0.8, mode='nearest') + 12.0
    ref = (ref + rng.normal(0, 0.5, ref.shape)).astype(np.float32)
    sci = (sci + rng.normal(0, 3.0, sci.shape)).astype(np.float32)
    bpm = np.zeros((ny, nx), np.uint8)
    bpm[60:63, 70:73] = 1
    bpm[20, 130] = 1
    srms = np.full((ny, nx), 3.0, np.float32)
    rrms = np.full((ny, nx), 0.5, np.float32)
    kw = dict(r=4.0, rss=8.0, nsx=4, nsy=4, nrx=1, nry=1, ko=1, bgo=0, tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
    d, n, info = ohp.subtract(sci, ref, srms, rrms, bpm, **kw)
    reg = [r for r in info['regions'] if r is not None][0]
    save('oracle_hotpants.npz', sci=sci, ref=ref, sci_rms=srms, ref_rms=rrms, bpm=bpm,
         kw=np.array(list(kw.items()), dtype=object), diff=d, noise=n,
         nstamps_total=np.int64(reg['nstamps_total']), nstamps_used=np.int64(reg['nstamps_used']),
         niter=np.int64(reg['niter']), kernel_sum=np.float64(reg['kernel_sum']),
         nmasked=np.int64(info['nmasked']))


def g_photometry():
    rng = np.random.default_rng(505)
    ny, nx = 64, 72
    data = rng.normal(0, 3, (ny, nx)).astype(np.float32)
    synth.add_stars(data, [20.3, 50.77, 3.2], [30.6, 12.25, 60.9], [5e3, 2e4, 8e3], 2.1)
    rms = rng.uniform(2.5, 3.5, (ny, nx)).astype(np.float32)
    mask = np.zeros((ny, nx), np.int32)
    mask[29:32, 21] = 256
    mask[12, 50] = 2
    x = np.array([20.3, 50.77, 3.2, 36.0, 71.4, 10.5, -5.0])
    y = np.array([30.6, 12.25, 60.9, 32.0, 1.2, 10.5, 20.0])
    flux, err, flags = ophot.aperture_photometry(data, rms, mask, x, y, 3.0)
    save('oracle_photometry.npz', data=data, rms=rms, mask=mask, x=x, y=y, r=np.float64(3.0),
         flux=flux, fluxerr=err, flags=flags)


if __name__ == '__main__':
    g_resample()
    g_background()
    g_combine()
    g_hotpants()
    g_photometry()
