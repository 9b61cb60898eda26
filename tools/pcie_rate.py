"""Developer tool: the host-pointer boundary (numpy buffers in, numpy out) timed end to end:
H2D of every frame + background / resample / WEIGHTED coadd + D2H, and the same for one
subtraction.  The PCIe-inclusive rate quoted in DESIGN.md."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z = importlib.import_module('zuds-pipeline_amd')
s = importlib.import_module('zuds-pipeline_amd.synth')

N, NF = 3072, int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(0)
base = s.ztf_wcs(N, N, tpv=True)
frames = []
for i in range(NF):
    w = s.ztf_wcs(N, N, dx=rng.uniform(-15, 15), dy=rng.uniform(-15, 15), rot_deg=rng.uniform(-0.1, 0.1))
    img = rng.normal(200, 6, (N, N)).astype(np.float32)
    mask = (rng.uniform(size=(N, N)) < 1e-3).astype(np.int32) * 256
    wgt = np.where(mask > 0, 0, 1 / 36.0).astype(np.float32)
    frames.append(dict(img=img, wgt=wgt, mask=mask, wcs=w, flxscale=0.4))
eng = z.get_engine(0)
p = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True)
eng.coadd(frames[:2], base, p)                      # warm up (allocations, code objects)
for rep in range(2):
    t0 = time.perf_counter()
    img, wgt, msk, mw = eng.coadd(frames, base, p)
    dt = time.perf_counter() - t0
    gb = NF * N * N * 12 / 1e9
    print(f'coadd of {NF} frames, host buffers: {dt * 1e3:.1f} ms = {NF * N * N / 1e6 / dt:.0f} Mpix/s '
          f'({gb:.2f} GB in: {gb / dt:.1f} GB/s)')
rms = np.full((N, N), 6.0, np.float32)
ref = img + 150
for rep in range(2):
    t0 = time.perf_counter()
    d, n, info = eng.subtract(frames[0]['img'], rms, ref, rms, (frames[0]['mask'] > 0).astype(np.uint8),
                              r=10.0, rss=24.0, nsx=10, nsy=10, nrx=3, nry=3, ko=4, bgo=0,
                              tu=5e3, iu=5e3, tl=-1e3, il=-1e3)
    dt = time.perf_counter() - t0
    print(f'subtraction, host buffers: {dt * 1e3:.1f} ms = {N * N / 1e6 / dt:.0f} Mpix/s (status {info["status"]})')
