"""Developer tool: J subtractions side by side (nightly.SubtractionPool) on synthetic config-2 data,
for a kernel trace.  usage: nightly_trace.py J [njobs [batch [mixed]]]   (batch >= 2: J lanes of batched fits; rocprofv3 --kernel-trace -- python3 tools/nightly_trace.py 4)
Prints ms per subtraction; tools/rocpd_overlap.py turns the trace into per-kernel times and the overlap."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    J = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    njobs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    mixed = len(sys.argv) > 4 and sys.argv[4] == 'mixed'      # three seeing groups: r = 9 / 10 / 11, rss = 21 / 24 / 26
    size = 3072
    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    dev = importlib.import_module('zuds-pipeline_amd.device')
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    eng = z.Engine(0)
    base, frames = bench.make_device_frames(synth, torch, njobs, size, 2000, device)
    p = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True)
    co = dev.DeviceCoadd(base, p, device=0, engine=eng, want_mask=True)
    co.run(dev.DeviceFrames(frames, device))
    npx = size * size
    L, check = eng.L, z._lib.check
    ref_rms = torch.empty_like(co.wgt)
    with torch.cuda.stream(co.stream):
        check(L.zm_mask_flag_dev(eng.ctx, co.mask.data_ptr(), co.mask_wgt.data_ptr(), 0.0, 1 << 16, npx))
        check(L.zm_add_scalar_dev(eng.ctx, co.img.data_ptr(), 150.0, npx))
        check(L.zm_rms_from_weight_dev(eng.ctx, co.wgt.data_ptr(), None, npx, float(np.sqrt(50000.0)), ref_rms.data_ptr()))
    co.stream.synchronize()
    ref = dict(img=co.img, rms=ref_rms, mask=co.mask, wcs=base, flxscale=1.0)
    rng = np.random.default_rng(5)
    ra, dec = base.all_pix2world(rng.uniform(50, size - 50, 500), rng.uniform(50, size - 50, 500), 0)
    jobs = []
    g = torch.Generator(device='cpu')
    for i, f in enumerate(frames):
        g.manual_seed(177 + i)
        bx = torch.randint(2, size - 2, (300,), generator=g)
        by = torch.randint(2, size - 2, (300,), generator=g)
        m = torch.zeros((size, size), dtype=torch.int32)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                m[by + dy, bx + dx] = 256
        m = m.to(device)
        wgt = torch.where(m != 0, 0.0, float(f['wgt'].max())).to(torch.float32)
        rms = torch.where(wgt > 0, 1.0 / torch.sqrt(wgt.clamp_min(1e-20)), float(np.sqrt(50000.0))).to(torch.float32)
        jobs.append(nm.SubtractionJob(dict(img=f['img'], rms=rms, mask=m, wgt=wgt, wcs=f['wcs'], seeing=(3.6, 4.0, 4.4)[i % 3] if mixed else 4.0), ref,
                                      radec=(ra, dec), nreg_side=3))
    pool = nm.SubtractionPool(J, device=0, batch=batch)
    try:
        pool.map(jobs[:J * max(batch, 1)], keep=False)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            res = pool.map(jobs, keep=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'J = {J} batch = {batch}: {1e3 * dt / njobs:.2f} ms per subtraction ({njobs} jobs, '
                  f'{sum(("error" in r) or r["info"]["status"] != 0 for r in res)} failed)', flush=True)
    finally:
        pool.close()


if __name__ == '__main__':
    main()
