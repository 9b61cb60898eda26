"""Developer tool: the device-resident subtraction at other parameters than the bench's (regions per
side, kernel order, seeing): ms per subtraction and the per-scope kernel times."""
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
    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    dev = importlib.import_module('zuds-pipeline_amd.device')
    size = 3072
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    eng = z.Engine(0)
    base, frames = bench.make_device_frames(synth, torch, 9, size, 2000, device)
    sci = frames.pop()
    m = torch.zeros((size, size), dtype=torch.int32, device=device)
    sci['mask'] = m
    sci['wgt'] = torch.full((size, size), float(sci['wgt'].max()), device=device)
    sci['rms'] = 1.0 / torch.sqrt(sci['wgt'])
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
    sub = dev.DeviceSubtraction(sci['wcs'], base, device=0, engine=eng, stream=co.stream)
    names = ['resample', 'median_mad', 'hp_masks', 'hp_cells', 'hp_vectors', 'hp_gram', 'hp_solve', 'hp_apply']
    for nreg, ko, seeing in ((3, 4, 4.0), (1, 2, 4.0), (1, 4, 4.0), (2, 4, 4.0), (3, 2, 4.0), (3, 0, 4.0), (4, 4, 4.0), (1, 2, 2.0)):
        def one():
            sub.run(sci['img'], sci['rms'], sci['mask'], sci['wgt'], co.img, ref_rms, co.mask, seeing=seeing,
                    nreg_side=nreg, hotpants_kws=dict(ko=ko))
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        eng.timing(True)
        eng.timing_reset()
        one()
        torch.cuda.synchronize()
        eng.timing(False)
        kt = {}
        for nme in names:
            ms, cnt = eng.timing_read(nme)
            if cnt:
                kt[nme] = (round(ms, 2), cnt)
        i = sub.info
        print(f'nreg_side {nreg} ko {ko} seeing {seeing}: {1e3 * dt:.2f} ms; unknowns {i.ncoeff}, rounds {i.niter}, '
              f'stamps {i.nstamps_used}/{i.nstamps_total}, status {i.status}; {kt}', flush=True)


if __name__ == '__main__':
    main()
