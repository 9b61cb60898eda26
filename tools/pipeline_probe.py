"""Developer probe: the bench step software-pipelined - the subtraction of step k (own context and
stream, Cholesky on 1 / share of the CUs) beside the coadd of step k + 1.
usage: pipeline_probe.py share [steps]"""
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
    share = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    size, nfr = 3072, 32
    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    dev = importlib.import_module('zuds-pipeline_amd.device')
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    check = z._lib.check
    eng_c, eng_s = z.Engine(0), z.Engine(0)
    eng_s.set_share(share)
    base, frames = bench.make_device_frames(synth, torch, nfr + 1, size, 2000, device)
    sci = frames.pop()
    g = torch.Generator(device='cpu')
    g.manual_seed(77)
    bx = torch.randint(2, size - 2, (300,), generator=g)
    by = torch.randint(2, size - 2, (300,), generator=g)
    smask = torch.zeros((size, size), dtype=torch.int32)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            smask[by + dy, bx + dx] = 256
    sci['mask'] = smask.to(device)
    sci['wgt'] = torch.where(sci['mask'] != 0, 0.0, float(sci['wgt'].max())).to(torch.float32)
    sci['rms'] = torch.where(sci['wgt'] > 0, 1.0 / torch.sqrt(sci['wgt'].clamp_min(1e-20)), float(np.sqrt(50000.0))).to(torch.float32)
    params = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True)
    dframes = dev.DeviceFrames(frames, device)
    co = dev.DeviceCoadd(base, params, device=0, engine=eng_c, want_mask=True)
    sub = dev.DeviceSubtraction(sci['wcs'], base, device=0, engine=eng_s)
    A, B = co.stream, sub.stream
    npx = size * size
    L = eng_c.L
    big = float(np.sqrt(50000.0))
    snap = [dict(img=torch.empty_like(co.img), rms=torch.empty_like(co.img), mask=torch.empty_like(co.mask)) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [None, None]

    def enqueue_coadd(k):
        s = snap[k & 1]
        if freed[k & 1] is not None:
            A.wait_event(freed[k & 1])
        co.run(dframes)
        with torch.cuda.stream(A):
            check(L.zm_mask_flag_dev(eng_c.ctx, co.mask.data_ptr(), co.mask_wgt.data_ptr(), 0.0, 1 << 16, npx))
            check(L.zm_add_scalar_dev(eng_c.ctx, co.img.data_ptr(), 150.0, npx))
            check(L.zm_rms_from_weight_dev(eng_c.ctx, co.wgt.data_ptr(), None, npx, big, s['rms'].data_ptr()))
            s['img'].copy_(co.img)
            s['mask'].copy_(co.mask)
            ready[k & 1].record(A)

    def subtract(k):
        s = snap[k & 1]
        B.wait_event(ready[k & 1])
        sub.run(sci['img'], sci['rms'], sci['mask'], sci['wgt'], s['img'], s['rms'], s['mask'], seeing=4.0, nreg_side=3)
        ev = torch.cuda.Event()
        ev.record(B)
        freed[k & 1] = ev

    state = {'k': 0}
    enqueue_coadd(0)

    def step():
        k = state['k']
        enqueue_coadd(k + 1)
        subtract(k)
        state['k'] = k + 1

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'share {share}: {1e3 * dt / steps:.2f} ms per step pipelined, {(nfr + 1) * npx / 1e6 * steps / dt:.0f} Mpix/s, '
          f'status {sub.info.status}, stamps {sub.info.nstamps_used}', flush=True)


if __name__ == '__main__':
    main()
