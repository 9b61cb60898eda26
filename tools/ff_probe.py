#!/usr/bin/env python
"""Developer probe of the fused coadd kernel: the bench's 32-frame 3072^2 stack (cheap synthetic
pixels: noise, no stars), timed per scope for a list of ZM_FF_DBG ablations.

    python tools/ff_probe.py [--frames 32] [--size 3072] [--dbg 0,1,2,4] [--no-mask] [--reps 6]

ZM_FF_DBG bits (k_coadd_fused, developer only): 1 = no pixel work, 2 = no prep + LDS store of the
staged box, 4 = no staging loads.  The products are garbage with any bit set."""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--size', type=int, default=3072)
    ap.add_argument('--dbg', default='0')
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--no-mask', action='store_true')
    ap.add_argument('--combine', default='WEIGHTED')
    ap.add_argument('--rot', type=float, default=0.1)
    a = ap.parse_args()
    import torch
    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    dev = importlib.import_module('zuds-pipeline_amd.device')
    device = torch.device('cuda', 0)
    eng = z.Engine(0)
    n = a.size
    base = synth.ztf_wcs(n, n, tpv=True)
    g = torch.Generator(device=device)
    frames = []
    for i in range(a.frames):
        r = np.random.default_rng(2000 + i)
        w = synth.ztf_wcs(n, n, dx=r.uniform(-15, 15), dy=r.uniform(-15, 15), rot_deg=r.uniform(-a.rot, a.rot), tpv=True)
        g.manual_seed(2000 + i)
        sky = r.uniform(100, 300)
        img = sky + torch.randn((n, n), generator=g, device=device) * float(np.sqrt(sky / 6.2))
        bad = torch.rand((n, n), generator=g, device=device) < 1e-3
        frames.append(dict(img=img.float(), wgt=torch.where(bad, 0.0, 6.2 / sky).float(),
                           mask=None if a.no_mask else torch.where(bad, 256, 0).to(torch.int32), wcs=w,
                           flxscale=10 ** (-0.4 * (r.uniform(25.8, 26.6) - 25.0))))
    params = z.coadd_params(combine=a.combine, subtract_back=True, rescale_weights=True)
    dfr = dev.DeviceFrames(frames, device)
    co = dev.DeviceCoadd(base, params, device=0, engine=eng, want_mask=not a.no_mask)
    scopes = ['coadd_fused', 'mask_box', 'bk_rows', 'mesh_stats', 'mesh_filter', 'lattice', 'prep', 'combine']
    for dbg in a.dbg.split(','):
        os.environ['ZM_FF_DBG'] = dbg
        for _ in range(3):
            co.run(dfr)
        torch.cuda.synchronize()
        eng.timing(True)
        eng.timing_reset()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(co.stream):
            t0.record()
            for _ in range(a.reps):
                co.run(dfr)
            t1.record()
        torch.cuda.synchronize()
        eng.timing(False)
        out = {}
        for s in scopes:
            ms, cnt = eng.timing_read(s)
            if cnt:
                out[s] = round(ms / a.reps, 3)
        print(f'ZM_FF_DBG={dbg}: leg {t0.elapsed_time(t1) / a.reps:.3f} ms (all scopes timed)', out, flush=True)


if __name__ == '__main__':
    main()
