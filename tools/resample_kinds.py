"""Developer tool: one 3072 x 3072 frame through zm_resample_dev with every RESAMPLING_TYPE (device-resident,
image + weight + mask), ms per call."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import ctypes as C
    import torch
    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    size = 3072
    device = torch.device('cuda', 0)
    eng = z.Engine(0)
    base, frames = bench.make_device_frames(synth, torch, 1, size, 2000, device)
    f = frames[0]
    win, wout = z._lib.wcs_struct(f['wcs']), z._lib.wcs_struct(base)
    oi = torch.empty((size, size), dtype=torch.float32, device=device)
    ow = torch.empty_like(oi)
    om = torch.empty((size, size), dtype=torch.int32, device=device)
    for name, kind in (('LANCZOS3', z._lib.RESAMPLE['LANCZOS3']), ('BILINEAR', z._lib.RESAMPLE['BILINEAR']),
                       ('NEAREST', z._lib.RESAMPLE['NEAREST'])):
        for with_img in (True, False):
            def run():
                z._lib.check(eng.L.zm_resample_dev(eng.ctx, f['img'].data_ptr() if with_img else None,
                                                   f['wgt'].data_ptr() if with_img else None, f['mask'].data_ptr(),
                                                   C.byref(win), C.byref(wout), kind, 1.0,
                                                   oi.data_ptr() if with_img else None, ow.data_ptr() if with_img else None,
                                                   om.data_ptr()))
            run()
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            eng.synchronize()
            print(f'{name} {"image + weight + mask" if with_img else "mask only"}: {1e2 * (time.perf_counter() - t0):.3f} ms', flush=True)


if __name__ == '__main__':
    main()
