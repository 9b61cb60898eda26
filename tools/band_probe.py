"""Developer probe: the CLIPPED / MEDIAN combine of one rank's row band of an 8-rank, 256-frame stack
(k_combine_wide<4>: BASELINE configs[3]) - time, HBM fraction, and a checksum of the products for A / B builds."""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
z = importlib.import_module('zuds-pipeline_amd')
eng = z.Engine(0)
depth, rows, size = 256, 384, 3072
g = torch.Generator(device='cuda')
g.manual_seed(9)
stack = torch.empty((depth, rows, size, 2), dtype=torch.float32, device='cuda')
stack[..., 0] = torch.randn((depth, rows, size), generator=g, device='cuda') * 5 + 100
stack[..., 1] = torch.where(torch.rand((depth, rows, size), generator=g, device='cuda') < 0.02, 0.0, 0.04)
o1 = torch.empty((rows, size), dtype=torch.float32, device='cuda')
o2 = torch.empty_like(o1)
bpx = rows * size
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for kind in ('CLIPPED', 'MEDIAN'):
    p = z.coadd_params(combine=kind, subtract_back=False, rescale_weights=False)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10):
            z._lib.check(eng.L.zm_combine_stack_dev(eng.ctx, depth, stack.data_ptr(), bpx, bpx, C.byref(p), o1.data_ptr(), o2.data_ptr()))
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
    byt = (8 * depth + 8) * bpx
    print(f'{kind}: {t * 1e3:.3f} ms = {byt / t / 1e9:.0f} GB/s = {byt / t / 8e12:.3f} of 8 TB/s; checksum {float(o1.double().sum()):.6f} {float(o2.double().sum()):.6f}')
