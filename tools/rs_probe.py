"""Developer probe: time of zm_median_mad2_async_dev on two 3072^2 frames with int32 masks (the bench's call), under
ZM_RS_THREADS / ZM_RS_GRID (shape of the histogram passes) and ZM_RS_BITS (0: masks read in every pass)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
z = importlib.import_module('zuds-pipeline_amd')
eng = z.Engine(0)
n = 3072 * 3072
g = torch.Generator(device='cuda').manual_seed(1)
a = (torch.randn(n, device='cuda', generator=g) * 4 + 150).float()
b = (torch.randn(n, device='cuda', generator=g) * 1 + 150).float()
ma = (torch.rand(n, device='cuda', generator=g) < 0.01).int()
mb = (torch.rand(n, device='cuda', generator=g) < 0.01).int()
out = torch.zeros(6, dtype=torch.float64, device='cuda')
L, check = eng.L, z._lib.check
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        check(L.zm_median_mad2_async_dev(eng.ctx, a.data_ptr(), ma.data_ptr(), b.data_ptr(), mb.data_ptr(), n, out.data_ptr()))
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 20
print(f'ZM_RS_THREADS={os.environ.get("ZM_RS_THREADS", "1024")} ZM_RS_GRID={os.environ.get("ZM_RS_GRID", "256")} '
      f'ZM_RS_BITS={os.environ.get("ZM_RS_BITS", "1")}: {t * 1e6:.1f} us per call', out.cpu().numpy()[:2])
