"""Developer tool: time of the CLIPPED / MEDIAN combine kernels against the stack depth (same number of samples)."""
import importlib, sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import torch
z = importlib.import_module('zuds-pipeline_amd')
eng = z.Engine(0)
L = eng.L
check = z._lib.check
for n, rows in ((32, 3072), (64, 1536), (96, 1024), (128, 768), (256, 384), (512, 192)):
    nx = 3072
    npx = rows * nx
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    stack = torch.empty((n, rows, nx, 2), dtype=torch.float32, device='cuda')
    stack[..., 0] = torch.randn((n, rows, nx), generator=g, device='cuda') * 5 + 100
    stack[..., 1] = 0.04
    stack[..., 1][torch.rand((n, rows, nx), generator=g, device='cuda') < 0.02] = 0
    img = torch.empty((rows, nx), dtype=torch.float32, device='cuda'); wgt = torch.empty_like(img)
    for kind in ('CLIPPED', 'MEDIAN'):
        p = z.coadd_params(combine=kind)
        def run():
            check(L.zm_combine_stack_dev(eng.ctx, n, stack.data_ptr(), npx, npx, C.byref(p), img.data_ptr(), wgt.data_ptr()))
        run(); torch.cuda.synchronize(); eng.synchronize() if hasattr(eng, 'synchronize') else None
        L.zm_ctx_synchronize(eng.ctx)
        t0 = time.perf_counter()
        for _ in range(3): run()
        L.zm_ctx_synchronize(eng.ctx)
        dt = (time.perf_counter() - t0) / 3
        print(f'n {n} rows {rows} {kind}: {1e3*dt:.2f} ms = {n*npx*8/dt/1e9:.0f} GB/s of samples', flush=True)
