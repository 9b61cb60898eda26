"""Developer probe: what bounds fitsring.FITSRing on this box.  Reads alone (page cache -> pinned memory) against the
number of reader threads, reads + H2D + decode, and the product side (encode + D2H + write) against writers.
usage: python3 tools/ring_probe.py [nfiles] [size]"""
import importlib
import os
import sys
import tempfile
import shutil
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 66
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
    z = importlib.import_module('zuds-pipeline_amd')
    ringmod = importlib.import_module('zuds-pipeline_amd.fitsring')
    d = tempfile.mkdtemp(prefix='zmring_', dir=os.environ.get('TMPDIR') or None)
    try:
        rng = np.random.default_rng(1)
        a = rng.normal(0, 1, (size, size)).astype(np.float32)
        m = rng.integers(0, 300, (size, size)).astype(np.int16)
        wanted = []
        for i in range(nfiles):
            p = os.path.join(d, f'p{i:03d}.fits')
            z.fits.write(p, m if i % 3 == 2 else a, {'MAGZP': 26.0})
            wanted.append((p, 'mask' if i % 3 == 2 else 'f32'))
        nbytes = sum(os.path.getsize(p) for p, _ in wanted)
        print(f'{nfiles} files, {nbytes / 1e9:.2f} GB, cores {len(os.sched_getaffinity(0))}', flush=True)
        for nr in (8, 12, 16):
            for pinned in (3 << 29,):
                ring = ringmod.FITSRing(0, nreaders=nr, nwriters=2, pinned_in=pinned)
                ring.prefetch(wanted).result()
                torch.cuda.synchronize()
                # reads alone: the reader pool without the feeder
                t0 = time.perf_counter()
                futs = [ring._readers.submit(ring._read, p, False) for p, _ in wanted]
                for f in futs:
                    ring._pin_in.put(f.result()[0])
                t_read = time.perf_counter() - t0
                t0 = time.perf_counter()
                for _ in range(3):
                    t = ring.prefetch(wanted)
                    out = t.result()
                    del out
                torch.cuda.synchronize()
                t_all = (time.perf_counter() - t0) / 3
                print(f'readers {nr:2d} pinned {pinned / 2**30:.1f} GiB: reads alone {nbytes / t_read / 1e9:6.1f} GB/s, '
                      f'read + H2D + decode {nbytes / t_all / 1e9:6.1f} GB/s ({1e3 * t_all:.1f} ms)', flush=True)
                ring.close()
        nprod = int(os.environ.get('ZM_PROBE_PRODUCTS', '48'))
        prods = [torch.randn((size, size), device='cuda') for _ in range(6)]
        for nw in (2, 3, 4, 5, 6, 8, 4, 6):
            ring = ringmod.FITSRing(0, nreaders=2, nwriters=nw)
            for k in range(nprod):
                ring.save(os.path.join(d, f'o{k}.fits'), prods[k % 6])
            ring.flush()
            for k in range(nprod):
                os.remove(os.path.join(d, f'o{k}.fits'))
            t0 = time.perf_counter()
            for k in range(nprod):
                ring.save(os.path.join(d, f'o{k}.fits'), prods[k % 6])
            ring.flush()
            dt = time.perf_counter() - t0
            for k in range(nprod):
                os.remove(os.path.join(d, f'o{k}.fits'))
            print(f'writers {nw:2d}: {nprod} new files ({nprod * size * size * 4 / 1e9:.2f} GB) in {1e3 * dt:.1f} ms = '
                  f'{nprod * size * size * 4 / dt / 1e9:.1f} GB/s', flush=True)
            ring.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
