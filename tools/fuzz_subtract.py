"""Developer tool: differential fuzzing of the subtraction across the two forms of the solver - the same random
scene and parameters through one context that owns the GPU (k_chol_fused: 26 workgroups per region) and
through contexts declared to share it 3 and 9 ways (k_chol_tp: one workgroup per region): difference image,
noise image and the fit summary must agree bit for bit (the two forms run the same arithmetic).  Every case also
runs three scenes of its size and parameters (the case's own, two more; the data limits differ per scene) as ONE
batch (zm_subtract_batch: the job as a grid dimension of every launch of the fit) against the three one at a time.
usage: fuzz_subtract.py [ncases] [seed]"""
import importlib
import os
import sys

import numpy as np
from scipy.ndimage import gaussian_filter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scene(s, rng, nx, ny):
    nstars = int(nx * ny / 1000)
    ref = np.full((ny, nx), 150.0)
    s.add_stars(ref, rng.uniform(8, nx - 8, nstars), rng.uniform(8, ny - 8, nstars),
                np.exp(rng.uniform(np.log(3e3), np.log(8e4), nstars)), 2.0)
    sci = rng.uniform(0.7, 1.6) * gaussian_filter(ref, rng.uniform(0.6, 1.4), mode='nearest') + rng.uniform(0, 40)
    ref = ref + rng.normal(0, 0.5, ref.shape)
    sci = sci + rng.normal(0, 3.0, sci.shape)
    bpm = np.zeros((ny, nx), np.uint8)
    for _ in range(int(rng.integers(0, 12))):
        bx, by = rng.integers(5, nx - 8), rng.integers(5, ny - 8)
        bpm[by:by + 3, bx:bx + 3] = 1
    return (sci.astype(np.float32), np.full((ny, nx), 3.0, np.float32), ref.astype(np.float32),
            np.full((ny, nx), 0.5, np.float32), bpm)


def run(ncases, seed, verbose=True):
    z = importlib.import_module('zuds-pipeline_amd')
    s = importlib.import_module('zuds-pipeline_amd.synth')
    engines = []
    for share in (1, 3, 9):
        e = z.Engine(0)
        e.set_share(share)
        engines.append(e)
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        nx, ny = int(rng.integers(220, 720)), int(rng.integers(220, 720))
        hwk = int(rng.integers(2, 9))
        rss = int(rng.integers(hwk + 3, 22))
        if rng.random() < 0.12:
            # round 5: wide kernels / large substamps (the chunked form of k_hp_vectors, the wide convolutions)
            hwk = int(rng.integers(11, 21))
            rss = int(rng.integers(hwk + 4, 61))
            nx, ny = max(nx, 2 * (hwk + rss) + 150), max(ny, 2 * (hwk + rss) + 130)
        data = scene(s, rng, nx, ny)
        nreg = int(rng.integers(1, 4))
        kw = dict(r=float(hwk), rss=float(rss), nrx=nreg, nry=int(rng.integers(1, 4)),
                  nsx=int(rng.integers(2, 6)), nsy=int(rng.integers(2, 6)), ko=int(rng.integers(0, 4)),
                  bgo=int(rng.integers(0, 2)), tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
        if min(nx // kw['nrx'], ny // kw['nry']) < 2 * (hwk + int(kw['rss'])) + 8:
            kw['nrx'] = kw['nry'] = 1
        if kw['nrx'] * kw['nry'] * 2 > 26:
            kw['nry'] = 1
        try:
            outs = [e.subtract(*data, **kw) for e in engines]
        except z.ZMError as err:
            if verbose:
                print(f'case {case}: refused ({str(err)[:80]})', flush=True)
            continue
        ok = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) and outs[0][2] == o[2]
                 for o in outs[1:])
        # the batch: this scene and two more of its shape, different lower limits per job
        frames = [data, scene(s, rng, nx, ny), scene(s, rng, nx, ny)]
        hp = importlib.import_module('zuds-pipeline_amd.engine').hp_params
        plist = [hp(**dict(kw, il=-1e3 + 7 * k, tl=-1e3 - 3 * k)) for k in range(3)]
        lone = [engines[0].subtract(*frames[k], params=plist[k]) for k in range(3)]
        try:
            got = engines[0].subtract_batch(frames, params=plist)
        except z.ZMError as err:
            got = None
            ok = False
            print(f'case {case}: batch refused ({str(err)[:120]})', flush=True)
        if got is not None:
            ok = ok and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
                            for a, b in zip(lone, got))
        if not ok:
            bad += 1
            print(f'case {case}: MISMATCH {nx}x{ny} {kw} info {[o[2] for o in outs]}', flush=True)
        elif verbose and case % 10 == 0:
            i = outs[0][2]
            print(f'case {case}: ok ({nx}x{ny}, r {hwk}, {kw["nrx"]}x{kw["nry"]} regions, ko {kw["ko"]}: '
                  f'{i["ncoeff"]} unknowns, {i["niter"]} rounds, status {i["status"]})', flush=True)
    for e in engines:
        e.close()
    return bad


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(ncases, seed)
    print(f'{ncases} cases, {bad} mismatches', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
