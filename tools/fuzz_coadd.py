"""Developer tool: differential fuzzing of the fused kernel (summing and STACK forms) against the k_resample
path (ZM_COADD_FUSED=0) - random stack depths, frame sizes, rotations, scale changes, dithers, masks, combine
and mask-combine types, backgrounds on / off; every product must agree bit for bit.
usage: fuzz_coadd.py [ncases] [seed]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ncases, seed, eng=None, verbose=True):
    """Number of cases whose products differ between the two paths."""
    z = importlib.import_module('zuds-pipeline_amd')
    s = importlib.import_module('zuds-pipeline_amd.synth')
    eng = eng or z.Engine(0)
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        n = int(rng.integers(1, 8))
        onx, ony = int(rng.integers(90, 700)), int(rng.integers(90, 700))
        tpv = bool(rng.integers(0, 2))
        base = s.ztf_wcs(onx, ony, tpv=tpv)
        big_rot = rng.random() < 0.25
        frames = []
        for i in range(n):
            nx, ny = (onx, ony) if rng.random() < 0.6 else (int(rng.integers(80, 700)), int(rng.integers(80, 700)))
            rot = rng.uniform(-40, 40) if big_rot else rng.uniform(-0.3, 0.3)
            dx, dy = rng.uniform(-60, 60, 2) if rng.random() < 0.8 else rng.integers(-20, 20, 2).astype(float)
            w = s.ztf_wcs(nx, ny, dx=float(dx), dy=float(dy), rot_deg=float(rot), tpv=tpv)
            if rng.random() < 0.3:
                w.cd = np.asarray(w.cd) * float(rng.choice([0.5, 0.8, 1.3, 2.2]))
            f = s.make_frame(nx, ny, 10000 * seed + 10 * case + i, w, nstars=20, nbad=int(rng.integers(0, 400)))
            if rng.random() < 0.2:
                f['mask'] = None
            elif rng.random() < 0.2:
                f['mask'][ny // 3:ny // 3 + 9, nx // 4:nx // 4 + 30] |= 1 << 16
            if rng.random() < 0.15:
                f['wgt'] = None
            frames.append(f)
        kind = str(rng.choice(['WEIGHTED', 'AVERAGE', 'CLIPPED', 'MEDIAN']))
        mk = str(rng.choice(['AND', 'OR']))
        back = bool(rng.integers(0, 2))
        p = z.coadd_params(combine=kind, mask_combine=mk, subtract_back=back, rescale_weights=back,
                           back_size=int(rng.choice([32, 64, 128])))
        out = {}
        for mode in ('0', '1'):
            os.environ['ZM_COADD_FUSED'] = mode
            out[mode] = eng.coadd(frames, base, p, want_mask=True)
        os.environ.pop('ZM_COADD_FUSED', None)
        ok = all((x is None) == (y is None) and (x is None or np.array_equal(x, y, equal_nan=True))
                 for x, y in zip(out['0'], out['1']))
        if not ok:
            bad += 1
            diffs = [None if x is None else int((x != y).sum()) for x, y in zip(out['0'], out['1'])]
            print(f'case {case}: MISMATCH n={n} grid={onx}x{ony} tpv={tpv} big_rot={big_rot} {kind} {mk} back={back} '
                  f'sizes={[f["img"].shape for f in frames]} differing pixels {diffs}', flush=True)
        elif verbose and case % 25 == 0:
            print(f'case {case}: ok (n={n}, {onx}x{ony}, {kind}, {mk})', flush=True)
    return bad


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(ncases, seed)
    print(f'{ncases} cases, {bad} mismatches', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
