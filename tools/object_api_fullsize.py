"""Developer tool: the ZUDS object API (the reference's call sites) at full size - N frames 3072 x 3072 as
IPAC-style FITS files on disk, `ReferenceImage.from_images` (scripts/makeref.py / dostack.py) and
`SingleEpochSubtraction.from_images` (scripts/dosub.py) with the reference's defaults, wall times per stage.
usage: object_api_fullsize.py [nframes] [dir]"""
import cProfile
import importlib
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    d = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix='zm_full_')
    z = importlib.import_module('zuds-pipeline_amd')
    s = importlib.import_module('zuds-pipeline_amd.synth')
    size = 3072
    base = s.ztf_wcs(size, size, tpv=True)
    rng = np.random.default_rng(1)
    xs, ys = rng.uniform(0, size, 1500), rng.uniform(0, size, 1500)
    fl = np.exp(rng.uniform(np.log(1e3), np.log(1e5), 1500))
    ra, dec = base.all_pix2world(xs, ys, 0)
    t0 = time.perf_counter()
    ims = []
    for i in range(n + 1):
        r = np.random.default_rng(100 + i)
        w = s.ztf_wcs(size, size, dx=r.uniform(-15, 15), dy=r.uniform(-15, 15), rot_deg=r.uniform(-0.1, 0.1), tpv=True)
        f = s.make_frame(size, size, 100 + i, w, star_sky=(ra, dec, fl), fwhm=2.0, nbad=3000)
        f['header']['SEEING'] = 2.0
        path = os.path.join(d, f'ztf_2020053{i:02d}_000651_zg_c03_o_q1_sciimg.fits')
        z.fits.write(path, f['img'], f['header'])
        z.fits.write(path.replace('sciimg', 'mskimg'), f['mask'].astype(np.int16), f['header'])
        z.fits.write(path.replace('.fits', '.weight.fits'), f['wgt'], f['header'])
        im = z.ScienceImage.from_file(path)
        im.mask_image = z.MaskImage.from_file(path.replace('sciimg', 'mskimg'))
        ims.append(im)
    print(f'{n + 1} frames written in {time.perf_counter() - t0:.1f} s', flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        pr = cProfile.Profile()
        pr.enable()
        ref = z.ReferenceImage.from_images(ims[:n], os.path.join(d, f'ref{rep}.000651_c03_q1_zg.fits'))
        pr.disable()
        t1 = time.perf_counter()
        print(f'ReferenceImage.from_images({n} frames): {t1 - t0:.2f} s = {n * size * size / 1e6 / (t1 - t0):.0f} Mpix/s', flush=True)
        if rep == 1:
            pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
    for rep in range(2):
        t0 = time.perf_counter()
        pr = cProfile.Profile()
        pr.enable()
        sub = z.SingleEpochSubtraction.from_images(ims[n], ref)
        pr.disable()
        t1 = time.perf_counter()
        print(f'SingleEpochSubtraction.from_images: {t1 - t0:.2f} s = {size * size / 1e6 / (t1 - t0):.1f} Mpix/s', flush=True)
        if rep == 1:
            pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
        os.remove(sub.local_path)


if __name__ == '__main__':
    main()
