"""scripts/donightly.py (the reference's scripts/donightly.py -> dosub.do_one, then
scripts/dophot.py) end to end on synthetic files: FITS in through the device decoder, J
subtractions in flight, products on disk with the reference's names - identical to what
``SingleEpochSubtraction.from_images`` writes for the same frames, photometry identical to
``raw_aperture_photometry`` on those files."""
import importlib.util
import os

import numpy as np
import pytest

from util import pkg, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'scripts', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_donightly_products_equal_the_object_api(tmp_path, engine):
    z, s = pkg(), synth()
    d = str(tmp_path)
    nx = ny = 1024
    base = s.ztf_wcs(nx, ny, tpv=True)
    rng = np.random.default_rng(77)
    xs, ys = rng.uniform(-20, nx + 20, 500), rng.uniform(-20, ny + 20, 500)
    fl = np.exp(rng.uniform(np.log(3e3), np.log(8e4), 500))
    ra, dec = base.all_pix2world(xs, ys, 0)

    def write(name, f, seeing):
        path = os.path.join(d, name)
        f['header']['SEEING'] = seeing
        f['header']['OBSJD'] = 2458000.5 + f['header']['OBSMJD'] - 58000.0
        z.fits.write(path, f['img'], f['header'])
        z.fits.write(path.replace('sciimg', 'mskimg'), f['mask'].astype(np.int16), f['header'])
        z.fits.write(path.replace('.fits', '.weight.fits'), f['wgt'], f['header'])
        im = z.ScienceImage.from_file(path)
        im.mask_image = z.MaskImage.from_file(path.replace('sciimg', 'mskimg'))
        return im

    refims = []
    for i in range(3):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-3, 3), dy=rng.uniform(-3, 3), rot_deg=rng.uniform(-0.03, 0.03))
        f = s.make_frame(nx, ny, 700 + i, w, star_sky=(ra, dec, fl), fwhm=2.0, noise=3.0,
                         bad_block=(100 + 200 * i, 300, 4))
        refims.append(write(f'ztf_2020010{i}_000651_zg_c03_o_q1_sciimg.fits', f, 2.0))
    refname = os.path.join(d, 'ref.000651_c03_q1_zg.fits')
    ref = z.ReferenceImage.from_images(refims, refname, sci_swarp_kws={'COMBINE_TYPE': 'WEIGHTED'})
    scis, names = [], []
    for i in range(3):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-5, 5), dy=rng.uniform(-5, 5), rot_deg=rng.uniform(-0.05, 0.05))
        f = s.make_frame(nx, ny, 800 + i, w, star_sky=(ra, dec, fl), fwhm=2.6, sky=170.0 + 15 * i,
                         bad_block=(150 + 250 * i, 600, 4))
        nm = f'ztf_2020020{i}_000651_zg_c03_o_q1_sciimg.fits'
        scis.append(write(nm, f, 2.6))
        names.append(os.path.join(d, nm))
    with open(os.path.join(d, 'images.txt'), 'w') as fh:
        fh.write('\n'.join(names) + '\n')
    pra, pdec = base.all_pix2world(rng.uniform(40, nx - 40, 50), rng.uniform(40, ny - 40, 50), 0)
    np.savetxt(os.path.join(d, 'positions.txt'), np.column_stack([pra, pdec]), fmt='%.10f')

    script = load_script('donightly')
    done = script.main([os.path.join(d, 'images.txt'), refname, os.path.join(d, 'positions.txt'),
                        '--jobs', '3', '--nreg-side', '1'])
    assert len(done) == 3
    # a second run finds its products and does nothing
    assert script.main([os.path.join(d, 'images.txt'), refname, '--jobs', '2', '--nreg-side', '1']) == []
    got = {}
    for out in done:
        assert os.path.basename(out) == os.path.basename(z.sub_name(names[done.index(out)], refname))
        got[out] = [z.fits.read(out)[0], z.fits.read(out.replace('.fits', '.rms.fits'))[0],
                    z.fits.read(out.replace('.fits', '.mask.fits'))[0],
                    np.loadtxt(out.replace('.fits', '.phot.txt'))]
        t = z.raw_aperture_photometry(out, out.replace('.fits', '.rms.fits'), out.replace('.fits', '.mask.fits'),
                                      pra, pdec)
        tab = got[out][3]
        np.testing.assert_allclose(tab[:, 2], t['flux'], rtol=2e-6, atol=1e-4)       # text round trip
        np.testing.assert_allclose(tab[:, 3], t['fluxerr'], rtol=2e-6)
        assert np.array_equal(tab[:, 4].astype(np.int64), t['flags'])
        for suffix in ('.fits', '.rms.fits', '.mask.fits', '.phot.txt'):
            os.remove(out.replace('.fits', suffix))
    # the same subtractions through the object API
    for im, out in zip(scis, done):
        sub = z.SingleEpochSubtraction.from_images(im, ref, nreg_side=1)
        assert sub.local_path == out
        assert np.array_equal(sub.data, got[out][0])
        assert np.array_equal(sub.rms_image.data, got[out][1])
        assert np.array_equal(sub.mask_image.data, got[out][2])
        assert sub.hotpants_info['ncoeff'] == 722 and sub.hotpants_info['status'] == 0
