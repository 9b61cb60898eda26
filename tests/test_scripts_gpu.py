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
    first = {out: [open(out.replace('.fits', sfx), 'rb').read() for sfx in ('.fits', '.rms.fits', '.mask.fits', '.phot.txt')]
             for out in done}
    for out in done:
        for suffix in ('.fits', '.rms.fits', '.mask.fits', '.phot.txt'):
            os.remove(out.replace('.fits', suffix))
    # ... and with the three kernel fits as one batch of launches (--fit-batch): the same files, byte for byte
    again = script.main([os.path.join(d, 'images.txt'), refname, os.path.join(d, 'positions.txt'),
                         '--jobs', '1', '--fit-batch', '3', '--nreg-side', '1'])
    assert again == done
    for out in done:
        for k, sfx in enumerate(('.fits', '.rms.fits', '.mask.fits', '.phot.txt')):
            assert open(out.replace('.fits', sfx), 'rb').read() == first[out][k], (out, sfx)
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


# ---------------------------------------------------------------------------------------------
# dostack.py / dosub.py / makeref.py: the reference's other drivers on the hot path
# (scripts/dostack.py:59, scripts/dosub.py:97, scripts/makeref.py:85).  Each is run on job files
# and its products are compared with the object API called directly.
def _scene(z, s, d, nx, ny, n, seed, prefix, fwhm=2.2, extra=None):
    """n dithered frames of one star field as IPAC-style files (sciimg + mskimg); -> (objects, paths)"""
    base = s.ztf_wcs(nx, ny, tpv=True)
    rng = np.random.default_rng(seed)
    nst = int(nx * ny / 2500)
    xs, ys = rng.uniform(-20, nx + 20, nst), rng.uniform(-20, ny + 20, nst)
    fl = np.exp(rng.uniform(np.log(3e3), np.log(8e4), nst))
    ra, dec = base.all_pix2world(xs, ys, 0)
    ims, paths = [], []
    for i in range(n):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-5, 5), dy=rng.uniform(-5, 5), rot_deg=rng.uniform(-0.05, 0.05))
        f = s.make_frame(nx, ny, seed + 1 + i, w, star_sky=(ra, dec, fl), fwhm=fwhm, sky=150.0 + 7 * i,
                         noise=4.0, bad_block=(60 + 37 * i, 90 + 23 * i, 4))
        hdr = f['header']
        hdr['OBSJD'] = 2458000.5 + hdr['OBSMJD'] - 58000.0 + i
        hdr.update(extra(i) if extra else {})
        path = os.path.join(d, f'ztf_{prefix}{i:02d}_000651_zg_c03_o_q1_sciimg.fits')
        z.fits.write(path, f['img'], hdr)
        z.fits.write(path.replace('sciimg', 'mskimg'), f['mask'].astype(np.int16), hdr)
        im = z.ScienceImage.from_file(path)
        im.mask_image = z.MaskImage.from_file(path.replace('sciimg', 'mskimg'))
        ims.append(im)
        paths.append(path)
    return ims, paths


def _products(z, path):
    return [z.fits.read(path.replace('.fits', sfx))[0] for sfx in ('.fits', '.weight.fits', '.mask.fits')]


def test_dostack_products_equal_from_images(tmp_path, engine):
    """scripts/dostack.py on a two-job file: names, skip-if-exists and pixels."""
    import pandas as pd
    z, s = pkg(), synth()
    d = str(tmp_path)
    _, p1 = _scene(z, s, d, 640, 600, 3, 4100, '202001')
    _, p2 = _scene(z, s, d, 640, 600, 4, 4200, '202002')
    pd.DataFrame({'target': [';'.join(p1), ';'.join(p2)], 'left': ['20200101', '20200201'],
                  'right': ['20200108', '20200208']}).to_csv(os.path.join(d, 'jobs.csv'), index=False)
    script = load_script('dostack')
    assert script.main([os.path.join(d, 'jobs.csv'), '--tmpdir', d]) == 0
    names = [os.path.join(d, f'000651_c03_q1_zg_{a}_{b}.coadd.fits')
             for a, b in (('20200101', '20200108'), ('20200201', '20200208'))]
    got = []
    for nm in names:
        assert os.path.exists(nm) and os.path.exists(nm.replace('.fits', '.mask.fits')), nm
        got.append(_products(z, nm))
    stamp = [os.path.getmtime(nm) for nm in names]
    assert script.main([os.path.join(d, 'jobs.csv'), '--tmpdir', d]) == 0       # resumes: nothing redone
    assert [os.path.getmtime(nm) for nm in names] == stamp
    for nm, paths, g in zip(names, (p1, p2), got):
        ims = script.load_inputs(';'.join(paths))
        want = z.ScienceCoadd.from_images(ims, outfile_name=nm.replace('.coadd.', '.direct.'), tmpdir=d)
        w = _products(z, want.local_path)
        for a, b, what in zip(g, w, ('coadd', 'weight', 'mask')):
            assert np.array_equal(a, b), (nm, what)
        hdr = z.fits.read(nm)[1]
        assert 'SEEING' in hdr and hdr['FIELD'] == 651       # calculate_seeing=True ran offline


def test_dosub_products_equal_from_images(tmp_path, engine):
    """scripts/dosub.py do_one: the mesh-rms branch (no weight / rms sibling, dosub.py:42-44), the
    checkpoint by name, and products equal to SingleEpochSubtraction.from_images."""
    z, s = pkg(), synth()
    d = str(tmp_path)
    refims, _ = _scene(z, s, d, 640, 600, 3, 4300, '201912', fwhm=2.0)
    refname = os.path.join(d, 'ref.000651_c03_q1_zg.fits')
    z.ReferenceImage.from_images(refims, refname, sci_swarp_kws={'COMBINE_TYPE': 'WEIGHTED'})
    _, spaths = _scene(z, s, d, 640, 600, 2, 4400, '202003', fwhm=2.6)
    script = load_script('dosub')
    subs = [script.do_one(p, z.ScienceImage, z.SingleEpochSubtraction, refname, tmpdir=d) for p in spaths]
    with pytest.raises(script.PredecessorError):
        script.do_one(spaths[0], z.ScienceImage, z.SingleEpochSubtraction, refname, tmpdir=d)
    for p, sub in zip(spaths, subs):
        out = z.sub_name(p, refname)
        assert sub.local_path == out and os.path.exists(out.replace('.fits', '.rms.fits'))
        got = [z.fits.read(out.replace('.fits', sfx))[0] for sfx in ('.fits', '.rms.fits', '.mask.fits')]
        hdr = z.fits.read(out)[1]
        assert hdr['ZMSTATUS'] == 0 and hdr['ZMUNSOLV'] == 0 and hdr['ZMRETRY'] == 0 and 'KSUM00' in hdr
        for sfx in ('.fits', '.rms.fits', '.mask.fits'):
            os.remove(out.replace('.fits', sfx))
        sci = z.ScienceImage.from_file(p)
        sci.mask_image = z.MaskImage.from_file(p.replace('sciimg', 'mskimg'))
        _ = sci.rms_image
        ref = z.ReferenceImage.from_file(refname, load_others=False)
        ref.mask_image = z.MaskImage.from_file(refname.replace('.fits', '.mask.fits'))
        ref._weightimg = z.FITSImage.from_file(refname.replace('.fits', '.weight.fits'))
        want = z.SingleEpochSubtraction.from_images(sci, ref, tmpdir=d)
        assert np.array_equal(want.data, got[0])
        assert np.array_equal(want.rms_image.data, got[1])
        assert np.array_equal(want.mask_image.data, got[2])


def test_makeref_selects_like_the_reference_and_coadds(tmp_path, engine):
    """scripts/makeref.py: the header cuts of scripts/makeref.py:57-78 (date window, seeing, limiting
    magnitude, infobits, the deepest MAX_FRAMES, at least MIN_FRAMES), the product name, and pixels
    equal to ReferenceImage.from_images of the selected frames."""
    import pandas as pd
    z, s = pkg(), synth()
    d = str(tmp_path / 'field')
    os.makedirs(d)
    maglim = [20.5, 20.9, 20.1, 20.7, 19.0, 20.8, 20.6, 20.4]

    def cards(i):
        return {'MAGLIM': maglim[i], 'INFOBITS': 1 if i == 1 else 0, 'SEEING': 2.9 if i == 2 else 2.1}
    ims, paths = _scene(z, s, d, 512, 480, 8, 4500, '202001', extra=cards)
    # frame 7 falls out of the date window, 1 has infobits, 2 bad seeing, 4 too shallow
    jd = [float(im.header['OBSJD']) for im in ims]
    lo = pd.to_datetime(jd[0] - 0.5, unit='D', origin='julian')
    hi = pd.to_datetime(jd[6] + 0.5, unit='D', origin='julian')
    with open(os.path.join(str(tmp_path), 'dirs.txt'), 'w') as fh:
        fh.write(d + '\n')
    script = load_script('makeref')
    top = script.select(d, lo, hi)
    assert [os.path.basename(t.local_path) for t in top] == [os.path.basename(paths[i]) for i in (5, 3, 6, 0)]
    args = [os.path.join(str(tmp_path), 'dirs.txt'), str(lo), str(hi), 'v9']
    assert script.main(args) == []                         # 4 < 14 frames: skipped, like the reference
    script.MIN_FRAMES = 3
    script.MAX_FRAMES = 3                                  # "the very best images"
    made = script.main(args)
    want_name = os.path.join(d, 'ref.000651_c03_q1_zg.v9.fits')
    assert made == [want_name]
    assert script.main(args) == []                         # exists: skipped
    got = _products(z, want_name)
    sel = script.select(d, lo, hi)
    assert len(sel) == 3
    direct = z.ReferenceImage.from_images(sel, os.path.join(d, 'direct.fits'), data_product=True, tmpdir=str(tmp_path))
    for a, b, what in zip(got, _products(z, direct.local_path), ('coadd', 'weight', 'mask')):
        assert np.array_equal(a, b), what
