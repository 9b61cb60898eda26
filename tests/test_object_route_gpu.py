"""The device route under the object API (objdev: raw FITS blocks -> HBM -> DeviceCoadd / DeviceSubtraction ->
encoded products, every file written once) against the host-pointer route it replaces (`ZM_OBJECT_API=host`:
numpy decode, zm_coadd / zm_subtract on host arrays, products written, re-read and saved again) for the
reference's own call sites - `ReferenceImage / ScienceCoadd.from_images` (zuds/coadd.py:25-236,
scripts/makeref.py:85, scripts/dostack.py:59) and `SingleEpochSubtraction.from_images`
(zuds/subtraction.py:57-226, scripts/dosub.py:97): the same pixels, the same header cards in the same order,
the same object state; inputs taken from files, from memory (loaded, modified) and mixed."""
import os

import numpy as np
import pytest

from test_scripts_gpu import _scene
from util import pkg, synth

pytestmark = pytest.mark.gpu


def route(name, fn):
    old = os.environ.get('ZM_OBJECT_API')
    os.environ['ZM_OBJECT_API'] = name
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop('ZM_OBJECT_API', None)
        else:
            os.environ['ZM_OBJECT_API'] = old


def same_file(z, a, b):
    da, ha, ca = z.fits.read(a)
    db, hb, cb = z.fits.read(b)
    assert da.dtype == db.dtype and np.array_equal(da, db, equal_nan=True), (a, int((da != db).sum()))
    assert list(ha.items()) == list(hb.items()), (a, [k for k in ha if ha.get(k) != hb.get(k)], set(ha) ^ set(hb))
    assert ca == cb, a


def with_weights(z, ims):
    for im in ims:                       # `.weight.fits` siblings (1 / rms^2, 0 on bad pixels) as the pipeline keeps them
        w = im.weight_image
        assert w.ismapped
        rms = im.local_path.replace('.fits', '.rms.fits')      # (the mesh rms map the weights were derived from)
        if os.path.exists(rms):
            os.remove(rms)


def reopen(z, paths):
    out = []
    for p in paths:
        im = z.ScienceImage.from_file(p)
        im.mask_image = z.MaskImage.from_file(p.replace('sciimg', 'mskimg'))
        out.append(im)
    return out


@pytest.mark.parametrize('kws', [None, {'COMBINE_TYPE': 'WEIGHTED'}])
def test_coadd_from_images_device_route_equals_host_route(tmp_path, engine, kws):
    z, s = pkg(), synth()
    d = str(tmp_path)
    ims, paths = _scene(z, s, d, 640, 600, 4, 5100, '202001')
    with_weights(z, ims)
    outs = {}
    for name in ('host', 'device'):
        fresh = reopen(z, paths)                     # nothing in memory: the device route reads the files
        out = os.path.join(d, f'ref_{name}.000651_c03_q1_zg.fits')
        outs[name] = route(name, lambda: z.ReferenceImage.from_images(fresh, out, sci_swarp_kws=kws))
    for sfx in ('.fits', '.weight.fits', '.mask.fits'):
        same_file(z, outs['host'].local_path.replace('.fits', sfx), outs['device'].local_path.replace('.fits', sfx))
    h, v = outs['host'], outs['device']
    assert list(h.header.items()) == list(v.header.items()) and 'SEEING' in v.header and v.header['FIELD'] == 651
    assert list(h.mask_image.header.items()) == list(v.mask_image.header.items())
    assert (h.field, h.ccdid, h.qid, h.fid) == (v.field, v.ccdid, v.qid, v.fid)
    assert (v.mask_image.data & (1 << 16)).any()
    # inputs in memory (loaded and modified) and mixed with file-mapped ones: the pixels of the object count
    mixed = {}
    for name in ('host', 'device'):
        fresh = reopen(z, paths)
        fresh[1].load()
        fresh[1].data = fresh[1].data + np.float32(3.0)
        fresh[2].mask_image.load()
        fresh[2].mask_image.data[100:120, 50:90] |= 2
        _ = fresh[3].weight_image.data
        out = os.path.join(d, f'mix_{name}.000651_c03_q1_zg.fits')
        mixed[name] = route(name, lambda: z.ScienceCoadd.from_images(fresh, out, sci_swarp_kws=kws, calculate_seeing=False,
                                                                     addbkg=False, set_date=False))
    for sfx in ('.fits', '.weight.fits', '.mask.fits'):
        same_file(z, mixed['host'].local_path.replace('.fits', sfx), mixed['device'].local_path.replace('.fits', sfx))
    assert not np.array_equal(z.fits.read(mixed['device'].local_path)[0], z.fits.read(outs['device'].local_path)[0])


def test_subtraction_from_images_device_route_equals_host_route(tmp_path, engine):
    z, s = pkg(), synth()
    d = str(tmp_path)
    refims, rpaths = _scene(z, s, d, 640, 600, 3, 5300, '201912', fwhm=2.0)
    with_weights(z, refims)
    refname = os.path.join(d, 'ref.000651_c03_q1_zg.fits')
    route('device', lambda: z.ReferenceImage.from_images(reopen(z, rpaths), refname, sci_swarp_kws={'COMBINE_TYPE': 'WEIGHTED'}))
    sims, spaths = _scene(z, s, d, 640, 600, 2, 5400, '202003', fwhm=2.6,
                          extra=lambda i: {'SEEING': 2.6, 'SATURATE': 40000.0})
    with_weights(z, sims)
    for p in spaths:
        res = {}
        for name in ('host', 'device'):
            sci = reopen(z, [p])[0]
            assert hasattr(sci, '_weightimg') and not hasattr(sci, '_rmsimg')
            ref = z.ReferenceImage.from_file(refname, load_others=False)
            ref.mask_image = z.MaskImage.from_file(refname.replace('.fits', '.mask.fits'))
            ref._weightimg = z.FITSImage.from_file(refname.replace('.fits', '.weight.fits'))
            sub = route(name, lambda: z.SingleEpochSubtraction.from_images(sci, ref, nreg_side=1, tmpdir=d))
            res[name] = sub
            for sfx in ('.fits', '.rms.fits', '.mask.fits'):
                os.replace(sub.local_path.replace('.fits', sfx), sub.local_path.replace('.fits', f'.{name}{sfx}'))
            for f in os.listdir(d):                  # (the host route leaves rms siblings of its inputs behind)
                if f.endswith('.rms.fits') and not f.startswith('sub.'):
                    os.remove(os.path.join(d, f))
        out = res['device'].local_path
        for sfx in ('.fits', '.rms.fits', '.mask.fits'):
            same_file(z, out.replace('.fits', f'.host{sfx}'), out.replace('.fits', f'.device{sfx}'))
        h, v = res['host'], res['device']
        assert list(h.header.items()) == list(v.header.items())
        assert {k: h.hotpants_info[k] for k in h.hotpants_info} == {k: v.hotpants_info[k] for k in v.hotpants_info}
        assert v.header['ZMSTATUS'] == 0 and v.hotpants_info['ncoeff'] > 0 and v.header['SEEING'] == 2.6
