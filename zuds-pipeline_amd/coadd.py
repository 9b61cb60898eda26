"""Coadd products (``zuds/coadd.py``): ``Coadd / ReferenceImage /
ScienceCoadd.from_images`` with the reference's signature, defaults, file
products and bookkeeping; the two SWarp processes are replaced by one
``zm_coadd`` call (science + mask coadd share the resampling lattice)."""
import os
import uuid
import warnings
from pathlib import Path

import numpy as np

from .constants import BKG_VAL, GROUP_PROPERTIES
from .image import CalibratableImage, FITSImage
from .utils import ensure_images_have_the_same_properties, get_time

__all__ = ['Coadd', 'ReferenceImage', 'ScienceCoadd']


def _coadd_from_images(cls, images, outname=None, data_product=False, tmpdir='/tmp',
                       sci_swarp_kws=None, mask_swarp_kws=None, calculate_seeing=True,
                       addbkg=True, enforce_partition=True, solve_astrometry=False,
                       swarp_zp_key='MAGZP', scamp_kws=None, set_date=True,
                       outfile_name=None, nthreads=None, **ignored):
    """Make a coadd from a bunch of input images (``zuds/coadd.py:25-236``).

    ``outfile_name`` and ``nthreads`` are the keyword spellings the driver
    scripts use (``scripts/dostack.py:60-63``, ``scripts/makeref.py:87``); other
    unknown keywords are ignored like the reference's SWarp pass-through
    would."""
    from . import fits as _fits
    from .engine import coadd_params, get_engine
    from .mask import MaskImage
    from .swarp import (MSK_COPY_KEYWORDS, SCI_COPY_KEYWORDS, output_header,
                        prepare_swarp_mask, prepare_swarp_sci)

    if outname is None:
        outname = outfile_name
    if outname is None:
        raise TypeError('from_images() missing required argument: "outname"')
    outname = str(outname)
    images = np.atleast_1d(images)
    mskoutname = outname.replace('.fits', '.mask.fits')
    if solve_astrometry:
        raise NotImplementedError('solve_astrometry needs SCAMP (zuds/scamp.py), which is '
                                  'outside the coadd / subtraction path')
    if enforce_partition:
        # make sure all images have the same field, filter, ccdid, qid:
        ensure_images_have_the_same_properties(images, GROUP_PROPERTIES)
    for image in images:
        if getattr(image, 'mask_image', None) is None:
            raise ValueError(f'Image "{image.basename}" does not have a mask. '
                             f'Map this image to a mask and try again.')
        if 'MJD-OBS' not in image.header and set_date:
            image.header['MJD-OBS'] = get_time(image, 'mjd')
            image.header_comments['MJD-OBS'] = 'MJD of observation (DG)'

    # the reference isolates every call in tmpdir/<uuid> because SWarp exchanges
    # files; buffers are handed over in memory here, only the call objects keep
    # the directory name for their legacy command strings
    directory = Path(tmpdir) / uuid.uuid4().hex
    sci = prepare_swarp_sci(list(images), outname, directory, swarp_kws=sci_swarp_kws,
                            swarp_zp_key=swarp_zp_key)
    masks = [image.mask_image for image in images]
    mskoutweightname = mskoutname.replace('.fits', '.weight.fits')
    msk = prepare_swarp_mask(masks, mskoutname, mskoutweightname, directory,
                             swarp_kws=mask_swarp_kws)
    try:
        os.rmdir(directory)
    except OSError:
        pass

    params = dict(sci.params)
    params['mask_combine'] = msk.params['mask_combine']
    from . import objdev
    if objdev.enabled():
        # the device route: raw FITS blocks -> HBM -> kernels -> encoded products, every file written once
        return _coadd_device(cls, images, masks, sci, params, outname, mskoutname, addbkg, enforce_partition,
                             set_date, calculate_seeing, data_product)

    # one fused device pass: science frames and their masks share the lattice
    eng = get_engine()
    wout = sci.output_grid()
    frames = sci.frames()
    for f, m in zip(frames, masks):
        # (a ZTF mask read from its BITPIX 16 file is int16 and goes to the GPU as it is: zm_frame.mask_type)
        f['mask'] = m.data if m.data.dtype == np.int16 else np.ascontiguousarray(m.data).astype(np.int32)
    oimg, owgt, omask, omw = eng.coadd(frames, wout, coadd_params(**params), want_mask=True)

    weight_outname = outname.replace('.fits', '.weight.fits')
    hdr = output_header(images, wout, SCI_COPY_KEYWORDS)
    mhdr = output_header(masks, wout, MSK_COPY_KEYWORDS)
    if addbkg:
        oimg = oimg + np.float32(BKG_VAL)
    _fits.write(outname, oimg, hdr)
    _fits.write(weight_outname, owgt, hdr)
    _fits.write(mskoutname, omask, mhdr)

    # load the result
    coadd = cls.from_file(outname, load_others=False)
    coadd._weightimg = FITSImage.from_file(weight_outname)
    coaddmask = MaskImage.from_file(mskoutname)
    coaddmaskweight = FITSImage()
    coaddmaskweight.data = omw
    coaddmask.update_from_weight_map(coaddmaskweight)   # bit 16, zuds/coadd.py:182-184

    # keep a record of the images that went into the coadd
    coadd.input_images = images.tolist()
    coadd.mask_image = coaddmask
    coaddmask.parent_image = coadd
    if enforce_partition:
        for prop in GROUP_PROPERTIES:
            for img in [coadd, coaddmask]:
                setattr(img, prop, getattr(images[0], prop, None))
                if getattr(images[0], prop, None) is not None:
                    img.header[prop.upper()] = getattr(images[0], prop)
    if set_date:
        mjds = [get_time(i, 'mjd') for i in images]
        coadd.header['MJD-OBS'] = float(np.median(mjds))
        coadd.header_comments['MJD-OBS'] = 'Median MJD of the coadd inputs (DG)'
    coadd.save()
    coaddmask.save()
    if calculate_seeing:
        # zuds/coadd.py:225-226; the stars come from the pixels instead of a Gaia match
        # (seeing.py).  A coadd without a usable star keeps the inputs' median FWHM.
        from .seeing import estimate_seeing
        try:
            estimate_seeing(coadd)
        except RuntimeError:
            see = [i.header['SEEING'] for i in images if 'SEEING' in i.header]
            if see:
                coadd.header['SEEING'] = float(np.median(see))
                coadd.header_comments['SEEING'] = 'Median SEEING of the coadd inputs (pixels)'
                coadd.save()
    if data_product:
        warnings.warn('data_product=True: archiving is not part of this package')
    return coadd


def _coadd_device(cls, images, masks, sci, params, outname, mskoutname, addbkg, enforce_partition, set_date,
                  calculate_seeing, data_product):
    """``_coadd_from_images`` from the SWarp call objects on, on device planes (``objdev``): the same
    products with the same cards as the host-pointer route below writes (``tests/test_object_route_gpu.py``),
    without its numpy decode / widen / encode passes and without writing any file twice.  Replaces the file
    traffic of ``zuds/coadd.py:126-163`` (two SWarp runs) and ``:165-217`` (reload, bit 16, pedestal, saves)."""
    import ctypes as C

    from . import _lib, objdev
    from .constants import MASK_BITS, MASK_COMMENTS
    from .device import DeviceCoadd, DeviceFrames
    from .engine import coadd_params
    from .mask import MaskImage
    from .seeing import measure_seeing_dev
    from .swarp import MSK_COPY_KEYWORDS, SCI_COPY_KEYWORDS, output_header
    from .constants import BAD_SUM
    oio = objdev.get_io()
    torch, eng, L = oio.torch, oio.engine, oio.engine.L
    check = _lib.check
    wout = sci.output_grid()
    if sci.use_weights:
        # frames that came without maps (science image + mask only): their rms and weight maps in one batch on the
        # device, instead of one `weight_image` after the other (zuds/swarp.py:43-51 asks frame by frame)
        cold = [im for im in images if not hasattr(im, '_weightimg') and objdev.can_derive(im) and
                (not hasattr(im, '_rmsimg') or (im._rmsimg.ismapped and '_data' not in im._rmsimg.__dict__))]
        objdev.derive_maps_many(cold, want_weight=True)
    want = []
    for im, m in zip(images, masks):
        want += [(im, 'f32'), (im.weight_image if sci.use_weights else None, 'f32'), (m, 'mask')]
    planes = oio.planes(want)
    frames = [dict(img=planes[3 * i], wgt=planes[3 * i + 1], mask=planes[3 * i + 2], wcs=im.wcs,
                   flxscale=sci.flxscales[i] if sci.flxscales else 1.0) for i, im in enumerate(images)]
    co = DeviceCoadd(wout, coadd_params(**params), device=oio.device.index, engine=eng, want_mask=True,
                     stream=oio.stream)
    co.run(DeviceFrames(frames, oio.device))
    n = co.img.numel()
    weight_outname = outname.replace('.fits', '.weight.fits')
    # the cards of the products: what SWarp would give them + what from_images adds afterwards
    shape = tuple(co.shape)
    hdr, com = objdev.written_header(output_header(images, wout, SCI_COPY_KEYWORDS), {}, shape, -32)
    whdr = dict(output_header(images, wout, SCI_COPY_KEYWORDS))
    mhdr, mcom = objdev.written_header(output_header(masks, wout, MSK_COPY_KEYWORDS), {}, shape, 32)
    mhdr.update(MASK_BITS)                                # refresh_bit_mask_entries_in_header (zuds/mask.py:19-24)
    mcom.update(MASK_COMMENTS)
    if enforce_partition:
        for prop in GROUP_PROPERTIES:
            v = getattr(images[0], prop, None)
            if v is not None:
                hdr[prop.upper()] = v
                mhdr[prop.upper()] = v
    if set_date:
        hdr['MJD-OBS'] = float(np.median([get_time(i, 'mjd') for i in images]))
        com['MJD-OBS'] = 'Median MJD of the coadd inputs (DG)'
    eng.set_stream(oio.stream.cuda_stream)
    with torch.cuda.stream(oio.stream):
        # bit 16 where the resampler found no data (zuds/coadd.py:182-184, zuds/mask.py:26-33), the pedestal
        check(L.zm_mask_flag_dev(eng.ctx, co.mask.data_ptr(), co.mask_wgt.data_ptr(), 0.0, 1 << 16, n), 'bit 16')
        if addbkg:
            check(L.zm_add_scalar_dev(eng.ctx, co.img.data_ptr(), float(BKG_VAL), n), 'pedestal')
        if calculate_seeing:
            # zuds/coadd.py:225-226 on the plane in HBM; a coadd without a usable star keeps the inputs' median
            bad = torch.empty(co.shape, dtype=torch.uint8, device=oio.device)
            check(L.zm_mask_bad_dev(eng.ctx, co.mask.data_ptr(), None, BAD_SUM, n, None, bad.data_ptr()), 'bpm')
            try:
                see, _ = measure_seeing_dev(co.img, bad, None, engine=eng)
                hdr['SEEING'] = float(see)
                com['SEEING'] = 'FWHM of seeing in pixels (Goldstein)'
            except RuntimeError:
                see = [i.header['SEEING'] for i in images if 'SEEING' in i.header]
                if see:
                    hdr['SEEING'] = float(np.median(see))
                    com['SEEING'] = 'Median SEEING of the coadd inputs (pixels)'
    oio.save_all([(outname, co.img, hdr, com), (weight_outname, co.wgt, whdr, {}),
                  (mskoutname, co.mask, mhdr, mcom)])
    coadd = cls.from_file(outname, load_others=False)
    coadd._weightimg = FITSImage.from_file(weight_outname)
    coaddmask = MaskImage.from_file(mskoutname)
    coadd.input_images = images.tolist()
    coadd.mask_image = coaddmask
    coaddmask.parent_image = coadd
    if enforce_partition:
        for prop in GROUP_PROPERTIES:
            for img in [coadd, coaddmask]:
                setattr(img, prop, getattr(images[0], prop, None))
    if data_product:
        warnings.warn('data_product=True: archiving is not part of this package')
    return coadd


class Coadd(CalibratableImage):
    """``zuds/coadd.py:239-284``."""
    input_images = None

    @property
    def mjd(self):
        mjds = [image.mjd for image in self.input_images] if self.input_images else None
        return float(np.median(mjds)) if mjds else self.header.get('MJD-OBS')

    @property
    def min_mjd(self):
        return min([image.mjd for image in self.input_images])

    @property
    def max_mjd(self):
        return max([image.mjd for image in self.input_images])

    from_images = classmethod(_coadd_from_images)


class ReferenceImage(Coadd):
    """``zuds/coadd.py:287-299``."""
    version = None


class ScienceCoadd(Coadd):
    """``zuds/coadd.py:302-315``."""
    binleft = None
    binright = None

    @property
    def winsize(self):
        return self.binright - self.binleft
