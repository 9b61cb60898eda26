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

    # one fused device pass: science frames and their masks share the lattice
    eng = get_engine()
    wout = sci.output_grid()
    params = dict(sci.params)
    params['mask_combine'] = msk.params['mask_combine']
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
