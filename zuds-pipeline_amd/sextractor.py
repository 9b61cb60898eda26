"""Mesh background / rms check-images (``zuds/sextractor.py``).

Only the check-images are on this path (``BACKGROUND_RMS``, ``-BACKGROUND``,
``BACKGROUND``; ``zuds/sextractor.py:21-26``); catalog extraction is out of
scope.  The SExtractor process is replaced by ``zm_background``.
"""
import os

import numpy as np

from .constants import BAD_SUM, BKG_BOX_SIZE, MASK_BORDER

__all__ = ['prepare_sextractor', 'run_sextractor']

checkimage_map = {'rms': 'BACKGROUND_RMS', 'segm': 'SEGMENTATION',
                  'bkgsub': '-BACKGROUND', 'bkg': 'BACKGROUND'}
_SUPPORTED = ['rms', 'bkgsub', 'bkg']


def prepare_sextractor(image, directory=None, checkimage_type=None,
                       catalog_type='FITS_LDAC', use_weightmap=True, sextractor_kws=None):
    """Parameters of one background run: dict(weight, mesh, filtersize, outnames,
    types) (the reference returns a ``sex`` command line, ``zuds/sextractor.py:29-107``)."""
    sextractor_kws = sextractor_kws or {}
    checkimage_types = np.atleast_1d(checkimage_type or []).tolist()
    if 'all' in checkimage_types:
        checkimage_types = list(_SUPPORTED)
    for t in checkimage_types:
        if t not in checkimage_map:
            raise ValueError(f'Invalid CHECKIMAGE_TYPE "{t}". Must be one of '
                             f'{list(checkimage_map)}.')
        if t not in _SUPPORTED:
            raise NotImplementedError(f'CHECKIMAGE_TYPE "{t}" needs source extraction, '
                                      f'which is outside the coadd / subtraction path')
    if use_weightmap:
        weight = image.weight_image.data
    else:
        # false weight map: masked pixels (and a 10-pixel border of raw science
        # frames) are excluded from the background statistics
        # (zuds/sextractor.py:80-96)
        weight = np.ones(image.mask_image.data.shape, dtype='<f4')
        weight[(image.mask_image.data & BAD_SUM) > 0] = 0
        if image.basename.endswith('sciimg.fits'):
            weight[:MASK_BORDER] = 0
            weight[-MASK_BORDER:] = 0
            weight[:, :MASK_BORDER] = 0
            weight[:, -MASK_BORDER:] = 0
    base = image.local_path if image.ismapped else image.basename
    outnames = [base.replace('.fits', f'.{t}.fits') for t in checkimage_types]
    return dict(weight=weight, mesh=int(sextractor_kws.get('BACK_SIZE', BKG_BOX_SIZE)),
                filtersize=int(sextractor_kws.get('BACK_FILTERSIZE', 3)),
                outnames=outnames, types=checkimage_types)


def run_sextractor(image, checkimage_type=None, catalog_type='FITS_LDAC', tmpdir='/tmp',
                   use_weightmap=True, sextractor_kws=None):
    """Produce the requested check-images as FITSImage objects, written next to
    the image when it is mapped (``zuds/sextractor.py:110-150``).  The returned
    list starts with ``None`` in the catalog slot."""
    from .engine import get_engine
    from .image import FITSImage
    call = prepare_sextractor(image, None, checkimage_type=checkimage_type,
                              catalog_type=catalog_type, use_weightmap=use_weightmap,
                              sextractor_kws=sextractor_kws)
    want = {'bkg': 'bkg', 'rms': 'rms', 'bkgsub': 'sub'}
    bkg, rms, sub, stats = get_engine().background(
        image.data, call['weight'], mesh=call['mesh'], filtersize=call['filtersize'],
        want=tuple(want[t] for t in call['types']))
    planes = {'bkg': bkg, 'rms': rms, 'bkgsub': sub}
    result = [None]
    for t, name in zip(call['types'], call['outnames']):
        product = FITSImage()
        product.basename = os.path.basename(name)
        product.data = planes[t]
        product.header = dict(image.header)
        product.header_comments = dict(image.header_comments or {})
        for prop in ('field', 'ccdid', 'qid', 'fid'):
            setattr(product, prop, getattr(image, prop, None))
        if image.ismapped:
            product.map_to_local_file(name)
            product.save()
        result.append(product)
    return result
