"""SWarp front end (``zuds/swarp.py``): same function names and arguments, but
``prepare_*`` return a :class:`SwarpCall` (parameters + borrowed arrays) whose
``run()`` calls libzudsmi where the reference ran ``subprocess.check_call`` on
the command string (``zuds/coadd.py:131-140,156``, ``zuds/swarp.py:173-182``).
``SwarpCall.command`` keeps the legacy command line for logs."""
import os
import uuid
from pathlib import Path

import numpy as np

from .constants import BKG_BOX_SIZE
from .utils import initialize_directory
from .wcs import WCS

__all__ = ['prepare_swarp_sci', 'prepare_swarp_mask', 'prepare_swarp_align',
           'run_align', 'SwarpCall']

CONF_DIR = Path(__file__).parent / 'astromatic/makecoadd'
SCI_CONF = CONF_DIR / 'default.swarp'
MSK_CONF = CONF_DIR / 'mask.swarp'

# COPY_KEYWORDS of default.swarp:104 / mask.swarp:103
SCI_COPY_KEYWORDS = ['FLXSCLZP', 'OBJECT', 'FIELDID', 'CCDID', 'QID', 'FILTERID', 'FID']
MSK_COPY_KEYWORDS = ['OBJECT']


def _yn(v):
    return str(v).strip().upper() in ('Y', 'YES', 'TRUE', '1')


# The `-KEY value` pass-through of the reference (``zuds/swarp.py:76-78,100-102``) reaches SWarp
# itself, so every SWarp keyword is a legal override.  Three classes here:
#   * implemented - mapped onto zm_coadd_params below;
#   * keys that CHANGE THE OPERATOR and that libzudsmi does not implement: accepted when the value
#     is the one the engine works with (the value of default.swarp / mask.swarp), ``ValueError``
#     otherwise - a caller who overrides the projection or the centre must not get the default
#     operator without a word;
#   * bookkeeping keys (threads, scratch directories, logs ...): ignored;
# anything else warns once per key.
_SWARP_FIXED = {            # key -> values the engine's operator corresponds to
    'PROJECTION_TYPE': ('TPV', 'TAN'), 'PROJECTION_ERR': None,      # (the lattice is exact at its nodes)
    'CELESTIAL_TYPE': ('NATIVE',), 'CENTER_TYPE': ('ALL',), 'PIXELSCALE_TYPE': ('MEDIAN',),
    'PIXEL_SCALE': ('0', '0.0'), 'IMAGE_SIZE': ('0',), 'OVERSAMPLING': ('0',),
    'INTERPOLATE': ('N',), 'FSCALASTRO_TYPE': ('FIXED',), 'FSCALE_KEYWORD': ('FLXSCALE',),
    'FSCALE_DEFAULT': ('1', '1.0'), 'BACK_TYPE': ('AUTO',), 'BACK_FILTTHRESH': ('0', '0.0'),
    'BLANK_BADPIXELS': ('N',), 'COMBINE': ('Y',), 'RESAMPLE': ('Y',), 'HEADER_ONLY': ('N',),
    'GAIN_DEFAULT': ('0', '0.0'), 'WEIGHT_TYPE': ('MAP_WEIGHT', 'NONE'),
}
_SWARP_UNSUPPORTED = ('CENTER', 'BACK_DEFAULT', 'WEIGHT_IMAGE', 'WEIGHT_SUFFIX', 'IMAGEOUT_NAME',
                      'WEIGHTOUT_NAME', 'HEADER_SUFFIX', 'GAIN_KEYWORD', 'SATLEV_KEYWORD',
                      'SATLEV_DEFAULT', 'COPY_KEYWORDS')
_SWARP_IGNORED = ('NTHREADS', 'VMEM_DIR', 'VMEM_MAX', 'MEM_MAX', 'COMBINE_BUFSIZE', 'RESAMPLE_DIR',
                  'RESAMPLE_SUFFIX', 'DELETE_TMPFILES', 'VERBOSE_TYPE', 'WRITE_XML', 'XML_NAME',
                  'XSL_URL', 'WRITE_FILEINFO', 'NNODES', 'NODE_INDEX', 'CLIP_WRITELOG',
                  'CLIP_LOGNAME', 'NOPENFILES_MAX', 'TILE_COMPRESS',
                  # not SWarp keys: keyword arguments MultiEpochSubtraction.from_images forwards
                  # as swarp_kws (zuds/subtraction.py:308-310)
                  'REFINED', 'FORCE_MAP_SUBS', 'TMPDIR')
_warned_keys = set()


def _params_from_kws(base, swarp_kws, kind='sci'):
    """Overlay the ``-KEY value`` pass-through of the reference
    (``zuds/swarp.py:76-78``) on the config-file defaults; see the key classes above."""
    p = dict(base)
    for k, v in (swarp_kws or {}).items():
        ku = str(k).upper()
        if ku == 'COMBINE_TYPE':
            p['combine'] = str(v).upper()
        elif ku == 'RESAMPLING_TYPE':
            p['resample'] = str(v).upper()
        elif ku == 'CLIP_SIGMA':
            p['clip_sigma'] = float(v)
        elif ku == 'CLIP_AMPFRAC':
            p['clip_ampfrac'] = float(v)
        elif ku == 'SUBTRACT_BACK':
            p['subtract_back'] = _yn(v)
        elif ku == 'BACK_SIZE':
            p['back_size'] = int(v)
        elif ku == 'BACK_FILTERSIZE':
            p['back_filtersize'] = int(v)
        elif ku == 'RESCALE_WEIGHTS':
            p['rescale_weights'] = _yn(v)
        elif ku == 'WEIGHT_THRESH':
            p['weight_thresh'] = float(v)
        elif ku in _SWARP_FIXED:
            ok = _SWARP_FIXED[ku]
            if ok is not None and str(v).strip().upper() not in ok:
                raise ValueError(f'SWarp keyword -{ku} {v}: libzudsmi implements {" / ".join(ok)} only '
                                 f'(the {kind} operator of default.swarp / mask.swarp)')
        elif ku in _SWARP_UNSUPPORTED:
            raise ValueError(f'SWarp keyword -{ku} {v} changes the {kind} coadd and is not implemented '
                             f'by libzudsmi')
        elif ku in _SWARP_IGNORED:
            continue
        elif ku not in _warned_keys:
            import warnings
            _warned_keys.add(ku)
            warnings.warn(f'SWarp keyword -{ku} {v} has no equivalent in libzudsmi and is ignored')
    return p


# default.swarp: CLIPPED, 0.3 / 4.0, LANCZOS3, SUBTRACT_BACK Y, RESCALE_WEIGHTS Y
_SCI_DEFAULTS = dict(combine='CLIPPED', mask_combine='AND', resample='LANCZOS3',
                     subtract_back=True, back_size=256, back_filtersize=3,
                     rescale_weights=True, clip_sigma=4.0, clip_ampfrac=0.3,
                     weight_thresh=1e-30)


class SwarpCall(object):
    """One resample + combine job."""

    def __init__(self, kind, images, outname, wgtout, params, command, grid=None,
                 flxscales=None, use_weights=True):
        self.kind = kind                  # 'sci' | 'mask' | 'align'
        self.images = list(images)
        self.outname = str(outname)
        self.wgtout = str(wgtout) if wgtout is not None else None
        self.params = params
        self.command = command
        self.grid = grid                  # forced output WCS (the `.head` file) or None
        self.flxscales = flxscales
        self.use_weights = use_weights

    def split(self):
        return self.command.split()

    def output_grid(self):
        from .engine import get_engine
        if self.grid is not None:
            return self.grid
        return get_engine().autogrid([im.wcs for im in self.images])

    def frames(self, as_mask=False):
        frames = []
        for i, im in enumerate(self.images):
            if as_mask:
                frames.append(dict(img=np.zeros(im.data.shape, dtype=np.float32), wgt=None,
                                   mask=im.data if im.data.dtype == np.int16
                                   else np.ascontiguousarray(im.data).astype(np.int32),
                                   wcs=im.wcs, flxscale=1.0))
            else:
                wgt = im.weight_image.data if self.use_weights else None
                frames.append(dict(img=im.data, wgt=wgt, mask=None, wcs=im.wcs,
                                   flxscale=self.flxscales[i] if self.flxscales else 1.0))
        return frames

    def run(self):
        """Execute and write ``outname`` / ``wgtout`` like SWarp would."""
        from . import fits as _fits
        from .engine import coadd_params, get_engine
        eng = get_engine()
        wout = self.output_grid()
        p = coadd_params(**self.params)
        hdr = output_header(self.images, wout,
                            MSK_COPY_KEYWORDS if self.kind == 'mask' else SCI_COPY_KEYWORDS)
        if self.kind == 'mask' or (self.kind == 'align' and _is_mask(self.images[0])):
            _, _, omask, omw = eng.coadd(self.frames(as_mask=True), wout, p, want_mask=True)
            _fits.write(self.outname, omask, hdr)
            if self.wgtout:
                _fits.write(self.wgtout, omw, hdr)
            return omask, omw
        oimg, owgt, _, _ = eng.coadd(self.frames(), wout, p, want_mask=False)
        _fits.write(self.outname, oimg, hdr)
        if self.wgtout:
            _fits.write(self.wgtout, owgt, hdr)
        return oimg, owgt


def _is_mask(image):
    from .mask import MaskImageBase
    return isinstance(image, MaskImageBase)


def output_header(images, wout, copy_keywords):
    """Header SWarp gives its products: output grid + COPY_KEYWORDS of the first
    input that carries them."""
    hdr = {'NAXIS1': int(wout.naxis[0]), 'NAXIS2': int(wout.naxis[1])}
    hdr.update(wout.to_header())
    for k in copy_keywords:
        for im in images:
            if im.header and k in im.header:
                hdr[k] = im.header[k]
                break
    hdr['SOFTNAME'] = 'zudsmi'
    hdr['NCOMBINE'] = len(images)
    return hdr


def flux_scale_of(im, swarp_zp_key='MAGZP'):
    """FLXSCALE = 10^(-0.4 (MAGZP - 25)) (``zuds/swarp.py:29-35``); without the
    zero-point key SWarp falls back to an existing FLXSCALE card, else 1
    (``FSCALE_KEYWORD`` / ``FSCALE_DEFAULT``, default.swarp:59-62)."""
    if swarp_zp_key in im.header:
        return 10 ** (-0.4 * (im.header[swarp_zp_key] - 25.))
    return float(im.header.get('FLXSCALE', 1.0))


def prepare_swarp_sci(images, outname, directory, swarp_kws=None, swarp_zp_key='MAGZP'):
    """Science coadd job (``zuds/swarp.py:20-80``)."""
    conf = SCI_CONF
    initialize_directory(directory)
    directory = Path(directory)
    impaths = [im.local_path if im.ismapped else im.basename for im in images]
    flx = []
    for im in images:
        # normalize all images to the same zeropoint
        if swarp_zp_key in im.header:
            im.header['FLXSCALE'] = flux_scale_of(im, swarp_zp_key)
            im.header_comments['FLXSCALE'] = 'Flux scale factor for coadd / DG'
            im.header['FLXSCLZP'] = 25.
            im.header_comments['FLXSCLZP'] = 'FLXSCALE equivalent ZP / DG'
        flx.append(flux_scale_of(im, swarp_zp_key))
    wgtout = str(outname).replace('.fits', '.weight.fits')
    inlist = directory / 'images.in'
    inweight = directory / 'weight.in'
    syscall = f'swarp -c {conf} @{inlist} ' \
              f'-BACK_SIZE {BKG_BOX_SIZE} ' \
              f'-IMAGEOUT_NAME {outname} ' \
              f'-VMEM_DIR {directory} ' \
              f'-RESAMPLE_DIR {directory} ' \
              f'-WEIGHT_IMAGE @{inweight} ' \
              f'-WEIGHTOUT_NAME {wgtout} '
    if swarp_kws is not None:
        for kw in swarp_kws:
            syscall += f'-{kw.upper()} {swarp_kws[kw]} '
    base = dict(_SCI_DEFAULTS, back_size=BKG_BOX_SIZE)
    params = _params_from_kws(base, swarp_kws)
    return SwarpCall('sci', images, outname, wgtout, params, syscall, flxscales=flx)


def prepare_swarp_mask(masks, outname, mskoutweightname, directory, swarp_kws=None):
    """Mask coadd job: ``mask.swarp`` (AND, WEIGHT_TYPE NONE) with
    ``-SUBTRACT_BACK N`` (``zuds/swarp.py:83-104``)."""
    conf = MSK_CONF
    initialize_directory(directory)
    allims = ' '.join([c.local_path if c.ismapped else c.basename for c in masks])
    syscall = f'swarp -c {conf} {allims} ' \
              f'-SUBTRACT_BACK N ' \
              f'-IMAGEOUT_NAME {outname} ' \
              f'-VMEM_DIR {directory} ' \
              f'-RESAMPLE_DIR {directory} ' \
              f'-WEIGHTOUT_NAME {mskoutweightname} '
    if swarp_kws is not None:
        for kw in swarp_kws:
            syscall += f'-{kw.upper()} {swarp_kws[kw]} '
    base = dict(_SCI_DEFAULTS, combine='CLIPPED', mask_combine='AND', subtract_back=False,
                rescale_weights=False)
    params = _params_from_kws(base, swarp_kws, kind='mask')
    mc = str((swarp_kws or {}).get('COMBINE_TYPE', (swarp_kws or {}).get('combine_type', 'AND'))).upper()
    params['mask_combine'] = mc if mc in ('AND', 'OR') else 'AND'
    params['combine'] = 'WEIGHTED'
    params['subtract_back'] = False
    return SwarpCall('mask', masks, outname, mskoutweightname, params, syscall,
                     use_weights=False)


def prepare_swarp_align(image, other, directory, nthreads=1, persist_aligned=False):
    """Single-image alignment job onto the grid of ``other``
    (``zuds/swarp.py:107-154``): grid from the `.head` cards = target NAXIS +
    target WCS, ``-SUBTRACT_BACK N -WEIGHT_TYPE NONE``, ``COMBINE_TYPE`` OR for
    masks else CLIPPED."""
    conf = SCI_CONF
    directory = Path(directory)
    impath = str(directory / image.basename)
    align_header = other.astropy_header
    grid = WCS.from_header(align_header)
    extension = f'_aligned_to_{other.basename[:-5]}.remap'
    if persist_aligned:
        outname = image.local_path.replace('.fits', f'{extension}.fits')
    else:
        outname = impath.replace('.fits', f'{extension}.fits')
    weightname = directory / image.basename.replace('.fits', f'{extension}.weight.fits')
    combtype = 'OR' if _is_mask(image) else 'CLIPPED'
    syscall = f'swarp -c {conf} {impath} ' \
              f'-BACK_SIZE {BKG_BOX_SIZE} ' \
              f'-IMAGEOUT_NAME {outname} ' \
              f'-NTHREADS {nthreads} ' \
              f'-VMEM_DIR {directory} ' \
              f'-RESAMPLE_DIR {directory} ' \
              f'-SUBTRACT_BACK N ' \
              f'-WEIGHTOUT_NAME {weightname} ' \
              f'-WEIGHT_TYPE NONE ' \
              f'-COMBINE_TYPE {combtype} '
    params = dict(_SCI_DEFAULTS, back_size=BKG_BOX_SIZE, subtract_back=False,
                  rescale_weights=False, combine='CLIPPED', mask_combine='OR')
    call = SwarpCall('align', [image], outname, weightname, params, syscall, grid=grid,
                     # an align run does not rewrite FLXSCALE from MAGZP: SWarp applies
                     # whatever FLXSCALE card the file carries (FSCALE_KEYWORD)
                     flxscales=[float((image.header or {}).get('FLXSCALE', 1.0))]
                     if not _is_mask(image) else None,
                     use_weights=False)
    return call, outname, weightname


def run_align(image, other, tmpdir='/tmp', nthreads=1, persist_aligned=False):
    """Resample ``image`` onto the grid of ``other`` (``zuds/swarp.py:157-204``).

    No temporary directory is needed: the product is returned in memory and
    only written when ``persist_aligned`` is set."""
    from . import fits as _fits
    from .engine import get_engine
    from .image import FITSImage
    from .mask import MaskImage, MaskImageBase

    directory = Path(tmpdir) / uuid.uuid4().hex
    call, outname, outweight = prepare_swarp_align(image, other, directory, nthreads=nthreads,
                                                   persist_aligned=persist_aligned)
    eng = get_engine()
    wout = call.grid
    ismask = _is_mask(image)
    if ismask:
        _, _, data, weight = eng.resample_mask(image.data, image.wcs, wout)
    else:
        # SWarp applies FLXSCALE on resampling whenever the header carries it
        fs = eng.flux_scale(image.wcs, wout, call.flxscales[0])
        data, weight, _ = eng.resample(image.data, image.wcs, wout, fscale=fs)
    restype = MaskImageBase if ismask else FITSImage
    result = restype()
    result.basename = os.path.basename(outname)
    result.header = output_header([image], wout, SCI_COPY_KEYWORDS)
    result.header_comments = {}
    result.data = data
    result.parent_image = image
    if isinstance(image, MaskImage):
        # bit 16 where the resampler found no data (zuds/swarp.py:190-191)
        result.data[weight == 0] += 2 ** 16
    if persist_aligned:
        _fits.write(outname, result.data, result.header)
        result.map_to_local_file(outname)
    return result
