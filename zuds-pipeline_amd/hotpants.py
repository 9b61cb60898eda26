"""hotpants front end (``zuds/hotpants.py``): ``prepare_hotpants`` keeps its
signature and returns a :class:`HotpantsCall` whose ``run()`` calls
``zm_subtract`` where the reference ran the ``hotpants`` binary
(``zuds/subtraction.py:162``).  ``HotpantsCall.command`` is the legacy command
line (``zuds/hotpants.py:77-93``)."""
import os

import numpy as np

from .constants import BIG_RMS, BKG_VAL
from .utils import initialize_directory, quick_background_estimate

__all__ = ['prepare_hotpants', 'HotpantsCall']

_INT_KEYS = ('ko', 'bgo', 'nss', 'nsx', 'nsy', 'nrx', 'nry')
_FLT_KEYS = ('tu', 'tl', 'iu', 'il', 'r', 'rss', 'fin', 'fi', 'ft', 'ks')


def chunk(iterable, chunksize):
    isize = len(iterable)
    nchunks = isize // chunksize if isize % chunksize == 0 else isize // chunksize + 1
    for i in range(nchunks):
        yield i, iterable[i * chunksize: (i + 1) * chunksize]


class HotpantsCall(object):

    def __init__(self, command, params, sci, scirms, ref, refrms, bpm, outname, subrms, header):
        self.command = command
        self.params = params
        self.sci, self.scirms, self.ref, self.refrms, self.bpm = sci, scirms, ref, refrms, bpm
        self.outname, self.subrms, self.header = outname, subrms, header
        self.info = None

    def split(self):
        return self.command.split()

    def run(self):
        """Subtract and write ``outname`` / ``.rms.fits`` (hotpants ``-outim`` / ``-oni``)."""
        from . import fits as _fits
        from .engine import get_engine, hp_params
        p = hp_params(**self.params)
        diff, noise, info = get_engine().subtract(self.sci, self.scirms, self.ref,
                                                  self.refrms, self.bpm, params=p)
        self.info = info
        hdr = dict(self.header)
        hdr['KSUM00'] = float(info['kernel_sum'])        # -hki: kernel info in the header
        hdr['NSTAMPS'] = int(info['nstamps_used'])
        _fits.write(self.outname, diff, hdr)
        _fits.write(self.subrms, noise, hdr)
        return diff, noise


def prepare_hotpants(sci, ref, outname, submask, directory, tmpdir='/tmp',
                     nreg_side=3, subtract_new_back=True, hotpants_kws=None):
    """Assemble one subtraction job exactly as ``zuds/hotpants.py:15-95``:
    background-subtracted science + 150 counts, r = 2.5 SEEING, rss = 6 SEEING,
    nsx = NAXIS / 100 / nreg_side, reference rms aligned to the science grid,
    lower valid limits = robust background - 10 sigma, upper limits 5e3."""
    from .sextractor import run_sextractor
    initialize_directory(directory)
    if hotpants_kws is None:
        hotpants_kws = {}
    if subtract_new_back:
        scimbkg = run_sextractor(sci, checkimage_type=['bkgsub'])[1]
        scimbkg.data = scimbkg.data + np.float32(BKG_VAL)
        if scimbkg.ismapped:
            scimbkg.save()
    else:
        scimbkg = sci
    if 'SEEING' not in sci.header:
        # zuds/hotpants.py:38-42: measured here from the pixels (seeing.py)
        from .seeing import estimate_seeing
        estimate_seeing(sci)
        sci.save()
    seepix = sci.header['SEEING']   # header seeing is FWHM in pixels
    r = 2.5 * seepix
    rss = 6. * seepix
    nsx = sci.header['NAXIS1'] / 100.
    nsy = sci.header['NAXIS2'] / 100.
    scirms = sci.rms_image
    refrms = ref.parent_image.rms_image.aligned_to(scirms, tmpdir=tmpdir)
    scibkg, scibkgstd = quick_background_estimate(scimbkg, mask_image=sci.mask_image)
    refbkg, refbkgstd = quick_background_estimate(ref)
    subrms = outname.replace('.fits', '.rms.fits')
    il = scibkg - 10 * scibkgstd
    tl = refbkg - 10 * refbkgstd
    satlev = 5e3  # not perfect, but close enough.

    def p(im):
        return im.local_path if im.ismapped else im.basename
    syscall = f'hotpants -inim {p(scimbkg)} -hki -n i -c t ' \
              f'-tmplim {p(ref)} -outim {outname} ' \
              f'-tu {satlev} -iu {satlev}  -tl {tl} -il {il} -r {r} ' \
              f'-rss {rss} -tni {p(refrms)} ' \
              f'-ini {p(scirms)} ' \
              f'-imi {p(submask)}  -v 0 -oni {subrms} ' \
              f'-fin {BIG_RMS} -nsx {nsx / nreg_side} -nsy {nsy / nreg_side} ' \
              f'-nrx {nreg_side} -nry {nreg_side} '
    for key in hotpants_kws:
        syscall += f' -{key} {hotpants_kws[key]}'
    if 'bgo' not in hotpants_kws:
        syscall += ' -bgo 0'
    if 'ko' not in hotpants_kws:
        syscall += ' -ko 4'
    # hotpants parses -nsx / -nsy / -r / -rss with integer conversions
    params = dict(tu=satlev, iu=satlev, tl=tl, il=il, r=r, rss=rss, fin=float(BIG_RMS),
                  nsx=max(int(nsx / nreg_side), 1), nsy=max(int(nsy / nreg_side), 1),
                  nrx=int(nreg_side), nry=int(nreg_side), bgo=0, ko=4, normalize=0)
    for key, val in hotpants_kws.items():
        if key in _INT_KEYS:
            params[key] = int(val)
        elif key in _FLT_KEYS:
            params[key] = float(val)
        elif key == 'n':
            params['normalize'] = 1 if str(val) == 't' else 0
    bpm = np.ascontiguousarray(submask.data).astype(np.uint8)
    return HotpantsCall(syscall, params, scimbkg.data, scirms.data, ref.data, refrms.data,
                        bpm, outname, subrms, dict(sci.header))
