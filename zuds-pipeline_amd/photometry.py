"""Forced aperture photometry (``zuds/photometry.py:61-249``): same two entry
points and table columns; photutils' exact circular apertures and the flag OR are
evaluated by ``zm_aperture_photometry``."""
import numpy as np

from . import fits as _fits
from .constants import APER_KEY, APERTURE_RADIUS
from .wcs import WCS

__all__ = ['PhotTable', 'raw_aperture_photometry', 'aperture_photometry']


class PhotTable(dict):
    """Minimal column table (the reference returns an astropy Table)."""

    def rename_column(self, old, new):
        self[new] = self.pop(old)

    def __len__(self):
        return len(next(iter(self.values()))) if self else 0

    @property
    def colnames(self):
        return list(self.keys())


def _table(wcs, header, data, rms, mask, ra, dec):
    from .engine import get_engine
    ra = np.atleast_1d(np.asarray(ra, dtype=np.float64))
    dec = np.atleast_1d(np.asarray(dec, dtype=np.float64))
    x, y = wcs.all_world2pix(ra, dec, 0)
    flux, err, flags = get_engine().aperture_photometry(data, x, y, rms=rms, mask=mask,
                                                        radius=APERTURE_RADIUS)
    t = PhotTable()
    t['id'] = np.arange(1, ra.size + 1)
    t['xcenter'] = x
    t['ycenter'] = y
    t['ra'] = ra
    t['dec'] = dec
    t['flux'] = flux                    # photutils: aperture_sum
    t['fluxerr'] = err                  # photutils: aperture_sum_err
    t['flags'] = flags.astype(np.int64)
    t['zp'] = np.full(ra.size, header['MAGZP'] + header[APER_KEY])
    t['obsjd'] = np.full(ra.size, header['OBSJD']) if 'OBSJD' in header else None
    t['filtercode'] = np.full(ra.size, 'z' + str(header['FILTER'])[-1]) if 'FILTER' in header else None
    return t


def raw_aperture_photometry(sci_path, rms_path, mask_path, ra, dec, apply_calibration=False):
    """Photometry straight from three FITS files (``zuds/photometry.py:61-113``)."""
    scipix, header, _ = _fits.read(sci_path)
    rmspix, _, _ = _fits.read(rms_path)
    maskpix, _, _ = _fits.read(mask_path)
    return _table(WCS.from_header(header), header, scipix, rmspix, maskpix, ra, dec)


def aperture_photometry(calibratable, ra, dec, apply_calibration=False,
                        assume_background_subtracted=False, use_cutout=False, direct_load=None):
    """Photometry on an image object (``zuds/photometry.py:116-249``).  ``use_cutout``
    only changes how the reference reads pixels from disk; the result is the same."""
    if not assume_background_subtracted:
        pixels = calibratable.background_subtracted_image.data
    else:
        pixels = calibratable.data
    t = _table(calibratable.wcs, calibratable.header, pixels, calibratable.rms_image.data,
               calibratable.mask_image.data, ra, dec)
    if apply_calibration:
        magzp = calibratable.header['MAGZP']
        apcor = calibratable.header[APER_KEY]
        with np.errstate(invalid='ignore', divide='ignore'):
            t['mag'] = -2.5 * np.log10(t['flux']) + magzp + apcor
            t['magerr'] = 1.0826 * t['fluxerr'] / t['flux']
    return t
