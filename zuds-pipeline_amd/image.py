"""Image / weight / rms object model (``zuds/image.py``), DB-free.

``weight_image`` and ``rms_image`` keep the reference's lazy semantics; the maps
the reference obtains from SExtractor check-images (``BACKGROUND_RMS``,
``-BACKGROUND``, ``BACKGROUND``; ``zuds/image.py:95-129,206,212-226``) come from
libzudsmi's mesh-background kernels instead.
"""
import os
from pathlib import Path

import numpy as np

from .constants import APER_KEY, BIG_RMS
from .fitsfile import HasWCS

__all__ = ['FITSImage', 'CalibratableImageBase', 'CalibratableImage',
           'CalibratedImage', 'ScienceImage']


class FITSImage(HasWCS):
    """A FITS file whose data member is an image (``zuds/image.py:28-88``)."""

    @property
    def datatype(self):
        return 'float' if 'float' in self.data.dtype.name else 'int'


class CalibratableImageBase(FITSImage):
    __diskmapped_cached_properties__ = ['_path', '_data', '_weightimg', '_bkgimg',
                                        '_filter_kernel', '_rmsimg', '_threshimg',
                                        '_segmimg', '_sourcelist', '_bkgsubimg']

    mask_image = None
    field = ccdid = qid = fid = None

    def _call_source_extractor(self, checkimage_type=None, tmpdir='/tmp',
                               use_weightmap=True, sextractor_kws=None):
        """Produce the requested check-images (``zuds/image.py:103-134``)."""
        from . import sextractor
        results = sextractor.run_sextractor(self, checkimage_type=checkimage_type,
                                            tmpdir=tmpdir, use_weightmap=use_weightmap,
                                            sextractor_kws=sextractor_kws)
        for result in results:
            if result is None:          # the catalog slot (no source extraction on this path)
                continue
            if result.basename.endswith('.rms.fits'):
                self._rmsimg = result
            elif result.basename.endswith('.bkgsub.fits'):
                self._bkgsubimg = result
            elif result.basename.endswith('.bkg.fits'):
                self._bkgimg = result

    def _derived(self, suffix, data):
        im = FITSImage()
        im.basename = self.basename.replace('.fits', suffix)
        im.data = data
        im.header = self.header
        im.header_comments = self.header_comments
        if self.ismapped:
            im.map_to_local_file(os.path.join(os.path.dirname(self.local_path), im.basename))
            im.save()     # guarantees the mapped file exists
        return im

    @property
    def weight_image(self):
        """Inverse-variance map: 1 / rms^2, 0 where the mask is bad or the pixel
        is within 10 % of SATURATE (``zuds/image.py:136-171``)."""
        try:
            return self._weightimg
        except AttributeError:
            from . import objdev
            if objdev.can_derive(self) and (not hasattr(self, '_rmsimg') or
                                            (self._rmsimg.ismapped and '_data' not in self._rmsimg.__dict__)):
                # a frame on disk without maps: derived on the device planes (objdev.derive_maps), same files
                objdev.derive_maps(self, want_weight=True)
                return self._weightimg
            ind = self.mask_image.boolean.data
            wgt = np.empty_like(ind, dtype='<f4')
            wgt[~ind] = 1 / self.rms_image.data[~ind] ** 2
            wgt[ind] = 0.
            if 'SATURATE' in self.header:
                wgt[self.data >= 0.9 * self.header['SATURATE']] = 0.
            self._weightimg = self._derived('.weight.fits', wgt)
        return self._weightimg

    @property
    def rms_image(self):
        """1 / sqrt(weight) with BIG_RMS on bad pixels when a weight map exists,
        else the mesh BACKGROUND_RMS map (``zuds/image.py:173-208``)."""
        try:
            return self._rmsimg
        except AttributeError:
            if hasattr(self, '_weightimg'):
                ind = self.mask_image.boolean.data
                rms = np.empty_like(ind, dtype='<f4')
                with np.errstate(divide='ignore'):
                    rms[~ind] = 1 / np.sqrt(self.weight_image.data[~ind])
                rms[ind] = BIG_RMS
                if 'SATURATE' in self.header:
                    rms[self.data >= 0.9 * self.header['SATURATE']] = BIG_RMS
                self._rmsimg = self._derived('.rms.fits', rms)
                return self._rmsimg
            else:
                from . import objdev
                if objdev.can_derive(self):
                    objdev.derive_maps(self, want_weight=False)
                else:
                    self._call_source_extractor(checkimage_type=['rms'], use_weightmap=False)
        return self._rmsimg

    @property
    def background_image(self):
        try:
            return self._bkgimg
        except AttributeError:
            self._call_source_extractor(checkimage_type=['bkg'])
        return self._bkgimg

    @property
    def background_subtracted_image(self):
        try:
            return self._bkgsubimg
        except AttributeError:
            self._call_source_extractor(checkimage_type=['bkgsub'])
        return self._bkgsubimg

    @classmethod
    def from_file(cls, fname, use_existing_record=True, load_others=True):
        """Also picks up sibling ``.weight/.rms/.bkg/.bkgsub.fits`` files
        (``zuds/image.py:236-262``)."""
        obj = super().from_file(fname, use_existing_record=use_existing_record)
        d = Path(fname).parent
        if load_others:
            for suffix, attr in (('.weight.fits', '_weightimg'), ('.rms.fits', '_rmsimg'),
                                 ('.bkg.fits', '_bkgimg'), ('.thresh.fits', '_threshimg'),
                                 ('.bkgsub.fits', '_bkgsubimg'), ('.segm.fits', '_segmimg')):
                path = d / obj.basename.replace('.fits', suffix)
                if path.exists() and path.name != obj.basename:
                    setattr(obj, attr, FITSImage.from_file(f'{path}'))
        for key, attr in (('FIELD', 'field'), ('FIELDID', 'field'), ('CCDID', 'ccdid'),
                          ('QID', 'qid'), ('FID', 'fid'), ('FILTERID', 'fid')):
            if obj.header and key in obj.header and getattr(obj, attr, None) is None:
                setattr(obj, attr, obj.header[key])
        return obj

    @property
    def mjd(self):
        from .utils import get_time
        return get_time(self, 'mjd')


class CalibratableImage(CalibratableImageBase):
    """``zuds/image.py:265-330`` without the ORM columns."""

    def basic_map(self, quiet=True):
        pass


class CalibratedImage(CalibratableImage):
    """An image with MAGZP / APCOR calibration (``zuds/image.py:333-432``)."""

    @property
    def magzp(self):
        return self.header['MAGZP'] + self.header[APER_KEY]


class ScienceImage(CalibratedImage):
    """A single-epoch IPAC science frame (``zuds/image.py:435-567``)."""

    @property
    def obsjd(self):
        from .utils import get_time
        return get_time(self, 'jd')
