"""Bit-mask images (``zuds/mask.py``)."""
import numpy as np

from .constants import BAD_SUM, MASK_BITS, MASK_COMMENTS
from .image import FITSImage

__all__ = ['MaskImageBase', 'MaskImage']


class MaskImageBase(FITSImage):

    __diskmapped_cached_properties__ = FITSImage.__diskmapped_cached_properties__ + ['_boolean']

    def load_data(self):
        """Masks are kept as int32 in memory whatever their BITPIX on disk (ZTF
        ships int16; bits 16 and 17 do not fit there)."""
        from . import fits as _fits
        data, _, _ = _fits.read(self.local_path)
        self._data = np.ascontiguousarray(data).astype(np.int32)

    def refresh_bit_mask_entries_in_header(self):
        """Write the bit dictionary into the header and save
        (``zuds/mask.py:19-24``)."""
        if self.header is None:
            self.header = {}
        if self.header_comments is None:
            self.header_comments = {}
        self.header.update(MASK_BITS)
        self.header_comments.update(MASK_COMMENTS)
        self.save()

    def update_from_weight_map(self, weight_image):
        """Flag bit 16 where the resampler found no data (``zuds/mask.py:26-33``)."""
        mskarr = np.asarray(self.data).astype(np.int32)
        ftsarr = weight_image.data
        mskarr[ftsarr == 0] += 2 ** 16
        self.data = mskarr
        self.refresh_bit_mask_entries_in_header()

    @property
    def boolean(self):
        """True where a pixel is unusable for science: ``(data & BAD_SUM) > 0``
        (``zuds/mask.py:42-72``; BAD_SUM = 198589, ``zuds/constants.py:45-46``)."""
        try:
            return self._boolean
        except AttributeError:
            maskpix = (np.asarray(self.data).astype(np.int32) & BAD_SUM) > 0
            _boolean = FITSImage()
            _boolean.data = maskpix
            _boolean.header = self.header
            _boolean.header_comments = self.header_comments
            _boolean.basename = self.basename.replace('.fits', '.bpm.fits')
            self._boolean = _boolean
        return self._boolean


class MaskImage(MaskImageBase):
    """The reference's DB-mapped mask class (``zuds/mask.py:75-94``); without a
    database it only carries the partition attributes."""
    field = ccdid = qid = fid = None
    parent_image = None
