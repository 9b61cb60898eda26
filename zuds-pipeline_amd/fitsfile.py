"""FITS-file and WCS mixins: the interface of ``zuds/fitsfile.py`` without the
database columns.  I/O goes through :mod:`fits` (no astropy); alignment goes
through libzudsmi instead of a SWarp process."""
from pathlib import Path

import numpy as np

from . import fits as _fits
from .file import File, UnmappedFileError
from .wcs import WCS

__all__ = ['FITSFile', 'HasWCS']


class FITSFile(File):
    """Maps a single-HDU FITS file (``zuds/fitsfile.py:19-211``)."""

    header = None
    header_comments = None
    __diskmapped_cached_properties__ = ['_path', '_data']
    _DATA_HDU = 0
    _HEADER_HDU = 0

    @classmethod
    def from_file(cls, f, use_existing_record=True):
        """Create an object mapped to ``f`` and load its header
        (``zuds/fitsfile.py:39-67``)."""
        f = Path(f)
        obj = cls()
        obj.basename = f.name
        obj.map_to_local_file(str(f.absolute()))
        obj.load_header()
        return obj

    def load_header(self):
        """Header -> dict of int / str / bool / float values plus comments
        (``zuds/fitsfile.py:69-84``)."""
        _, hd, hdc = _fits.read(self.local_path, header_only=True)
        self.header = hd
        self.header_comments = hdc

    def load_data(self):
        """uint8 data are booleans (``zuds/fitsfile.py:86-94``)."""
        data, _, _ = _fits.read(self.local_path)
        if data is not None and data.dtype.name == 'uint8':
            data = data.astype(bool)
        self._data = data

    def unload_data(self):
        try:
            del self._data
        except AttributeError:
            raise RuntimeError(f'Object "<{self.__class__.__name__} at {hex(id(self))}>" '
                               f'has no data loaded. Load some data with .load_data() and '
                               f'try again.')

    @property
    def data(self):
        try:
            return self._data
        except AttributeError:
            self.load_data()
        return self._data

    @data.setter
    def data(self, d):
        self._data = d

    @property
    def astropy_header(self):
        """The header as an ordered dict of cards (the reference returns an
        ``astropy.io.fits.Header``, ``zuds/fitsfile.py:125-144``)."""
        if self.header is None or self.header_comments is None:
            raise AttributeError('This image does not have a header or header comments '
                                 'record yet. Map it to a file, call .load_header() and retry.')
        return dict(self.header)

    def save(self):
        """Write data + header to the mapped file, then drop the cached data
        (``zuds/fitsfile.py:146-206``)."""
        try:
            f = self.local_path
        except UnmappedFileError:
            f = self.basename
            self.map_to_local_file(f)
        data = self.data
        if data.dtype.name == 'bool':
            data = data.astype('uint8')
        _fits.write(f, data, self.header or {}, self.header_comments or {})
        self.unload_data()

    def load(self):
        self.load_header()
        self.load_data()


class HasWCS(FITSFile):
    """A FITS file with a WCS solution (``zuds/fitsfile.py:229-314``)."""

    @property
    def wcs(self):
        return WCS.from_header(self.astropy_header)

    @classmethod
    def from_file(cls, fname, use_existing_record=True):
        self = super(HasWCS, cls).from_file(fname, use_existing_record=use_existing_record)
        try:
            w = self.wcs
        except (ValueError, KeyError):
            return self          # no WCS cards (e.g. a bare noise map)
        corners = w.calc_footprint()
        for i, values in enumerate(corners):
            setattr(self, f'ra{i + 1}', float(values[0]))
            setattr(self, f'dec{i + 1}', float(values[1]))
        naxis1 = self.header['NAXIS1']
        naxis2 = self.header['NAXIS2']
        ra, dec = w.all_pix2world([naxis1 / 2], [naxis2 / 2], 1)
        self.ra, self.dec = float(ra[0]), float(dec[0])
        return self

    @property
    def pixel_scale(self):
        """Pixel scales along x and y in arcsec (``zuds/fitsfile.py:276-288``)."""
        return self.wcs.proj_plane_pixel_scales() * 3600.0

    def aligned_to(self, other, persist_aligned=False, tmpdir='/tmp', nthreads=1):
        """A version of this object resampled pixel-by-pixel onto the grid of
        ``other`` (``zuds/fitsfile.py:290-314``)."""
        from .swarp import run_align
        if not isinstance(other, HasWCS):
            raise ValueError(f'WCS Alignment target must be an instance of '
                             f'HasWCS (got "{other.__class__}").')
        new = run_align(self, other, tmpdir=tmpdir, nthreads=nthreads,
                        persist_aligned=persist_aligned)
        if getattr(self, 'mask_image', None) is not None:
            newmask = run_align(self.mask_image, other, tmpdir=tmpdir, nthreads=nthreads,
                                persist_aligned=persist_aligned)
            new.mask_image = newmask
        return new
