"""Small host-side helpers the coadd/subtraction classes share.  Same names and
argument meaning as the reference's ``zuds/utils.py`` so calling code reads the
same; the background estimate runs on the device (radix select in libzudsmi)."""
import datetime
import os

import numpy as np

__all__ = ['initialize_directory', 'quick_background_estimate', 'fid_map', '_split',
           'print_time', 'ensure_images_have_the_same_properties', 'get_time']

fid_map = {1: 'zg', 2: 'zr', 3: 'zi'}

_MJD_EPOCH = datetime.datetime(1858, 11, 17)
_JD_MINUS_MJD = 2400000.5

# header keyword -> how its value converts to MJD; searched in this order
# (the order of zuds/utils.py:11-25)
_DATE_KEYS = (('OBSMJD', 'mjd'), ('MJD-OBS', 'mjd'), ('OBSJD', 'jd'), ('JD-OBS', 'jd'),
              ('DATE-OBS', 'iso'), ('UTC-OBS', 'iso'), ('OBSDATE', 'iso'))


def _iso_to_mjd(text):
    """UTC ISO date ('YYYY-MM-DD', optionally with 'T' or ' ' and a time) as MJD."""
    text = str(text).strip().replace('T', ' ')
    if ' ' not in text:
        pattern = '%Y-%m-%d'
    elif '.' in text:
        pattern = '%Y-%m-%d %H:%M:%S.%f'
    else:
        pattern = '%Y-%m-%d %H:%M:%S'
    delta = datetime.datetime.strptime(text, pattern) - _MJD_EPOCH
    return delta.total_seconds() / 86400.0


def _to_mjd(value, kind):
    if kind == 'iso':
        return _iso_to_mjd(value)
    return float(value) - (_JD_MINUS_MJD if kind == 'jd' else 0.0)


def get_time(image, format):
    """Observation epoch of ``image`` as ``format`` = 'mjd' or 'jd', from the first
    date keyword its header has; ValueError when it has none."""
    if format not in ('mjd', 'jd'):
        raise ValueError(f'unsupported time format "{format}"')
    hit = next(((k, kind) for k, kind in _DATE_KEYS if k in image.header), None)
    if hit is None:
        raise ValueError(f'No matching keys found for image "{image.basename}"')
    mjd = _to_mjd(image.header[hit[0]], hit[1])
    return mjd + _JD_MINUS_MJD if format == 'jd' else mjd


def initialize_directory(directory):
    os.makedirs(directory, exist_ok=True)


def quick_background_estimate(image, nsamp=None, mask_image=None):
    """(median, 1.4826 * MAD) over the pixels whose mask is 0, optionally over a
    random sample of ``nsamp`` of them (``zuds/utils.py:32-53``)."""
    from .engine import get_engine
    engine = get_engine()
    mask = (mask_image if mask_image is not None else image.mask_image).data
    if nsamp is None:
        return engine.median_mad(image.data, mask)
    good = image.data[mask == 0]
    return engine.median_mad(np.random.choice(good, size=nsamp), None)


def _split(iterable, n):
    """Cut ``iterable`` into ``n`` consecutive pieces, sizes as the reference's
    recursive lambda gives them (``zuds/utils.py:63-65``): piece k takes
    floor(remaining / (n - k)) items."""
    pieces, rest = [], iterable
    for left in range(n, 0, -1):
        cut = len(rest) // left
        pieces.append(rest[:cut])
        rest = rest[cut:]
    return pieces


def print_time(start, stop, detection, step):
    print(f'took {stop - start:.2f} sec to do {step} for {detection.id}', flush=True)


def ensure_images_have_the_same_properties(images, properties):
    """ValueError unless every image agrees with the first on each property."""
    for name in properties:
        values = np.asarray([getattr(im, name, None) for im in images])
        if not all(values == values[0]):
            listing = [(getattr(im, 'basename', None), getattr(im, name, None)) for im in images]
            raise ValueError(f'To be coadded, images must all have the same {name}. '
                             f'These images had: {listing}.')
