"""Helpers of the hot path (``zuds/utils.py``)."""
from pathlib import Path

import numpy as np

__all__ = ['initialize_directory', 'quick_background_estimate', 'fid_map', '_split',
           'print_time', 'ensure_images_have_the_same_properties', 'get_time']


def _iso_to_mjd(s):
    """'YYYY-MM-DD[T ]HH:MM:SS[.f]' -> MJD (UTC, proleptic Gregorian)."""
    import datetime
    s = str(s).strip().replace('T', ' ')
    fmt = '%Y-%m-%d %H:%M:%S.%f' if '.' in s else ('%Y-%m-%d %H:%M:%S' if ' ' in s else '%Y-%m-%d')
    t = datetime.datetime.strptime(s, fmt)
    return (t - datetime.datetime(1858, 11, 17)).total_seconds() / 86400.0


def get_time(image, format):
    """Observation date from the first of OBSMJD, MJD-OBS, OBSJD, JD-OBS, DATE-OBS,
    UTC-OBS, OBSDATE found in the header, as 'mjd' or 'jd' (``zuds/utils.py:11-25``)."""
    time_keys = ['OBSMJD', 'MJD-OBS', 'OBSJD', 'JD-OBS', 'DATE-OBS', 'UTC-OBS', 'OBSDATE']
    time_formats = ['mjd', 'mjd', 'jd', 'jd', 'iso', 'iso', 'iso']
    for k, f in zip(time_keys, time_formats):
        if k in image.header:
            v = image.header[k]
            if f == 'mjd':
                mjd = float(v)
            elif f == 'jd':
                mjd = float(v) - 2400000.5
            else:
                mjd = _iso_to_mjd(v)
            if format == 'mjd':
                return mjd
            if format == 'jd':
                return mjd + 2400000.5
            raise ValueError(f'unsupported time format "{format}"')
    raise ValueError(f'No matching keys found for image "{image.basename}"')


def initialize_directory(directory):
    Path(directory).mkdir(parents=True, exist_ok=True)


def quick_background_estimate(image, nsamp=None, mask_image=None):
    """Median and 1.4826 MAD of the pixels whose mask value is 0
    (``zuds/utils.py:32-53``); evaluated by libzudsmi's radix select."""
    from .engine import get_engine
    if mask_image is None:
        mask_image = image.mask_image
    if nsamp is not None:
        bkgpix = image.data[mask_image.data == 0]
        bkgpix = np.random.choice(bkgpix, size=nsamp)
        return get_engine().median_mad(bkgpix, None)
    return get_engine().median_mad(image.data, mask_image.data)


fid_map = {1: 'zg', 2: 'zr', 3: 'zi'}

# split an iterable over some processes recursively (zuds/utils.py:63-65)
_split = lambda iterable, n: [iterable[:len(iterable) // n]] + \
    _split(iterable[len(iterable) // n:], n - 1) if n != 0 else []


def print_time(start, stop, detection, step):
    print(f'took {stop - start:.2f} sec to do {step} for {detection.id}', flush=True)


def ensure_images_have_the_same_properties(images, properties):
    """Raise a ValueError if images differ in any of ``properties``
    (``zuds/utils.py:73-79``)."""
    for prop in properties:
        vals = np.asarray([getattr(image, prop, None) for image in images])
        if not all(vals == vals[0]):
            raise ValueError(f'To be coadded, images must all have the same {prop}. '
                             f'These images had: '
                             f'{[(getattr(i, "basename", None), getattr(i, prop, None)) for i in images]}.')
