"""Constants of the hot path, same names and values as ``zuds/constants.py``
(only the ones the coadd / subtraction path reads)."""
import numpy as np

BIG_RMS = np.sqrt(50000.)          # zuds/constants.py:3
BKG_BOX_SIZE = 128                 # zuds/constants.py:4
MJD_TO_JD = 2400000.5
APER_KEY = 'APCOR4'                # zuds/constants.py:13
APERTURE_RADIUS = 3.0              # pixels, zuds/constants.py:14
GROUP_PROPERTIES = ['field', 'ccdid', 'qid', 'fid']
NTHREADS_PER_NODE = 64
MASK_BORDER = 10                   # pix, zuds/constants.py:22
BKG_VAL = 150.                     # counts, zuds/constants.py:23

MASK_BITS = {f'BIT{i:02d}': i for i in range(17)}   # zuds/constants.py:25-43

BAD_BITS = np.asarray([0, 2, 3, 4, 5, 7, 8, 9, 10, 16, 17])   # zuds/constants.py:45
BAD_SUM = int(np.sum(2 ** BAD_BITS))                           # = 198589

MASK_COMMENTS = {
    'BIT00': 'AIRCRAFT/SATELLITE TRACK',
    'BIT01': 'CONTAINS SEXTRACTOR DETECTION',
    'BIT02': 'LOW RESPONSIVITY',
    'BIT03': 'HIGH RESPONSIVITY',
    'BIT04': 'NOISY',
    'BIT05': 'GHOST FROM BRIGHT SOURCE',
    'BIT06': 'RESERVED FOR FUTURE USE',
    'BIT07': 'PIXEL SPIKE (POSSIBLE RAD HIT)',
    'BIT08': 'SATURATED',
    'BIT09': 'DEAD (UNRESPONSIVE)',
    'BIT10': 'NAN (not a number)',
    'BIT11': 'CONTAINS PSF-EXTRACTED SOURCE POSITION',
    'BIT12': 'HALO FROM BRIGHT SOURCE',
    'BIT13': 'RESERVED FOR FUTURE USE',
    'BIT14': 'RESERVED FOR FUTURE USE',
    'BIT15': 'RESERVED FOR FUTURE USE',
    'BIT16': 'NON-DATA SECTION FROM SWARP ALIGNMENT'
}

REFERENCE_VERSION = 'zuds5'
