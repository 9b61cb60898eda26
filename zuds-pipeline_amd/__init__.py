"""zuds-pipeline_amd: the resample -> coadd -> subtract hot path of
zuds-survey/zuds-pipeline on MI355X (gfx950), behind the ZUDS object API.

The directory name is not a Python identifier; import it with
``importlib.import_module('zuds-pipeline_amd')`` or through the ``zuds_amd``
shim at the repository root (``import zuds_amd as zuds``).
"""
from . import _lib
from ._lib import ZMError
from .wcs import WCS
from .engine import Engine, get_engine, coadd_params, hp_params
from .constants import *
from .file import *
from .fitsfile import *
from .image import *
from .mask import *
from .utils import *
from .swarp import *
from .sextractor import *
from .hotpants import *
from .coadd import *
from .subtraction import *
from .mpi import *
from .photometry import *
from .seeing import *
from .filterobjects import *
from . import synth, fits

# same DB-free entry points as the reference
def init_db(*args, **kwargs):
    """The reference binds a PostgreSQL session here (``zuds/model_util.py``);
    this package keeps no database, the call is accepted and ignored."""
    return None
