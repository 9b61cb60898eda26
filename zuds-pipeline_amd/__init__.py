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

__all__ = ['ZMError', 'WCS', 'Engine', 'get_engine', 'coadd_params', 'hp_params']
