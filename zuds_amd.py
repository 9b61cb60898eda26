"""Import shim: ``import zuds_amd as zuds`` gives the ``zuds-pipeline_amd``
package (whose directory name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('zuds-pipeline_amd')
sys.modules[__name__] = _pkg
