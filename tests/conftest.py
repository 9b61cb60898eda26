import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def zuds():
    import zuds_amd
    return zuds_amd


@pytest.fixture(scope='session')
def engine(zuds):
    """Process-wide engine; raises (never skips) when the HIP library or the
    GPU is missing, so a gpu-marked test cannot pass on a silent fallback."""
    return zuds.get_engine(0)
