"""Developer tool: phase clocks of the one-workgroup-per-region factorisation (k_chol_tp) on the
config-2 system (9 regions x 722 unknowns).  usage: ZM_CHOL_PROF=1 ZM_CHOL_FORM=tp python3 tools/chol_tp_prof.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('ZM_CHOL_PROF', '1')
os.environ.setdefault('ZM_CHOL_FORM', 'tp')
import importlib  # noqa: E402

from test_subtract_gpu import COMMON, scene  # noqa: E402

z = importlib.import_module('zuds-pipeline_amd')
eng = z.Engine(0)
data = scene(nx=1024, ny=1024, seed=50, nstars=1200, gradient=0.3, nbad=30)
for _ in range(2):
    d, n, info = eng.subtract(*data, r=10.0, rss=24.0, nsx=10, nsy=10, nrx=3, nry=3, ko=4, bgo=0, **COMMON)
print(info['status'], info['ncoeff'], info['niter'])
