"""Developer tool: phase clocks of k_chol_fused on a config-3 sized subtraction."""
import importlib
import os
import sys

import numpy as np

os.environ.setdefault('ZM_CHOL_PROF', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z = importlib.import_module('zuds-pipeline_amd')
s = importlib.import_module('zuds-pipeline_amd.synth')
from scipy.ndimage import gaussian_filter

N = 3072
rng = np.random.default_rng(1)
ref = np.full((N, N), 150.0)
s.add_stars(ref, rng.uniform(20, N - 20, 3000), rng.uniform(20, N - 20, 3000),
            np.exp(rng.uniform(np.log(2e3), np.log(5e4), 3000)), 2.2)
sci = (1.3 * gaussian_filter(ref, 1.0) + 10 + rng.normal(0, 4, ref.shape)).astype(np.float32)
ref = (ref + rng.normal(0, 1, ref.shape)).astype(np.float32)
rms = np.full((N, N), 4.0, np.float32)
eng = z.get_engine(0)
d, n, info = eng.subtract(sci, rms, ref, np.ones_like(rms), None, r=10.0, rss=24.0, nsx=10, nsy=10, nrx=3,
                          nry=3, ko=4, bgo=0, tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
print(info)
