"""Developer: where does k_coadd_fused_own differ from the k_resample path?  One stack, every ZM_FF_DEAL."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from test_fused_coadd_gpu import run_both, stack  # noqa: E402
from util import pkg  # noqa: E402

z = pkg()
eng = z.get_engine(0)
for back in (False, True):
    frames, wout = stack(6, 700, 650, 100)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='AND', subtract_back=back, rescale_weights=back, back_size=128)
    for deal in sys.argv[1:] or ['0', '1', '2']:
        os.environ['ZM_FF_DEAL'] = deal
        for rep in range(2):
            a, b = run_both(eng, frames, wout, p)
            bad = np.argwhere(a[0] != b[0])
            msg = f'background {back} deal {deal} rep {rep}: {len(bad)} px differ'
            if len(bad):
                y, x = bad[:, 0], bad[:, 1]
                msg += f'; rows {y.min()}..{y.max()} cols {x.min()}..{x.max()}; row mod 32 hist {np.bincount(y % 32, minlength=32).tolist()}'
                msg += f'; col mod 64 nonzero {np.flatnonzero(np.bincount(x % 64, minlength=64)).tolist()[:70]}'
                msg += f'; tiles {sorted(set(zip((y // 32).tolist(), (x // 64).tolist())))[:12]}'
                d = np.abs(a[0] - b[0])[a[0] != b[0]]
                msg += f'; |diff| median {np.median(d):.3g} max {d.max():.3g}'
            print(msg, flush=True)
