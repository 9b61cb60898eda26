"""Developer tool: per-kernel timings of a short coadd-only bench run."""
import json
import os
import subprocess
import sys

extra = [a for a in sys.argv[1:] if a.startswith('--')]
for st in ([a for a in sys.argv[1:] if not a.startswith('--')] or ['0']):
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline', '--no-subtract', '--frames', '16'] + extra,
                         capture_output=True, text=True, env=dict(os.environ, ZM_DBG_BK=st)).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
    print(st, {k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})
