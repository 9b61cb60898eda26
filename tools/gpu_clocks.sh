#!/bin/bash
# Developer tool: the data-movement clocks of the bench line alone (PCIe overlap, FITS, object API warm / cold).
out=gpurun_out/${1:-clocks}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-nightly --no-pipelined > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1]); c = d['clocks']
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in c.items() if k not in ('pcie', 'object_api_ms', 'fits')})
print('pcie', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in c['pcie'].items() if k != 'overlap'})
print('object_api', json.dumps(c.get('object_api_ms'), indent=1))"
