#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r04f}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_bench_line_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-nightly --no-pipelined > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['legs'])
print('roofline', {k: v for k, v in d['roofline'].items() if k not in ('copy_ceiling',)})
print('solve', d.get('solve_roofline'))
print('clipped', {k: v for k, v in (d.get('clipped') or {}).items() if k != 'band_combine'})
print({k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})"
