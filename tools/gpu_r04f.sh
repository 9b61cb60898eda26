#!/bin/bash
# Cholesky iteration: the subtraction tests (forms compared bit for bit, oracle parity), then the short bench
set -o pipefail
out=gpurun_out/${1:-r04i}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_subtract_gpu.py tests/test_nightly_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
timeout -k 10 300 python3 $B > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
ZM_CHOL_STEP=32 timeout -k 10 300 python3 $B > $out/bench32.json 2> $out/bench32.err || { tail -20 $out/bench32.err; exit 1; }
for f in bench bench32; do python3 -c "
import json; d = json.loads([l for l in open('$out/$f.json') if l.startswith('{')][-1])
print('$f', round(d['value']), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['legs'].items()}, d['config']['hotpants'])
print({k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items() if k.startswith('hp')}, round(d['kernels']['hp_chol']['avg_us'], 1))"; done
