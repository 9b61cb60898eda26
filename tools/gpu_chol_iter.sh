#!/bin/bash
# round 5, solver iteration: the subtraction's parity tests (oracle, the three solver forms against each other, the
# batch, the pool), the differential fuzzer, the factor probe, a short bench line.
#   bash tools/gpu_chol_iter.sh <tag> [nfuzz]
set -o pipefail
out=gpurun_out/${1:-r05chol}
nf=${2:-100}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_subtract_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py tests/test_device_chain_gpu.py tests/test_golden_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
timeout -k 10 600 python3 tools/fuzz_subtract.py $nf 707 > $out/fuzz.log 2>&1 || { tail -20 $out/fuzz.log; exit 1; }
tail -2 $out/fuzz.log
[ -x tools/_build/potrf_probe ] && timeout -k 10 120 tools/_build/potrf_probe | grep "^form\|differ"
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-pipelined"
timeout -k 10 400 python3 $B > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(round(d['value']), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['legs'].items()})
print({k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()})
print('solve', d.get('solve_roofline', {}).get('avg_us'))
n = d.get('nightly') or {}
print('pool', {k: round(v['ms_per_subtraction'], 3) for k, v in (n.get('pools') or {}).items()}, {k: round(v['ms_per_subtraction'], 3) for k, v in (n.get('batched') or {}).items()}, n.get('batched_best'))"
