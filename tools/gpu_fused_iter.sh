#!/bin/bash
# round 4, fused-kernel iteration: the tests that pin k_coadd_fused_dma bit for bit, then the short bench line
set -o pipefail
out=gpurun_out/${1:-r04b}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_fused_coadd_gpu.py tests/test_mask_i16_gpu.py tests/test_coadd_gpu.py tests/test_configs_gpu.py tests/test_fuzz_oracle_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
timeout -k 10 300 python3 $B > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(round(d['value']), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['legs'].items()})
print({k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()})"
ZM_FF_PROF=1 timeout -k 10 300 python3 $B --no-subtract --steps 2 > $out/prof.json 2> $out/prof.err; grep "phases" $out/prof.err | tail -2
