#!/bin/bash
# round 4, call a: GPU tests + A / B of the mask box kernels (int16 / int32 streaming, int32 tiled)
set -o pipefail
out=gpurun_out/${1:-r04a}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
timeout -k 10 300 python3 $B > $out/bench_i16.json 2> $out/bench_i16.err || { tail -20 $out/bench_i16.err; exit 1; }
timeout -k 10 300 python3 $B --mask-dtype int32 > $out/bench_i32.json 2> $out/bench_i32.err || { tail -20 $out/bench_i32.err; exit 1; }
ZM_MASK_BOX=tile timeout -k 10 300 python3 $B --mask-dtype int32 > $out/bench_i32_tile.json 2> $out/bench_i32_tile.err || { tail -20 $out/bench_i32_tile.err; exit 1; }
for f in i16 i32 i32_tile; do
python3 -c "
import json; d = json.loads([l for l in open('$out/bench_$f.json') if l.startswith('{')][-1])
print('$f', round(d['value']), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['legs'].items()})
print({k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()})"
done
