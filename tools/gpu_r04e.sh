#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r04h}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nightly --no-pipelined --no-secondary > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
print('clocks', d.get('clocks'))"
