#!/bin/bash
# Developer tool: A/B of two library builds (tools/_build/lib_<name>.so) on ONE box for the pool: the bench's nightly
# leg (32 jobs: pools of 1 .. 16 workers, batched lanes) and the subtraction tests that compare the solver forms.
#   gpurun -- 'bash tools/ab_pool.sh "head ct256" 2'
names=${1:-"head new"}
reps=${2:-2}
for v in $names; do
    cp tools/_build/lib_$v.so zuds-pipeline_amd/lib/libzudsmi.so || exit 1
    echo "== $v: solver forms"
    timeout -k 10 600 python3 -m pytest tests/test_subtract_gpu.py -m gpu -x -q -k "throughput_form or batch or timeout or two_engines" 2>&1 | tail -1
done
for r in $(seq $reps); do
    for v in $names; do
        cp tools/_build/lib_$v.so zuds-pipeline_amd/lib/libzudsmi.so || exit 1
        timeout -k 10 500 python3 bench.py --steps 5 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-pipelined 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); n = d['nightly']
print('== $v (pass $r)', {k: round(v['ms_per_subtraction'], 2) for k, v in n['pools'].items()}, {k: round(v['ms_per_subtraction'], 2) for k, v in n['batched'].items()}, 'best', round(n['batched_best']['ms_per_subtraction'], 3))"
    done
done
