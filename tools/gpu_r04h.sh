#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r04m}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_subtract_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
for st in 32 64; do
ZM_CHOL_STEP=$st timeout -k 10 300 python3 $B > $out/bench$st.json 2> $out/bench$st.err || { tail -20 $out/bench$st.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench$st.json') if l.startswith('{')][-1])
print('step $st', round(d['value']), round(d['ms_per_step'], 3), round(d['legs']['subtract_ms'], 3), round(d['kernels']['hp_solve']['ms_per_step'], 3), round(d['kernels']['hp_chol']['avg_us'], 1))"
done
ZM_CHOL_STEP=32 ZM_CHOL_PROF=1 timeout -k 10 300 python3 tools/chol_prof.py 2>&1 | grep "chol wg" | tail -5
