#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r04p}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_subtract_gpu.py tests/test_bench_ranks_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
