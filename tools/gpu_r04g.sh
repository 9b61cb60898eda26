#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r04j}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/chol_prof.py > $out/prof.log 2>&1 || { tail -30 $out/prof.log; exit 1; }
grep "chol wg" $out/prof.log | tail -10
