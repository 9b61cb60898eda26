#!/bin/bash
# Developer tool: the long differential / oracle fuzz runs at the final sources of a round, one GPU call.
#   bash tools/gpu_fuzz_long.sh <tag> [ncoadd] [nsubtract] [oracle seeds]
out=gpurun_out/${1:-fuzz}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/fuzz_coadd.py ${2:-1000} 5005 > $out/fuzz_coadd.log 2>&1; echo "fuzz_coadd: $(tail -1 $out/fuzz_coadd.log)"
timeout -k 10 500 python3 tools/fuzz_subtract.py ${3:-400} 5006 > $out/fuzz_subtract.log 2>&1; echo "fuzz_subtract: $(tail -1 $out/fuzz_subtract.log)"
ZM_FUZZ_SEEDS=${4:-100} timeout -k 10 700 python3 -m pytest tests/test_fuzz_oracle_gpu.py -m gpu -q > $out/fuzz_oracle.log 2>&1; echo "fuzz_oracle (ZM_FUZZ_SEEDS=${4:-100}): $(tail -1 $out/fuzz_oracle.log)"
