#!/bin/bash
# round 5: k_coadd_fused_own - bit-identity tests, then (ZM_FF_DEAL, ZM_FF_PRIO) variants and per-wave phase clocks
# on one box.   bash tools/gpu_own_prio.sh <tag> "<deal:prio list>" "<deal:prio list for the per-wave clocks>"
set -o pipefail
out=gpurun_out/${1:-r05b}
combos=${2:-"0:0 1:0"}
profs=${3:-"1:0"}
mkdir -p $out
export TMPDIR=/tmp
for c in ${4:-"1:0"}; do
    d=${c%%:*}; pr=${c##*:}
    ZM_FF_DEAL=$d ZM_FF_PRIO=$pr timeout -k 10 600 python3 -m pytest tests/test_fused_coadd_gpu.py tests/test_mask_i16_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $out/tests_${d}_$pr.log 2>&1 || { tail -40 $out/tests_${d}_$pr.log; exit 1; }
    echo "tests deal $d prio $pr: $(tail -1 $out/tests_${d}_$pr.log)"
    ZM_FF_DEAL=$d ZM_FF_PRIO=$pr timeout -k 10 600 python3 tools/fuzz_coadd.py 30 79 > $out/fuzz.log 2>&1 || { tail -20 $out/fuzz.log; exit 1; }
    tail -1 $out/fuzz.log
done
for pass in 1 2 3; do
    echo "== dma (pass $pass)  $(ZM_FF_FORM=dma timeout -k 10 200 python3 tools/ff_probe.py --dbg 0 2>&1 | grep ZM_FF_DBG | cut -c1-90)"
    for c in $combos; do
        d=${c%%:*}; pr=${c##*:}
        echo "== own, deal $d prio $pr (pass $pass)  $(ZM_FF_DEAL=$d ZM_FF_PRIO=$pr timeout -k 10 200 python3 tools/ff_probe.py --dbg 0 2>&1 | grep ZM_FF_DBG | cut -c1-90)"
    done
done
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
for c in $profs; do
    d=${c%%:*}; pr=${c##*:}
    ZM_FF_DEAL=$d ZM_FF_PRIO=$pr ZM_FF_PROF=2 timeout -k 10 300 python3 $B --no-subtract --steps 2 > $out/prof_own_${d}_${pr}.json 2> $out/prof_own_${d}_$pr.err
    echo "-- per-wave phase clocks, deal $d prio $pr"
    grep -A8 phases $out/prof_own_${d}_$pr.err | tail -9
done
