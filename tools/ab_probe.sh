#!/bin/bash
# Developer tool: A/B two builds of libzudsmi.so on ONE box with the fused-kernel probe (tools/ff_probe.py).
#   gpurun -- 'bash tools/ab_probe.sh "prev new" 3 [tests ...]'      (libraries: tools/_build/lib_<name>.so)
names=${1:-"prev new"}
reps=${2:-3}
shift 2
tests=${@:-"tests/test_fused_coadd_gpu.py tests/test_mask_i16_gpu.py tests/test_configs_gpu.py"}
last=""
for r in $(seq $reps); do
    for v in $names; do
        cp tools/_build/lib_$v.so zuds-pipeline_amd/lib/libzudsmi.so || exit 1
        echo "== $v (pass $r)  $(timeout -k 10 200 python3 tools/ff_probe.py --dbg 0 2>&1 | grep ZM_FF_DBG | cut -c1-150)"
        last=$v
    done
done
timeout -k 10 900 python3 -m pytest $tests -m gpu -x -q 2>&1 | tail -3
echo "(tests ran on lib_$last)"
