#!/bin/bash
# Developer tool: A/B two builds of libzudsmi.so on ONE GPU box (boxes differ by ~2 %, so numbers from
# different gpurun calls do not compare).  Build the variants here, keep each as tools/_build/lib_<name>.so
# (git-ignored, shipped to the box), then:
#   gpurun -- 'bash tools/ab_libs.sh "prev head" 2 --no-clocks --no-secondary --no-nightly --no-pipelined'
# prints the bench summary of every variant, alternating, `reps` times; the in-tree library is restored
# from the LAST name given.
names=${1:-"prev head"}
reps=${2:-2}
shift 2
for r in $(seq $reps); do
    for v in $names; do
        cp tools/_build/lib_$v.so zuds-pipeline_amd/lib/libzudsmi.so || exit 1
        echo "== $v (pass $r)"
        python3 tools/bench_brief.py --steps 20 --warmup 5 "$@" || exit 1
    done
done
