#!/bin/bash
# Developer tool: the PCIe-overlapped step (clocks.with_pcie_ms) under the switches that could make the copy stream
# wait for the compute streams.   bash tools/pcie_bisect.sh <tag>
set -o pipefail
out=gpurun_out/${1:-pcie}
mkdir -p $out
export TMPDIR=/tmp ZM_BENCH_PCIE_ONLY=1
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
run() {
    name=$1; shift
    env "$@" timeout -k 10 300 python3 $B > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; return; }
    python3 -c "
import json; d = json.loads([l for l in open('$out/$name.json') if l.startswith('{')][-1]); c = d['clocks']
print('$name', 'device', round(c['device_ms'], 2), 'with_pcie', c.get('with_pcie_ms') and round(c['with_pcie_ms'], 1), 'copy alone', round(c['pcie']['h2d_copy_alone_ms'], 1), 'ratio', round(c['pcie']['ratio_to_max_of_copy_and_device'], 3))"
}
run default ZM_X=0
run nofork ZM_FF_FORK=0
run hostlimits ZM_HOST_LIMITS=1
run q4 GPU_MAX_HW_QUEUES=4
run q16 GPU_MAX_HW_QUEUES=16
run nofork_hostlimits ZM_FF_FORK=0 ZM_HOST_LIMITS=1
run default2 ZM_X=0
