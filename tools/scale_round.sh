#!/bin/bash
# The first 8-GPU box in one command: bench.py at 1, 2, 4 and 8 ranks for the two exchanges of
# SURVEY.md 8(e) - WEIGHTED (RCCL all-reduce of the two partial-sum planes) and exact CLIPPED (row-band all-to-all of
# the resampled stacks) - then the 8-rank WEIGHTED step once more with the reductions called from inside libzudsmi
# (ZM_NATIVE_RCCL=1, csrc/comm.hip), each line's per-rank step time and per-exchange wall clock collected into ONE
# table with the figures the design expects beside them (tools/scale_table.py).
#   usage (on an N-GPU node, from the repo root):  bash tools/scale_round.sh [tag] [max ranks, default: all GPUs]
# ZM_SCALE_ARGS: extra bench.py arguments.  The rehearsal that was run (round 6, one card, two gloo ranks, 3.5 min):
#   ZM_DIST_BACKEND=gloo ZM_SCALE_ARGS="--size 2048 --frames 8 --no-subtract" ZM_SCALE_STEPS=3 bash tools/scale_round.sh rehearsal 2
# (--no-subtract: a 2048-px frame cut into the full-size job's 3 x 3 regions has regions without usable stamps, which
# bench.py reports as a failure; the exchanges being rehearsed are the coadd's).
# Nothing here runs on the one-GPU boxes of the build rounds except the rehearsal: `ZM_DIST_BACKEND=gloo bash
# tools/scale_round.sh rehearsal 2` puts two ranks on one card over gloo (tests/test_bench_ranks_gpu.py does the same).
set -o pipefail
tag=${1:-scale}
ngpu=$(python3 -c "import torch; print(torch.cuda.device_count())")
maxn=${2:-$ngpu}
out=gpurun_out/$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
STEPS=${ZM_SCALE_STEPS:-10}
run() {   # run <name> <ranks> <env...> -- <bench args...>
    local name=$1 n=$2; shift 2
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    local port=$((29600 + RANDOM % 300))
    if [ "$n" = 1 ]; then
        env "${envs[@]}" timeout -k 10 900 python3 bench.py --gpus 1 --steps $STEPS --warmup 3 --no-clocks --no-cpu-baseline --no-nightly --no-pipelined --no-secondary $ZM_SCALE_ARGS "$@" \
            > $out/$name.json 2> $out/$name.err
    else
        env "${envs[@]}" timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
            bench.py --gpus $n --steps $STEPS --warmup 3 --no-clocks --no-cpu-baseline --no-nightly --no-pipelined --no-secondary $ZM_SCALE_ARGS "$@" \
            > $out/$name.json 2> $out/$name.err
    fi
    local rc=$?
    [ $rc = 0 ] || { echo "$name: exit $rc"; tail -15 $out/$name.err; }
    return $rc
}
for n in 1 2 4 8; do
    [ $n -le $maxn ] || continue
    run weighted_$n $n -- || exit 1
    run clipped_$n $n -- --combine CLIPPED || exit 1
done
if [ $maxn -ge 2 ] && [ "${ZM_DIST_BACKEND:-nccl}" = nccl ]; then
    n=$maxn; [ $n -gt 8 ] && n=8
    run weighted_native_$n $n ZM_NATIVE_RCCL=1 -- || exit 1
fi
python3 tools/scale_table.py $out | tee $out/scale_table.txt
