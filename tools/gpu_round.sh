#!/bin/bash
# Developer tool: one consolidated GPU call - tests, the default bench line, the kernel trace of the
# bench and the counter passes of the roofline kernel.  usage (through gpurun, from the repo root):
#   bash tools/gpu_round.sh <tag> [tests|notests]
# Order: kernel trace and counter passes first, then the full bench line, which quotes this round's counters.
# Everything lands under gpurun_out/<tag>/ ; the summaries to be judged are copied into profiles/ by hand.
set -o pipefail
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 20 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
if [ "${2:-tests}" = tests ]; then
    timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
    tail -2 $out/tests.log
fi
# a short bench line first: the size / frames / step time the counter profile records beside its figures
timeout -k 10 300 python3 $B > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python3 $B > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
db=$(find $out/prof -name '*results.db' | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db $out/kernel_stats.csv --top 14
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc --output-format csv -- python3 $B > $out/pmc_$c.log 2>&1 || { tail -20 $out/pmc_$c.log; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $out/pmc_SQ -o pmc --output-format csv -- python3 $B > $out/pmc_SQ.log 2>&1 || { tail -20 $out/pmc_SQ.log; exit 1; }
python3 tools/pmc_summary.py $(find $out/pmc_* -name '*counter_collection.csv') > $out/pmc_summary.txt
grep -E "k_coadd_fused|k_mask_box|k_mesh_stats|k_chol|k_hp_apply" $out/pmc_summary.txt
# the counter profile bench.py quotes, stamped with the hash of these kernel sources (copy it to profiles/)
python3 tools/make_pmc_json.py $out $out/pmc_coadd_fused.json
# ... and the full bench line LAST, with this round's counter profile in the place bench.py reads it from (on
# this box's copy of the tree), so that the committed line quotes counters taken at its own commit
cp $out/pmc_coadd_fused.json profiles/r03_pmc_coadd_fused.json
cp $out/bench.json $out/bench_short.json
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['legs'], d['roofline'])
print({k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})
print(d.get('nightly'))"
