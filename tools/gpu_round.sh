#!/bin/bash
# Developer tool: one consolidated GPU call - tests, the default bench line, the kernel trace of the
# bench and the counter passes behind the rooflines.  usage (through gpurun, from the repo root):
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
    timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
    tail -2 $out/tests.log
fi
# a short bench line first: the size / frames / step time the counter profile records beside its figures
timeout -k 10 300 python3 $B > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python3 $B > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
db=$(find $out/prof -name '*results.db' | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db $out/kernel_stats.csv --top 14
# counters: FETCH_SIZE, WRITE_SIZE and the SQ set each in a pass of their own (they do not fit one pass), for the
# headline command and for its COMBINE_TYPE CLIPPED form (the reference's default operator: `clipped` of the line)
for mode in weighted clipped; do
    [ $mode = clipped ] && X="--combine CLIPPED" || X=""
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${mode}_$c -o pmc --output-format csv -- python3 $B $X > $out/pmc_${mode}_$c.log 2>&1 || { tail -20 $out/pmc_${mode}_$c.log; exit 1; }
    done
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $out/pmc_${mode}_SQ -o pmc --output-format csv -- python3 $B $X > $out/pmc_${mode}_SQ.log 2>&1 || { tail -20 $out/pmc_${mode}_SQ.log; exit 1; }
done
python3 tools/pmc_summary.py $(find $out/pmc_weighted_* -name '*counter_collection.csv') > $out/pmc_summary.txt
python3 tools/pmc_summary.py $(find $out/pmc_clipped_* -name '*counter_collection.csv') > $out/pmc_summary_clipped.txt
grep -E "k_coadd_fused|k_mask_box|k_mesh_stats|k_chol_df|k_hp_apply" $out/pmc_summary.txt | grep -E "FETCH|WRITE|INSTS_VALU"
if [ "${2:-tests}" = tests ]; then tail -3 $out/tests.log > $out/tests_tail.log; else echo "tests not run in this call (notests mode)" > $out/tests_tail.log; fi
grep -E "k_coadd_fused|k_combine" $out/pmc_summary_clipped.txt | grep -E "FETCH|WRITE"
# the counter profile bench.py quotes, stamped with the hash of these kernel sources (copy it to profiles/)
python3 tools/make_pmc_json.py $out $out/pmc.json
# ... and the full bench line LAST, with this round's counter profile in the place bench.py reads it from (on
# this box's copy of the tree), so that the committed line quotes counters taken at its own commit
cp $out/pmc.json profiles/r06_pmc.json
cp $out/bench.json $out/bench_short.json
rm -rf $out/pmc_*_FETCH_SIZE $out/pmc_*_WRITE_SIZE $out/pmc_*_SQ $out/prof    # (raw traces: tens of MB)
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['legs'])
print('roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'traffic', 'avg_launch_us', 'valu_insts_per_px', 'traffic_over_algorithmic')}, d['roofline'].get('leg'))
print('solve', d.get('solve_roofline'))
print('clipped', {k: v for k, v in (d.get('clipped') or {}).items() if k != 'band_combine'})
print({k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})
print(d.get('nightly'))"
