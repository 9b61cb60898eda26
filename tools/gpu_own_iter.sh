#!/bin/bash
# round 5, fused-kernel iteration: the tests that pin the fused coadd bit for bit (k_coadd_fused_own where the
# footprints fit its fixed slot, k_coadd_fused_dma elsewhere), the differential fuzzer, then the two forms side by
# side on ONE box (tools/ff_probe.py with ZM_FF_FORM=dma / own, ablations of the new one), phase clocks, short bench.
#   bash tools/gpu_own_iter.sh <tag> [nfuzz]
set -o pipefail
out=gpurun_out/${1:-r05a}
nf=${2:-150}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_fused_coadd_gpu.py tests/test_mask_i16_gpu.py tests/test_coadd_gpu.py tests/test_configs_gpu.py tests/test_fuzz_oracle_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
timeout -k 10 600 python3 tools/fuzz_coadd.py $nf 505 > $out/fuzz.log 2>&1 || { tail -20 $out/fuzz.log; exit 1; }
tail -2 $out/fuzz.log
for pass in 1 2; do
    for form in dma own; do
        echo "== $form (pass $pass)"
        ZM_FF_FORM=$form timeout -k 10 200 python3 tools/ff_probe.py --dbg 0 2>&1 | grep ZM_FF_DBG || exit 1
    done
done
echo "== own: ablations (1 no pixels, 2 no prep, 4 no staging loads)"
ZM_FF_FORM=own timeout -k 10 300 python3 tools/ff_probe.py --dbg 0,1,2,4,6,7 2>&1 | grep ZM_FF_DBG || exit 1
B="bench.py --steps 10 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined"
for form in dma own; do
    ZM_FF_FORM=$form ZM_FF_PROF=1 timeout -k 10 300 python3 $B --no-subtract --steps 2 > $out/prof_$form.json 2> $out/prof_$form.err
    echo "$form: $(grep phases $out/prof_$form.err | tail -1)"
done
timeout -k 10 300 python3 $B > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print(round(d['value']), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['legs'].items()})
print({k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()})
print('copy ceiling', d['roofline'].get('copy_ceiling'))"
timeout -k 10 300 python3 $B --combine CLIPPED --no-subtract > $out/bench_clipped.json 2> $out/bench_clipped.err || { tail -20 $out/bench_clipped.err; exit 1; }
python3 -c "
import json; d = json.loads([l for l in open('$out/bench_clipped.json') if l.startswith('{')][-1])
print('clipped', round(d['ms_per_step'], 3), {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()})"
