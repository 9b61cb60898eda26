# developer: pool and pipelined rates with the throughput form of the factorisation, one box
mkdir -p gpurun_out/$1
python bench.py --steps 24 --warmup 3 --no-clocks --no-cpu-baseline --no-secondary --nightly-pools 1,3,4,8,12,16 --nightly-jobs 32 > gpurun_out/$1/b.json 2> gpurun_out/$1/b.err || { tail -5 gpurun_out/$1/b.err; exit 1; }
python - <<P
import json
d=json.loads(open('gpurun_out/$1/b.json').read().strip().splitlines()[-1])
print('step', round(d['ms_per_step'],3), 'hp_solve', round(d['kernels']['hp_solve']['ms_per_step'],3), 'sub', round(d['legs']['subtract_ms'],3))
print('  nightly', {k: (round(v['ms_per_subtraction'],3), v['failed']) for k,v in d['nightly']['pools'].items()})
print('  pipelined', {k: d['pipelined'][k] for k in ('ms_per_step','mpix_s','status_ok','subtractions_in_flight')})
P
