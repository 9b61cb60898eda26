# A / B of the two forms of the kernel-fit factorisation in the pool, one box
mkdir -p gpurun_out/$1
for form in tp lat; do
  pools=1,4,6,8,12,16; [ $form = lat ] && pools=1,3,4
  ZM_CHOL_FORM=$form python bench.py --steps 5 --warmup 2 --no-clocks --no-cpu-baseline --no-secondary --no-pipelined --nightly-pools $pools --nightly-jobs 32 > gpurun_out/$1/b_$form.json 2> gpurun_out/$1/b_$form.err || { tail -5 gpurun_out/$1/b_$form.err; exit 1; }
  python - <<P
import json
d=json.loads(open('gpurun_out/$1/b_$form.json').read().strip().splitlines()[-1])
print('$form', 'step', round(d['ms_per_step'],3), 'hp_solve', round(d['kernels']['hp_solve']['ms_per_step'],3), 'sub', round(d['legs']['subtract_ms'],3))
print('  nightly', {k: (round(v['ms_per_subtraction'],3), v['failed']) for k,v in d['nightly']['pools'].items()})
P
done
