# developer A/B on one box: the step with the three-pass select (ZM_RS_CLASSIC=1, -DZM_DEV build) and with the bracketed one
mkdir -p gpurun_out/sel
ZM_HIPCC_FLAGS=-DZM_DEV python -c "
import importlib; b=importlib.import_module('zuds-pipeline_amd.build'); b.build(force=True, verbose=False)" > gpurun_out/sel/ab_build.log 2>&1 || exit 1
for rep in 1 2; do
for c in 1 0; do
  ZM_RS_CLASSIC=$c python bench.py --no-cpu-baseline --no-clocks --no-nightly --no-secondary --no-pipelined --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('classic=$c', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['legs'].items() if k.endswith('_ms')}, round(d['kernels']['median_mad']['ms_per_step'],3))
" >> gpurun_out/sel/ab.txt
done; done
cat gpurun_out/sel/ab.txt
