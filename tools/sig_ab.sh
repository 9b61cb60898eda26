# developer A/B on one box (-DZM_DEV build made there): round flags by copy + event (ZM_HP_SIGNAL=0) and by the kernels' own signal
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/sig.txt
ZM_HIPCC_FLAGS=-DZM_DEV python -c "
import importlib; b=importlib.import_module('zuds-pipeline_amd.build'); b.build(force=True, verbose=False)" > gpurun_out/ab/sig_build.log 2>&1 || exit 1
for rep in 1 2; do
for c in 0 1; do
  ZM_HP_SIGNAL=$c python bench.py --no-cpu-baseline --no-clocks --no-nightly --no-secondary --no-pipelined --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('signal=$c', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['legs'].items() if k.endswith('_ms')})
" >> gpurun_out/ab/sig.txt
done; done
cat gpurun_out/ab/sig.txt
