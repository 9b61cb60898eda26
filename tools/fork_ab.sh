# developer A/B on one box (-DZM_DEV build made there): item headers of the fused kernel on the main stream (ZM_FF_FORK=0) /
# early on the second stream
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/fork.txt
ZM_HIPCC_FLAGS=-DZM_DEV python -c "
import importlib; b=importlib.import_module('zuds-pipeline_amd.build'); b.build(force=True, verbose=False)" > gpurun_out/ab/fork_build.log 2>&1 || exit 1
for rep in 1 2 3; do
for c in 0 1; do
  ZM_FF_FORK=$c python bench.py --no-cpu-baseline --no-clocks --no-nightly --no-secondary --no-pipelined --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fork=$c', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['legs'].items() if k.endswith('_ms')}, round(d['roofline']['avg_launch_us'],1))
" >> gpurun_out/ab/fork.txt
done; done
cat gpurun_out/ab/fork.txt
