# developer A/B on one box: the step with every subtraction waiting for its summary (ZM_SUB_ASYNC=0) and without
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/async.txt
for rep in 1 2; do
for c in 0 1; do
  ZM_SUB_ASYNC=$c python bench.py --no-cpu-baseline --no-clocks --no-nightly --no-secondary --no-pipelined --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('async=$c', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['legs'].items() if k.endswith('_ms')})
" >> gpurun_out/ab/async.txt
done; done
cat gpurun_out/ab/async.txt
