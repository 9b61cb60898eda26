# developer: the fused kernel's yield mode - cost on the serial step, gain on the software-pipelined one
mkdir -p gpurun_out/$1
python -m pytest tests/test_fused_coadd_gpu.py -x -q -m gpu > gpurun_out/$1/t0.log 2>&1 || { tail gpurun_out/$1/t0.log; exit 1; }
ZM_FF_YIELD=2 python -m pytest tests/test_fused_coadd_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "fused or config1" > gpurun_out/$1/t1.log 2>&1 || { tail -20 gpurun_out/$1/t1.log; exit 1; }
tail -1 gpurun_out/$1/t1.log
for y in 0 3 2 0 3; do
  ZM_FF_YIELD=$y python bench.py --steps 24 --warmup 3 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --pipelined-depth 4 > gpurun_out/$1/y$y.json 2> gpurun_out/$1/y$y.err || { tail -8 gpurun_out/$1/y$y.err; exit 1; }
  python - <<P
import json
d=json.loads(open('gpurun_out/$1/y$y.json').read().strip().splitlines()[-1])
print('yield $y: serial', round(d['ms_per_step'],3), 'coadd', round(d['legs']['coadd_ms'],3), 'fused', round(d['kernels']['coadd_fused']['ms_per_step'],3), 'pipelined', round(d['pipelined']['ms_per_step'],3), d['pipelined']['status_ok'])
P
done
