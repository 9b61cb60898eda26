mkdir -p gpurun_out/r03af
python -m pytest tests -x -q -m gpu -k "fused or coadd or config1 or device_chain" > gpurun_out/r03af/tests.log 2>&1 || { tail -20 gpurun_out/r03af/tests.log; exit 1; }
tail -2 gpurun_out/r03af/tests.log
for f in 0 1 0 1; do
ZM_FF_FORK=$f python bench.py --steps 20 --warmup 3 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined > gpurun_out/r03af/f$f.json 2> gpurun_out/r03af/f$f.err || exit 1
python - <<P
import json
d=json.loads(open('gpurun_out/r03af/f$f.json').read().strip().splitlines()[-1])
print('fork $f', 'step', round(d['ms_per_step'],3), 'coadd', round(d['legs']['coadd_ms'],3), 'sub', round(d['legs']['subtract_ms'],3))
P
done
