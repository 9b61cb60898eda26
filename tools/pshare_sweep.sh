# developer: the software-pipelined step against the number of subtractions in flight
mkdir -p gpurun_out/$1
for d in 1 2 3 4 6; do
  python bench.py --steps 24 --warmup 3 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --pipelined-depth $d > gpurun_out/$1/p$d.json 2> gpurun_out/$1/p$d.err || { tail -8 gpurun_out/$1/p$d.err; exit 1; }
  python - <<P
import json
d=json.loads(open('gpurun_out/$1/p$d.json').read().strip().splitlines()[-1])
print($d, round(d['ms_per_step'],3), {k: d['pipelined'][k] for k in ('ms_per_step','mpix_s','status_ok','subtractions_in_flight')})
P
done
