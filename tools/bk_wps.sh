# developer: the mesh statistics at 4 / 5 / 6 waves per SIMD (register budgets 128 / 96 / 80: -DBKF_WPS), one box
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/wps.txt
for w in 4 5 6; do
ZM_HIPCC_FLAGS="-DBKF_WPS=$w" python -c "
import importlib; b=importlib.import_module('zuds-pipeline_amd.build'); b.build(force=True, verbose=False)" > gpurun_out/ab/wps_build.log 2>&1 || exit 1
python bench.py --no-cpu-baseline --no-clocks --no-nightly --no-secondary --no-pipelined --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('wps=$w', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['legs'].items() if k.endswith('_ms')}, 'mesh_stats', round(d['kernels']['mesh_stats']['ms_per_step'],3))
" >> gpurun_out/ab/wps.txt
done
cat gpurun_out/ab/wps.txt
