export TMPDIR=/tmp; mkdir -p gpurun_out/ft
ZM_HIPCC_FLAGS=-DZM_DEV python -c "
import importlib; b=importlib.import_module('zuds-pipeline_amd.build'); b.build(force=True, verbose=False)" > gpurun_out/ft/build.log 2>&1 || exit 1
for c in 0 1; do
rm -rf gpurun_out/ft/prof$c
ZM_FF_FORK=$c rocprofv3 --kernel-trace -d gpurun_out/ft/prof$c -o t -- python3 bench.py --steps 10 --warmup 3 --no-clocks --no-cpu-baseline --no-secondary --no-nightly --no-pipelined > gpurun_out/ft/b$c.log 2>&1
python3 - <<PY > gpurun_out/ft/tl$c.txt
import sqlite3, re
con = sqlite3.connect('gpurun_out/ft/prof$c/t_results.db')
rows = con.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall()
names = [re.sub(r'^void ', '', n).split('(')[0][:36] for n, *_ in rows]
fs = [i for i, n in enumerate(names) if n.startswith('k_coadd_fused')]
# the last but two fused launches: full steps
for a in fs[-4:-2]:
    s0, e0 = rows[a][1], rows[a][2]
    print(f'fused {1e-3*(e0-s0):.1f} us on q{rows[a][3]}')
    for i, (n, s, e, q, st) in enumerate(rows):
        if i != a and e > s0 and s < e0:
            print(f'   overlaps: {names[i]:36s} q{q}  {1e-3*(max(s,s0)-s0):8.1f} .. {1e-3*(min(e,e0)-s0):8.1f}  (len {1e-3*(e-s):.1f})')
PY
rm -rf gpurun_out/ft/prof$c
done
cat gpurun_out/ft/tl0.txt; echo ======; cat gpurun_out/ft/tl1.txt
