#!/bin/bash
# Developer tool: copy the files of a gpu_round.sh call (gpurun_out/<tag>/) into profiles/, regenerate the resource
# table, the README section and the DESIGN row of the final sources.   bash tools/install_profiles.sh r06   (the test count is read from the call's own tests_tail.log)
tag=${1:-r06}
o=gpurun_out/$tag
cp $o/bench.json profiles/${tag}_bench.json && cp $o/kernel_stats.csv profiles/${tag}_kernel_stats.csv && \
cp $o/pmc_summary.txt profiles/${tag}_pmc_summary.txt && cp $o/pmc_summary_clipped.txt profiles/${tag}_pmc_summary_clipped.txt && \
cp $o/pmc.json profiles/${tag}_pmc.json && cp $o/tests_tail.log profiles/${tag}_tests_tail.txt || exit 1
python3 tools/resource_usage.py profiles/${tag}_resource_usage.txt > /dev/null 2>&1
bash tools/update_profiles_readme.sh $tag
python3 - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
tail = open(f'profiles/{tag}_tests_tail.txt').read()
m = re.search(r'(\d+) passed', tail)
ntests = f'{m.group(1)} GPU tests green in the same call' if m and 'failed' not in tail else 'GPU tests NOT run in this call'
p = 'DESIGN.md'
s = open(p).read()
d = json.loads([l for l in open(f'profiles/{tag}_bench.json') if l.startswith('{')][-1])
k = {a: round(b['ms_per_step'], 2) for a, b in d['kernels'].items()}
n = d['nightly']
row = (f"| {tag}_b (round {int(tag[1:])}, HEAD, the final kernel sources; `profiles/{tag}_*` are this call's) | {d['value']:,.0f} | {d['ms_per_step']:.2f} | coadd leg {d['legs']['coadd_ms']:.2f} ms (fused kernel "
       f"{k['coadd_fused']} ms: {d['roofline']['valu_insts_per_px']:.1f} instructions per pixel and frame, `roofline.frac` {d['roofline']['frac']:.3f}; "
       f"mesh statistics {k['mesh_stats']}; box-OR {k['mask_box']}), subtraction leg {d['legs']['subtract_ms']:.2f} ms (hp_solve {k['hp_solve']} of "
       f"which `k_chol_df` 6 x {d['solve_roofline']['avg_us']:.0f} us; vectors {k['hp_vectors']}, Gram {k['hp_gram']}, apply {k['hp_apply']}, median / MAD "
       f"{k['median_mad']}, stamp search {k['hp_cells']}); `clipped` {d['clipped']['ms_per_step']:.2f} ms; pool "
       f"{n['pools']['1']['ms_per_subtraction']:.2f} ms per subtraction alone, batched best {n['batched_best']['ms_per_subtraction']:.2f} "
       f"({n['batched_best']['lanes_x_batch']}: {n['batched_best']['over_one_worker']:.2f} x one worker); `with_pcie_ms` "
       f"{d['clocks']['with_pcie_ms']:.1f}, `with_fits_ms` {d['clocks']['with_fits_ms']:.1f}; {ntests} |")
s2 = re.sub(rf'\| {tag}_b \(round.*?\n', row.replace('\\', '\\\\') + '\n', s, count=1)
open(p, 'w').write(s2)
PY
python3 -c "
import bench, json; print('hash ok' if bench.kernel_sources_sha16() == json.load(open('profiles/${tag}_pmc.json'))['kernel_sources_sha16'] else 'HASH MISMATCH')"
