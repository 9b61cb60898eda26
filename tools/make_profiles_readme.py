#!/usr/bin/env python3
"""Developer tool: the round's section of profiles/README.md FROM the files it describes (the
hand-written text named another source hash and test count than the JSON beside it).

    python3 tools/make_profiles_readme.py r05 > /tmp/section.md      # reads profiles/r05_*"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def main(tag):
    bench = json.loads([l for l in open(os.path.join(P, f'{tag}_bench.json')) if l.startswith('{')][-1])
    pmc = json.load(open(os.path.join(P, f'{tag}_pmc.json')))
    stats = {}
    with open(os.path.join(P, f'{tag}_kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            stats[re.sub(r'^void ', '', r['Name'])] = r
    tests = ''
    tl = os.path.join(P, f'{tag}_tests_tail.txt')
    if os.path.exists(tl):
        m = re.search(r'(\d+) passed', open(tl).read())
        tests = f'; the same call ran the {m.group(1)} `-m gpu` tests green' if m else ''

    def k(prefix, field='AverageNs'):
        for name, r in stats.items():
            if name.startswith(prefix):
                return name, float(r[field]) / 1e3
        return prefix, float('nan')
    r = bench['roofline']
    fk = (pmc.get('weighted') or {}).get('kernels', {}).get('k_coadd_fused') or {}
    ck = (pmc.get('clipped') or {}).get('kernels', {}).get('k_combine') or {}
    fn, fus = k('k_coadd_fused')
    cn, cus = k('k_chol_df', 'MedianNs')
    an, aus = k('k_hp_apply_w')
    bn, bus = k('k_chol_back_cols', 'MedianNs')
    c = bench.get('clipped') or {}
    oa = (bench.get('clocks') or {}).get('object_api_ms') or {}
    n = bench.get('nightly') or {}
    print(f'Round {int(tag[1:])}: the `{tag}_*` files come from ONE call of `tools/gpu_round.sh` on ONE box at ONE commit (the sources '
          f'hash to `kernel_sources_sha16` = `{pmc.get("kernel_sources_sha16")}` in `{tag}_pmc.json`{tests}).  This section is generated: '
          f'`python3 tools/make_profiles_readme.py {tag}`.\n')
    print('| File | What |\n|---|---|')
    print(f'| `{tag}_bench.json` | the one-line output of `python bench.py --gpus 1 --steps 20 --warmup 5`: **{bench["value"]:.0f} Mpix/s, '
          f'{bench["ms_per_step"]:.2f} ms per step**; `roofline` ({r["kernel"]}: {r["frac"]:.3f} of 8 TB/s on what it reads and writes, '
          f'{r.get("frac_of_achievable", float("nan")):.3f} of the 6.29 TB/s a copy reaches, traffic {r.get("wasted") or float("nan"):.2f} x the '
          f'algorithmic bytes, `valu_busy_frac` {r.get("valu_busy_frac")}; the leg {(r.get("leg") or {}).get("leg_frac", float("nan")):.3f}; '
          f'measured copy ceiling {(r.get("copy_ceiling") or {}).get("GBs_read_plus_write", float("nan")):.0f} GB/s), `solve_roofline` '
          f'({(bench.get("solve_roofline") or {}).get("avg_us", float("nan")):.0f} us per factorisation that runs), `clipped` (the reference\'s '
          f'default operator as a full step: {c.get("ms_per_step", float("nan")):.2f} ms; `k_combine` '
          f'{(c.get("combine_roofline") or {}).get("avg_launch_us", float("nan")):.0f} us = {(c.get("combine_roofline") or {}).get("frac", float("nan")):.2f} '
          f'of 8 TB/s, traffic {(c.get("combine_roofline") or {}).get("wasted") or float("nan"):.2f} x), `nightly` (batched best '
          f'{(n.get("batched_best") or {}).get("ms_per_subtraction", float("nan")):.2f} ms per subtraction = '
          f'{(n.get("batched_best") or {}).get("over_one_worker", float("nan")):.2f} x one worker), `clocks` (`with_pcie_ms` '
          f'{(bench.get("clocks") or {}).get("with_pcie_ms", float("nan")):.1f}, ratio to max(copy, device) '
          f'{((bench.get("clocks") or {}).get("pcie") or {}).get("ratio_to_max_of_copy_and_device", float("nan")):.3f}; `object_api_ms`: warm '
          f'{(oa.get("device") or {}).get("reference_from_images_ms", float("nan")):.0f} / {(oa.get("device") or {}).get("subtraction_from_images_ms", float("nan")):.0f} ms, '
          f'cold {(oa.get("cold") or {}).get("reference_from_images_ms", float("nan")):.0f} / {(oa.get("cold") or {}).get("subtraction_from_images_ms", float("nan")):.0f} ms), '
          f'`cpu_baseline` ({(bench.get("cpu_baseline") or {}).get("value", float("nan")):.1f} Mpix/s on {(bench.get("cpu_baseline") or {}).get("cores")} threads) |')
    print(f'| `{tag}_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats` of `bench.py --steps 20 --warmup 2 --no-clocks --no-cpu-baseline '
          f'--no-secondary --no-nightly --no-pipelined` (calls, total, average, min, max, **median**): `{fn[:40]}` {fus:.0f} us average, '
          f'`{cn[:16]}` median {cus:.0f} us, `{an[:18]}` {aus:.0f} us, `{bn[:18]}` median {bus:.0f} us |')
    print(f'| `{tag}_pmc_summary.txt`, `{tag}_pmc_summary_clipped.txt` | per-kernel means of the separate `--pmc` passes (`FETCH_SIZE`, '
          f'`WRITE_SIZE` in KiB; the `SQ_*` group) of the same command and of its `--combine CLIPPED` form |')
    print(f'| `{tag}_pmc.json` | the counters worked out per launch for the kernels the bench line quotes (`tools/make_pmc_json.py`; FETCH_SIZE '
          f'doubled as `MI355X_MICROARCH.md` prescribes): `{fk.get("name")}` {fk.get("hbm_bytes_per_launch", 0) / 1e9:.2f} GB of traffic per launch, '
          f'read {fk.get("read_over_needed", float("nan")):.2f} x the needed planes, {fk.get("valu_insts_per_px", float("nan")):.0f} vector '
          f'instructions per pixel and frame, `valu_busy_frac` {fk.get("valu_busy_frac", float("nan")):.2f}; `k_combine` '
          f'{ck.get("hbm_bytes_per_launch", 0) / 1e9:.2f} GB per launch |')
    print(f'| `{tag}_resource_usage.txt` | registers, spills, scratch and occupancy of every kernel as the compiler reports them '
          f'(`tools/resource_usage.py`, `hipcc -Rpass-analysis=kernel-resource-usage`) |')
    extra = [('fuzz.txt', 'the long differential fuzz runs at the final sources (`tools/gpu_fuzz_long.sh`): coadd forms against each other, subtraction forms against each other, random configurations against the oracle'),
             ('step_timeline.txt', 'one coadd leg kernel by kernel (start, length, gap to the previous end; `tools/step_timeline.py` on a kernel trace)'),
             ('subtract_timeline.txt', 'one subtraction leg kernel by kernel with queue and stream ids (`tools/sub_timeline.py`)'),
             ('pipelined_trace.txt', 'the pipelined step: what is in flight, what stretches, idle time (`tools/pipelined_trace.py`)')]
    for name, what in extra:
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', f'{tag}_{name}')):
            print(f'| `{tag}_{name}` | {what} |')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r05')
