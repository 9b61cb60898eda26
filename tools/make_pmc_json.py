#!/usr/bin/env python
"""Developer tool: the counter profile bench.py quotes (profiles/rNN_pmc.json), from one tools/gpu_round.sh run.

    python tools/make_pmc_json.py gpurun_out/<tag> profiles/r05_pmc.json

Inputs of that directory: pmc_summary.txt (+ pmc_summary_clipped.txt), written by tools/pmc_summary.py from
separate rocprofv3 --pmc passes of the short bench command (FETCH_SIZE, WRITE_SIZE and the SQ counters each in
a pass of their own); kernel_stats.csv (rocprofv3 --kernel-trace --stats, tools/rocpd_stats.py); bench.json.
Per bench command ('weighted' = the headline, 'clipped' = --combine CLIPPED) and kernel: the launch on the
largest grid (the 32-frame stack, not the science frame of the subtraction) with HBM bytes per launch =
2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section;
WRITE_SIZE as is) and the SQ instruction counters.  The JSON carries the hash of the kernel sources it was
measured on (`kernel_sources_sha16`, the same function bench.py uses): bench.py drops the counter figures from
its line when the hash of the tree it runs from differs."""
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, 'zuds-pipeline_amd', 'csrc')
    files = sorted(f for f in os.listdir(d) if f.endswith(('.hip', '.h')))
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    h.update(open(os.path.join(root, 'include', 'zudsmi.h'), 'rb').read())
    return h.hexdigest()[:16]


WANT = ('k_coadd_fused', 'k_mesh_stats', 'k_mesh_guess', 'k_mask_box', 'k_combine', 'k_chol_fused', 'k_chol_df', 'k_hp_apply', 'k_hp_ktable',
        'k_bk_rows', 'k_bk_cols', 'k_ff_headers')


def section(path, frames, npx):
    if not os.path.exists(path):
        return None
    rows = {}
    for line in open(path):
        m = re.match(r'(\S+)\s+grid\s+(\d+)\s+(\S+)\s+mean\s+([0-9.]+)\s+launches\s+(\d+)', line)
        if m and m.group(1).startswith(WANT):
            rows.setdefault(m.group(1), {}).setdefault(int(m.group(2)), {})[m.group(3)] = (float(m.group(4)), int(m.group(5)))
    kernels = {}
    for kn, grids in rows.items():
        gs = max(grids)                                   # the stack-sized launch
        c = {k: v[0] for k, v in grids[gs].items()}
        key = re.sub(r'<.*', '', kn)
        if key in ('k_coadd_fused_dma', 'k_coadd_fused_own'):
            key = 'k_coadd_fused'                        # whichever form the launcher picked (the entry keeps the name)
        if key in kernels and kernels[key]['grid'] >= gs:
            continue
        e = {'name': kn, 'grid': gs, 'launches_seen': max(v[1] for v in grids[gs].values()), 'counters': c}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            e['hbm_bytes_per_launch'] = int(c['FETCH_SIZE'] * 1024 * 2 + c['WRITE_SIZE'] * 1024)
            e['fetch_bytes_per_launch'] = int(c['FETCH_SIZE'] * 1024 * 2)
        if key == 'k_coadd_fused' and 'SQ_INSTS_VALU' in c:
            waves_px = frames * npx / 64.0
            e['valu_insts_per_px'] = c['SQ_INSTS_VALU'] / waves_px
            e['lds_insts_per_px'] = c.get('SQ_INSTS_LDS', 0.0) / waves_px
            if c.get('SQ_WAVE_CYCLES'):
                # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES both count quad-cycles, the latter summed over the waves of
                # a SIMD: at four waves per SIMD a SIMD that always issues VALU shows 1 / 4
                e['valu_active_over_wave_cycles'] = c.get('SQ_ACTIVE_INST_VALU', 0.0) / c['SQ_WAVE_CYCLES']
                # the share of a SIMD's time with a vector instruction in its pipe: x the resident waves per SIMD
                # (two 512-thread workgroups per CU = four waves per SIMD)
                e['waves_per_simd'] = 4
                e['valu_busy_frac'] = e['valu_active_over_wave_cycles'] * e['waves_per_simd']
        kernels[key] = e
    return {'kernels': kernels}


def main(src, out):
    bench = json.loads([l for l in open(os.path.join(src, 'bench.json')) if l.startswith('{')][-1])
    size, frames = bench['config']['size'], bench['config']['frames_per_gpu']
    npx = size * size
    d = {
        'source': 'tools/gpu_round.sh: rocprofv3 --kernel-trace --stats, then separate --pmc FETCH_SIZE / WRITE_SIZE / SQ_* passes '
                  'of the same bench command (and of its --combine CLIPPED form) on one MI355X box, one commit',
        'kernel_sources_sha16': kernel_sources_sha16(),
        'units': 'hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 tallies 128-B read requests at 64 B, '
                 'MI355X_MICROARCH.md HBM section); per kernel the launch on its largest grid (the stack, not the science frame)',
        'size': size, 'frames': frames, 'mask_dtype': bench['config'].get('mask_dtype'),
        'bench_ms_per_step': bench['ms_per_step'], 'bench_value_mpix_s': bench['value'],
    }
    for name, fn in (('weighted', 'pmc_summary.txt'), ('clipped', 'pmc_summary_clipped.txt')):
        sec = section(os.path.join(src, fn), frames, npx)
        if sec:
            d[name] = sec
    ks = os.path.join(src, 'kernel_stats.csv')
    if os.path.exists(ks):
        with open(ks) as f:
            for r in csv.DictReader(f):
                for key, e in (d.get('weighted') or {}).get('kernels', {}).items():
                    if r['Name'].startswith(('void ' + key, key)) and 'avg_duration_us_kernel_trace' not in e and \
                            (key != 'k_coadd_fused' or re.sub(r'^void ', '', r['Name']).startswith(re.sub(r'<.*', '', e['name']))):
                        e['avg_duration_us_kernel_trace'] = float(r['AverageNs']) / 1e3
    fk = (d.get('weighted') or {}).get('kernels', {}).get('k_coadd_fused')
    if fk and 'fetch_bytes_per_launch' in fk:
        fk['needed_read_bytes'] = frames * npx * (8 + (2 if d['mask_dtype'] else 0))
        fk['read_over_needed'] = fk['fetch_bytes_per_launch'] / fk['needed_read_bytes']
    json.dump(d, open(out, 'w'), indent=1)
    print(json.dumps({'kernel_sources_sha16': d['kernel_sources_sha16'],
                      'fused': {k: fk.get(k) for k in ('avg_duration_us_kernel_trace', 'hbm_bytes_per_launch', 'read_over_needed',
                                                        'valu_insts_per_px', 'lds_insts_per_px')} if fk else None}))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
