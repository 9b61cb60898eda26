#!/usr/bin/env python
"""Developer tool: the counter profile of the roofline kernel as the JSON bench.py reads
(profiles/rNN_pmc_coadd_fused.json), from one tools/gpu_round.sh run.

    python tools/make_pmc_json.py gpurun_out/<tag> profiles/r03_pmc_coadd_fused.json

Inputs of that directory: pmc_summary.txt (separate rocprofv3 --pmc passes, tools/pmc_summary.py),
kernel_stats.csv (rocprofv3 --kernel-trace, tools/rocpd_stats.py), bench.json.  The JSON carries the
hash of the kernel sources it was measured on (`kernel_sources_sha16`, the same function bench.py
uses): bench.py drops traffic / valu_frac / lds_frac from its line when the hash of the tree it runs
from differs, so a stale profile cannot be quoted for a changed kernel."""
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


def main(src, out):
    pm = {}
    for line in open(os.path.join(src, 'pmc_summary.txt')):
        m = re.match(r'(k_coadd_fused\S*)\s+(\S+)\s+mean\s+([0-9.]+)', line)
        if m:
            pm[m.group(2)] = float(m.group(3))
    avg_us = med_us = None
    with open(os.path.join(src, 'kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            if 'k_coadd_fused' in r['Name']:
                avg_us = float(r['AverageNs']) / 1e3
                med_us = float(r.get('MedianNs') or 0) / 1e3
                break
    bench = json.loads([l for l in open(os.path.join(src, 'bench.json')) if l.startswith('{')][-1])
    size, frames = bench['config']['size'], bench['config']['frames_per_gpu']
    npx = size * size
    fetch, write = pm['FETCH_SIZE'] * 1024 * 2, pm['WRITE_SIZE'] * 1024      # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    valu, lds = pm['SQ_INSTS_VALU'], pm['SQ_INSTS_LDS']
    waves_px = frames * npx / 64.0
    # Issue costs measured by tools/valu_rate.hip (profiles/r02_valu_rate.txt): v_pk_*_f32 2.12 ns, other VALU
    # 1.29 ns per wave-instruction and SIMD.  Packed share of the launch's VALU instructions: static count
    # of the ISA (pixel groups 218 of 464; staging and issue code carry none): 2 x 218 of ~1430 per item.
    packed = 0.30
    valu_s = valu * (packed * 2.12e-9 + (1 - packed) * 1.29e-9)
    # LDS pipe per 64 output pixels and frame: 13.5 + 2 ds_read_b64 (window rows shared by four pixels, table
    # tails), 8 ds_read_b128 (two tap-table nodes), 1 ds_read_u16; staging: 1.52 staged pixels per output
    # pixel in quads -> 0.76 ds_write_b128 pairs + 0.38 ds_write_b64
    lds_ns = 15.5 * 1.297 + 8 * 1.95 + 1 * 1.3 + 0.76 * 6.3 + 0.38 * 3.0
    d = {
        'source': 'tools/gpu_round.sh: rocprofv3 --kernel-trace --stats, then separate --pmc FETCH_SIZE / WRITE_SIZE / SQ_* passes '
                  'of the same bench command on one MI355X box, one commit',
        'kernel_sources_sha16': kernel_sources_sha16(),
        'units': 'FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, '
                 'MI355X_MICROARCH.md HBM section); WRITE_SIZE as is',
        'size': size, 'frames': frames, 'mask': True,
        'kernel': 'k_coadd_fused<LANCZOS3, mask coadd>',
        'avg_duration_us_kernel_trace': avg_us, 'median_duration_us_kernel_trace': med_us,
        'FETCH_SIZE_KiB': pm['FETCH_SIZE'], 'WRITE_SIZE_KiB': pm['WRITE_SIZE'],
        'hbm_bytes_per_launch': int(fetch + write),
        'algorithmic_bytes_per_launch': (frames * 12 + 12) * npx,
        'needed_read_bytes': frames * npx * (8 + 2),
        'read_over_needed': fetch / (frames * npx * (8 + 2)),
        'counters': pm,
        'valu_insts_per_px': valu / waves_px, 'lds_insts_per_px': lds / waves_px,
        'packed_share_of_valu': packed,
        'valu_simd_seconds_per_launch': valu_s,
        'lds_pipe': {'ns_per_64px': lds_ns, 'seconds_per_cu_per_launch': waves_px * lds_ns * 1e-9 / 256},
        'bench_ms_per_step': bench['ms_per_step'], 'bench_value_mpix_s': bench['value'],
    }
    json.dump(d, open(out, 'w'), indent=1)
    print(json.dumps({k: d[k] for k in ('kernel_sources_sha16', 'avg_duration_us_kernel_trace', 'hbm_bytes_per_launch',
                                        'read_over_needed', 'valu_insts_per_px', 'lds_insts_per_px')}))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
