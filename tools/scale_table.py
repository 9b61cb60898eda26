"""One table out of the bench lines of tools/scale_round.sh: per configuration the whole-job rate, the step time of
every rank, the wall clock of each exchange step (world.ranks[*].exchange_ms) and, beside them, what DESIGN.md
section 5 expects the exchange to cost on xGMI - so that the first run on an 8-GPU node says at a glance which step
is off.  usage: python3 tools/scale_table.py <dir with *.json>"""
import glob
import json
import os
import sys

LINK_GBS = 153.0          # one xGMI link, per direction (MI355X_MICROARCH.md); 7 links per GPU
PLANE = 3072 * 3072 * 4


def expected(name, n, frames):
    """The design's figures in ms, per exchange key."""
    if n < 2:
        return {}
    if name.startswith('weighted'):
        # all-reduce of the two fp32 partial-sum planes as one 75.5 MB buffer: direct reduce-scatter + all-gather moves
        # (n - 1) / n of it twice over n - 1 links; a ring is bound by ONE link
        buf = 2 * PLANE
        direct = 2 * buf * (n - 1) / n / ((n - 1) * LINK_GBS * 1e9) * 1e3
        ring = 2 * buf * (n - 1) / n / (LINK_GBS * 1e9) * 1e3
        mask = PLANE                                    # one int32 partial mask coadd per rank
        return {'all_reduce_planes': f'{direct:.2f} (direct) .. {ring:.2f} (ring)',
                'mask_band_exchange': f'{mask * (n - 1) / n / ((n - 1) * LINK_GBS * 1e9) * 1e3:.2f}',
                'mask_band_gather': f'{mask * (n - 1) / n / ((n - 1) * LINK_GBS * 1e9) * 1e3:.2f}',
                'mask_all_gather': f'(small worlds only) {mask * (n - 1) / ((n - 1) * LINK_GBS * 1e9) * 1e3:.2f}'}
    # exact CLIPPED: every rank sends (n - 1) / n of its resampled stack ({value, weight}: 8 B per pixel and frame)
    stack = frames * 2 * PLANE
    a2a = stack * (n - 1) / n / ((n - 1) * LINK_GBS * 1e9) * 1e3
    return {'stack_band_exchange': f'{a2a:.2f} (all links) .. {stack * (n - 1) / n / (LINK_GBS * 1e9) * 1e3:.2f} (one link)',
            'band_combine': f'~{0.86 * frames * n / 256:.2f} (k_combine_wide on {frames * n} samples x rows / {n})',
            'band_gather': f'{2 * PLANE * (n - 1) / n / ((n - 1) * LINK_GBS * 1e9) * 1e3:.2f}'}


def main():
    d = sys.argv[1]
    rows = []
    base = {}
    for path in sorted(glob.glob(os.path.join(d, '*.json'))):
        name = os.path.basename(path)[:-5]
        lines = [l for l in open(path) if l.startswith('{')]
        if not lines:
            rows.append((name, None))
            continue
        rows.append((name, json.loads(lines[-1])))
    print(f'{"run":22s} {"ranks":>5s} {"Mpix/s":>10s} {"ms/step":>8s} {"x 1 rank":>8s}  per-rank ms/step | exchange ms (max over ranks) | expected ms')
    for name, r in rows:
        if r is None:
            print(f'{name:22s}  no bench line')
            continue
        n = r['n_gpus']
        kind = name.rsplit('_', 1)[0].replace('_native', '')
        if n == 1:
            base[kind] = r['value']
        ranks = r['world']['ranks']
        per = ' '.join(f'{x["ms_per_step"]:.2f}' for x in ranks)
        ex = {}
        for x in ranks:
            for k, v in (x.get('exchange_ms') or {}).items():
                ex[k] = max(ex.get(k, 0.0), v)
        exs = ' '.join(f'{k}={v:.2f}' for k, v in sorted(ex.items())) or '-'
        exp = expected(name, n, r['config']['frames_per_gpu'])
        exps = ' '.join(f'{k}: {v}' for k, v in exp.items()) or '-'
        sp = f'{r["value"] / base[kind]:.2f}' if kind in base else '-'
        print(f'{name:22s} {n:5d} {r["value"]:10.0f} {r["ms_per_step"]:8.2f} {sp:>8s}  {per} | {exs} | {exps}')


if __name__ == '__main__':
    main()
