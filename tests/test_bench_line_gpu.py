"""The one JSON line of bench.py (the driver's contract): every key the contract names, the roofline and
cpu_baseline objects with their fields, the bookkeeping consistent (value = Mpix of the steps / time).
Run at a reduced size so that it takes seconds - the numbers are not looked at, the shape of the line is."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--size', '2048', '--frames', '4', '--steps', '2', '--warmup', '1',
           '--no-clocks', '--no-nightly', '--no-pipelined', '--cpu-frames', '1']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'Mpix/s' and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    mpix = (4 + 1) * 2048 * 2048 / 1e6
    assert d['value'] == pytest.approx(mpix / (d['ms_per_step'] * 1e-3), rel=1e-6)
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'copy_ceiling', 'frac_of_copy_ceiling'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'])
    # counters are quoted only for the workload and the sources they were taken on: not at this size
    assert r['traffic'] is None
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0
    assert d['apply_roofline']['valu_frac'] > 0
    # leg-level and solver rooflines, and the reference's default operator (COMBINE_TYPE CLIPPED) beside the headline
    leg = r['leg']
    assert leg['leg_frac'] == pytest.approx(leg['algorithmic_bytes'] / (leg['ms'] * 1e-3) / 1e9 / 8000.0) and leg['leg_traffic'] is None
    assert r['algorithmic_bytes_per_launch'] == (4 * (8 + 2) + 12) * 2048 * 2048          # int16 masks: 2-byte box-OR entries
    sr = d['solve_roofline']
    assert sr['bound'] == 'mfma' and sr['unit'] == 'TFLOP/s' and sr['frac'] == pytest.approx(sr['achieved'] / sr['peak']) and sr['frac'] > 0
    c = d['clipped']
    assert c['combine'] == 'CLIPPED' and c['ms_per_step'] > 0
    for k in ('stack_roofline', 'combine_roofline'):
        assert c[k]['bound'] == 'hbm' and c[k]['frac'] == pytest.approx(c[k]['achieved'] / 8000.0) and c[k]['traffic'] is None
    assert c['value_mpix_s'] == pytest.approx(mpix / (c['ms_per_step'] * 1e-3), rel=1e-6)


def test_bench_line_nightly_leg_with_batched_pools():
    """The many-subtractions leg of the line: `nightly.pools` (J separate chains) and `nightly.batched`
    (lanes x batch: the kernel fits of a batch as one chain of launches), no failed job, `batched_best` against one
    worker.  Reduced size; the numbers are not looked at."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--size', '2048', '--frames', '4', '--steps', '2', '--warmup', '1',
           '--no-clocks', '--no-pipelined', '--no-secondary', '--no-cpu-baseline', '--nightly-jobs', '4',
           '--nightly-pools', '1,2', '--nightly-batches', '1x2,2x2,1x4']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    n = d['nightly']
    assert n['jobs'] == 4 and set(n['pools']) == {'1', '2'} and set(n['batched']) == {'1x2', '2x2', '1x4'}
    # (`failed` counts jobs with a non-zero status: at this reduced size a region of some frame may have no stamp;
    # what matters here: no job raised, and every pool sees the same jobs fail - the products do not depend on it)
    for rec in list(n['pools'].values()) + list(n['batched'].values()):
        assert 'first_error' not in rec and rec['ms_per_subtraction'] > 0 and len(rec['passes_ms']) == 2
        assert rec['failed'] == n['pools']['1']['failed']
    bb = n['batched_best']
    assert bb['lanes_x_batch'] in n['batched']
    assert bb['over_one_worker'] == pytest.approx(n['pools']['1']['ms_per_subtraction'] / bb['ms_per_subtraction'])
    assert n['subtract_mpix_s'] == pytest.approx(max(r['subtract_mpix_s'] for r in list(n['pools'].values()) + list(n['batched'].values())))


def test_h2d_copies_overlap_the_step():
    """`clocks.with_pcie_ms`: the H2D copies of step k + 1 run under the kernels of step k, so a step with the data
    movement inside costs max(copy, device), not their sum (the ratio had slipped from 1.01 to
    1.22 unnoticed).  Sixteen full-size frames: the copy (1.6 GB, ~30 ms) is several times the device time, and the
    D2H of the products (0.23 GB on the same link: PCIe is full duplex, but the two directions are not free of each
    other - about 2 ms per step) stays below the tolerance."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--frames', '16', '--steps', '2', '--warmup', '1',
           '--no-secondary', '--no-pipelined', '--no-nightly', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900,
                         env=dict(os.environ, ZM_BENCH_PCIE_ONLY='1'))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    c = d['clocks']
    assert c.get('with_pcie_ms'), c
    p = c['pcie']
    assert p['h2d_copy_alone_ms'] > 2 * c['device_ms']            # (the case the overlap is for)
    assert p['ratio_to_max_of_copy_and_device'] <= 1.10, c
