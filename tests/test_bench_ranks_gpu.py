"""The N > 1 path of bench.py on the one card of the GPU box: `bench.py --gpus 2`
starts its two ranks itself (launch()), the ranks form a gloo group (ZM_DIST_BACKEND=gloo: RCCL
refuses two ranks on one device), shard the stack by rank, exchange (WEIGHTED: one all-reduce of the
partial sums; CLIPPED: row-band all-to-all + band all-gather) and rank 0 reports.  The coadd of the
two ranks must be the coadd ONE process makes of the same 2 x frames (`--emulate-ranks 2`): to 1e-6
for the sum-reduce (another summation order), bit for bit for the exact CLIPPED exchange."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--size', '1024', '--frames', '5', '--steps', '1', '--warmup', '0', '--no-secondary', '--no-nightly',
          '--no-clocks', '--no-cpu-baseline', '--no-pipelined', '--no-subtract']


def run_bench(extra, env_extra, dump):
    env = dict(os.environ, **env_extra)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + COMMON + extra + ['--dump-coadd', dump],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.parametrize('combine', ['WEIGHTED', 'CLIPPED'])
def test_two_gloo_ranks_on_one_card_equal_one_process(tmp_path, combine):
    two, c2 = run_bench(['--gpus', '2', '--combine', combine], {'ZM_DIST_BACKEND': 'gloo'}, str(tmp_path / 'two.npy'))
    w = two['world']
    assert w['world_size'] == 2 and w['backend'] == 'gloo' and w['launcher'] == 'bench.py'
    assert len(w['ranks']) == 2 and {r['rank'] for r in w['ranks']} == {0, 1}
    assert len({r['pid'] for r in w['ranks']}) == 2                      # two processes, really
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak' and two['value'] > 0
    one, c1 = run_bench(['--gpus', '1', '--emulate-ranks', '2', '--combine', combine], {}, str(tmp_path / 'one.npy'))
    assert one['world']['world_size'] == 1
    assert c1.shape == c2.shape == (2, 1024, 1024)
    assert (c1[1] > 0).mean() > 0.95
    if combine == 'CLIPPED':
        assert np.array_equal(c1, c2)
    else:
        assert np.array_equal(c1[1] > 0, c2[1] > 0)
        np.testing.assert_allclose(c2[0], c1[0], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(c2[1], c1[1], rtol=1e-6)


@pytest.mark.parametrize('combine', ['WEIGHTED', 'CLIPPED'])
def test_four_gloo_ranks_uneven_row_bands_equal_one_process(tmp_path, combine):
    """four ranks and a grid height that does not divide by four (1030 rows: bands of
    258 / 258 / 257 / 257) - the banded mask reduce (the default from four ranks on) and the row-band exchange of
    the exact CLIPPED stack with unequal bands, through bench.py's own launcher on the one card."""
    size = ['--size', '1030']
    four, c4 = run_bench(size + ['--gpus', '4', '--combine', combine, '--frames', '3'], {'ZM_DIST_BACKEND': 'gloo'},
                         str(tmp_path / 'four.npy'))
    w = four['world']
    assert w['world_size'] == 4 and w['backend'] == 'gloo' and len({r['pid'] for r in w['ranks']}) == 4
    assert four['n_gpus'] == 4 and four['value'] > 0
    one, c1 = run_bench(size + ['--gpus', '1', '--emulate-ranks', '4', '--combine', combine, '--frames', '3'], {},
                        str(tmp_path / 'one.npy'))
    assert c1.shape == c4.shape == (2, 1030, 1030)
    if combine == 'CLIPPED':
        assert np.array_equal(c1, c4)
    else:
        assert np.array_equal(c1[1] > 0, c4[1] > 0)
        np.testing.assert_allclose(c4[0], c1[0], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(c4[1], c1[1], rtol=1e-6)
