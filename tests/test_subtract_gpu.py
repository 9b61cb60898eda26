"""Parity of the subtraction path (zm_subtract) with the hotpants oracle.

The fit runs in fp64 on both sides; the convolution runs in fp32 on the GPU, so
pixels agree to 1e-5 of the magnitudes that enter the difference
(|I| + |T (x) K|), which is the scale the 1e-4 target of BASELINE.md refers to.
"""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from oracle import hotpants as ohp
from util import assert_close_masked, pkg, synth

pytestmark = pytest.mark.gpu


def scene(nx=384, ny=352, seed=1, nstars=120, ksig=0.9, scale=1.3, bg=20.0,
          gradient=0.0, nbad=6):
    s = synth()
    rng = np.random.default_rng(seed)
    ref = np.full((ny, nx), 150.0)
    xs = rng.uniform(10, nx - 10, nstars)
    ys = rng.uniform(10, ny - 10, nstars)
    fl = np.exp(rng.uniform(np.log(3e3), np.log(8e4), nstars))
    s.add_stars(ref, xs, ys, fl, 2.0)
    yy, xx = np.mgrid[0:ny, 0:nx]
    if gradient:
        a = gaussian_filter(ref, ksig * (1 - gradient / 2), mode='nearest')
        b = gaussian_filter(ref, ksig * (1 + gradient / 2), mode='nearest')
        t = (xx / (nx - 1.0))
        sci = scale * ((1 - t) * a + t * b) + bg
    else:
        sci = scale * gaussian_filter(ref, ksig, mode='nearest') + bg
    ref = ref + rng.normal(0, 0.5, ref.shape)
    sci = sci + rng.normal(0, 3.0, sci.shape)
    bpm = np.zeros((ny, nx), np.uint8)
    for _ in range(nbad):
        bx, by = rng.integers(20, nx - 20), rng.integers(20, ny - 20)
        bpm[by:by + 3, bx:bx + 3] = 1
    return (sci.astype(np.float32), np.full((ny, nx), 3.0, np.float32),
            ref.astype(np.float32), np.full((ny, nx), 0.5, np.float32), bpm)


def compare(engine, data, tol=1e-5, got=None, **kw):
    sci, srms, ref, rrms, bpm = data
    d, n, info = got if got is not None else engine.subtract(sci, srms, ref, rrms, bpm, **kw)
    rd, rn, rinfo = ohp.subtract(sci, ref, srms, rrms, bpm, **kw)
    fi = kw.get('fi', 1e-30)
    gm, rm = d == np.float32(fi), rd == fi
    assert np.array_equal(gm, rm), f'masks differ on {(gm != rm).sum()} pixels'
    regs = [r for r in rinfo['regions'] if r is not None]
    assert info['nstamps_total'] == sum(r['nstamps_total'] for r in regs)
    assert info['nstamps_used'] == sum(r['nstamps_used'] for r in regs)
    assert info['niter'] == max(r['niter'] for r in regs)
    assert info['nmasked'] == rinfo['nmasked']
    ks = np.mean([r['kernel_sum'] for r in regs])
    assert abs(info['kernel_sum'] - ks) < 1e-6 * abs(ks)
    good = ~gm
    scale = np.abs(sci.astype(np.float64)) + np.abs(sci.astype(np.float64) - rd)
    err = np.abs(d.astype(np.float64) - rd)[good]
    lim = (tol * scale)[good] + 1e-4
    assert (err <= lim).all(), f'diff: worst excess {np.max(err - lim):.3e}'
    assert_close_masked(n[good], rn[good], 2e-5, 1e-5, 'noise')
    fin = kw.get('fin', np.sqrt(50000.0))
    assert np.all(n[gm] == np.float32(fin))
    return d, n, info, rd


COMMON = dict(tu=1e6, iu=1e6, tl=-1e3, il=-1e3)


def test_a_batch_of_subtractions_against_the_oracle(engine):
    """``zm_subtract_batch``: three frames of one configuration, their kernel fits as one chain of launches with
    the frame as a grid dimension - each product against the oracle like a lone subtraction's, and bit for bit
    the lone subtraction's."""
    kw = dict(r=5.0, rss=12.0, nsx=4, nsy=4, nrx=2, nry=1, ko=1, bgo=1, **COMMON)
    frames = [scene(seed=11), scene(seed=12, ksig=1.2, scale=0.9, bg=5.0), scene(seed=13, gradient=0.3)]
    got = engine.subtract_batch(frames, **kw)
    for data, g in zip(frames, got):
        compare(engine, data, got=g, **kw)
        d, n, info = engine.subtract(*data, **kw)
        assert np.array_equal(d, g[0]) and np.array_equal(n, g[1]) and info == g[2]


def test_constant_kernel_recovers_the_injected_convolution(engine):
    data = scene()
    d, n, info, rd = compare(engine, data, r=5.0, rss=12.0, nsx=4, nsy=4, ko=0, bgo=0, **COMMON)
    assert abs(info['kernel_sum'] - 1.3) < 2e-3
    good = d != np.float32(1e-30)
    assert abs(d[good].std() - 3.0) < 0.15          # residual = science noise
    assert abs(np.median(n[good]) - 3.0) < 0.1       # sqrt(3^2 + 0.5^2 sum K^2)
    assert info['status'] == 0 and info['ncoeff'] == 50


def test_three_by_three_regions(engine):
    # the region layout the reference emits: -nrx 3 -nry 3 (zuds/hotpants.py:83-84)
    data = scene(nx=540, ny=510, seed=3, nstars=400, gradient=0.3)
    compare(engine, data, r=4.0, rss=9.0, nsx=3, nsy=3, nrx=3, nry=3, ko=1, bgo=0, **COMMON)


def test_odd_frame_width(engine):
    # a width that is not a multiple of 4: the byte-wise dilation and scalar load paths
    data = scene(nx=381, ny=350, seed=8, nstars=150, gradient=0.2)
    compare(engine, data, r=5.0, rss=11.0, nsx=4, nsy=4, nrx=2, nry=1, ko=1, bgo=0, **COMMON)


def test_large_cells_many_cells_and_high_order(engine):
    """Configurations beyond the fast paths' limits: cells of more than 12288 pixels (stamp
    search from global memory), more than 256 cells per region (block-wide rejection) and more
    than 16 spatial terms (per-pair normal-matrix build)."""
    data = scene(nx=384, ny=352, seed=12, nstars=160)
    compare(engine, data, r=4.0, rss=9.0, nsx=2, nsy=2, ko=1, bgo=0, **COMMON)          # 192 x 176 px cells
    data = scene(nx=540, ny=510, seed=13, nstars=500)
    compare(engine, data, r=3.0, rss=6.0, nsx=17, nsy=16, ko=1, bgo=0, **COMMON)        # 272 cells
    data = scene(nx=448, ny=416, seed=14, nstars=400)
    d, n, info, rd = compare(engine, data, tol=2e-4, r=3.0, rss=7.0, nsx=8, nsy=8, ko=5, bgo=0, **COMMON)
    assert info['ncoeff'] == 1 + 48 * 21 + 1            # 21 spatial terms: 1010 unknowns (> 960: 1024-thread back substitution)


def test_reference_orders_ko4_bgo0(engine):
    # -ko 4 -bgo 0 are the orders the reference passes (zuds/hotpants.py:89-93)
    data = scene(nx=512, ny=480, seed=4, nstars=500, gradient=0.3)
    d, n, info, rd = compare(engine, data, r=4.0, rss=8.0, nsx=6, nsy=6, ko=4, bgo=0, **COMMON)
    assert info['ncoeff'] == 722


def test_spatially_varying_kernel_ko2_bgo1(engine):
    data = scene(nx=448, ny=416, seed=5, nstars=220, gradient=0.4)
    d, n, info, rd = compare(engine, data, r=6.0, rss=11.0, nsx=5, nsy=5, ko=2, bgo=1, **COMMON)
    assert info['ncoeff'] == 1 + 48 * 6 + 3


@pytest.mark.parametrize('hwk', [2, 3, 7, 10, 14])
def test_kernel_half_widths(engine, hwk):
    data = scene(nx=320, ny=300, seed=10 + hwk, nstars=90)
    compare(engine, data, r=hwk + 0.7, rss=2 * hwk + 1.2, nsx=3, nsy=3, ko=1, bgo=0, **COMMON)


@pytest.mark.parametrize('r, rss', [(16.3, 38.5), (20.9, 48.2), (6.5, 60.4), (18.2, 27.0)])
def test_half_widths_above_15_and_substamps_above_48(engine, r, rss):
    """zuds/hotpants.py:42-44 passes -r 2.5 SEEING -rss 6 SEEING unclamped: SEEING 6.5 px is (16, 39), 8 px is
    (20, 48).  There the x-filtered patch, term 0 and the template patch no longer fit the LDS together:
    k_hp_vectors_big builds the basis vectors in column chunks with term 0 in global memory (the last case is a wide
    kernel on a small substamp: the resident form at half width 18).  Against the oracle like every other case."""
    hw = int(r) + int(rss)
    data = scene(nx=2 * hw + 260, ny=2 * hw + 230, seed=int(r) + int(rss), nstars=160, ksig=2.2)
    d, n, info, rd = compare(engine, data, r=r, rss=rss, nsx=2, nsy=2, ko=1, bgo=0, **COMMON)
    assert info['nstamps_used'] >= 2


def test_normalise_to_template(engine):
    data = scene(seed=21)
    d, n, info, rd = compare(engine, data, r=5.0, rss=12.0, nsx=4, nsy=4, ko=0, bgo=0,
                             normalize=1, **COMMON)
    good = d != np.float32(1e-30)
    assert abs(d[good].std() - 3.0 / 1.3) < 0.15


def test_valid_ranges_and_bpm_propagate_to_the_fill_value(engine):
    sci, srms, ref, rrms, bpm = scene(seed=30)
    sci = sci.copy()
    sci[100, 120] = 9e5                 # above -iu
    ref = ref.copy()
    ref[200, 60] = np.nan
    kw = dict(r=5.0, rss=12.0, nsx=4, nsy=4, ko=0, bgo=0, tu=5e5, iu=5e5, tl=-1e3, il=-1e3)
    d, n, info, rd = compare(engine, (sci, srms, ref, rrms, bpm), **kw)
    assert np.all(d[95:106, 115:126] == np.float32(1e-30))     # grown by the kernel half width
    assert np.all(d[195:206, 55:66] == np.float32(1e-30))
    assert np.all(d[:5] == np.float32(1e-30)) and np.all(d[:, -5:] == np.float32(1e-30))


def test_no_usable_stamp_fills_everything(engine):
    sci, srms, ref, rrms, bpm = scene(nstars=0, seed=40)
    d, n, info = engine.subtract(sci, srms, ref, rrms, bpm, r=5.0, rss=12.0, nsx=3, nsy=3,
                                 ko=0, bgo=0, **COMMON)
    assert info['nstamps_total'] == 0 and info['status'] == 1
    assert np.all(d == np.float32(1e-30)) and np.all(n == np.float32(np.sqrt(50000.0)))


def test_parameter_validation(engine):
    z = pkg()
    sci, srms, ref, rrms, bpm = scene(nx=128, ny=128, nstars=10)
    with pytest.raises(z.ZMError):
        engine.subtract(sci, srms, ref, rrms, bpm, r=40.0)
    with pytest.raises(z.ZMError):
        engine.subtract(sci, srms, ref, rrms, bpm, r=5.0, rss=80.0)
    with pytest.raises(ValueError):
        engine.subtract(sci, srms[:-1], ref, rrms, bpm)
    with pytest.raises(ValueError):
        z.hp_params(nonsense=1)


def test_median_mad_matches_numpy(engine):
    rng = np.random.default_rng(0)
    for n in [1, 2, 7, 1000, 65537, 300000]:
        img = rng.normal(150, 5, n).astype(np.float32)
        img[rng.uniform(size=n) < 0.01] += 3000
        mask = (rng.uniform(size=n) < 0.3).astype(np.int32) * 8
        if (mask == 0).sum() == 0:
            mask[0] = 0
        med, mad = engine.median_mad(img, mask)
        pix = img[mask == 0]
        rmed = np.median(pix)
        rmad = 1.4826 * np.median(np.abs(pix - rmed))
        assert med == float(rmed), (n, med, rmed)
        assert abs(mad - rmad) <= 1e-6 * max(rmad, 1e-30), (n, mad, rmad)
    med, mad = engine.median_mad(np.array([[-3.0, 5.0], [1.0, -0.0]], np.float32))
    assert med == 0.5 and abs(mad - 1.4826 * 2.0) < 1e-6
    # the two middle ranks of an even count in different top-level bins (-1 | +2), NaNs
    # skipped, constant arrays, sign changes, no mask
    for arr in ([-1.0, -1.0, 2.0, 2.0], [np.nan, 4.0, np.nan, -2.0, 1e30, -1e30],
                [7.25] * 1001, [-5.0, -4.0, -3.0, 0.0, 1e-38, 3.0], list(np.linspace(-1, 1, 4096))):
        a = np.array(arr, np.float32)
        med, mad = engine.median_mad(a)
        pix = a[~np.isnan(a)]
        rmed = np.median(pix)
        assert med == float(rmed), (arr[:6], med, rmed)
        assert mad == pytest.approx(1.4826 * float(np.median(np.abs(pix - rmed))), rel=1e-7, abs=0)
    z = pkg()
    with pytest.raises(z.ZMError):
        engine.median_mad(np.ones(4, np.float32), np.ones(4, np.int32))


@pytest.mark.parametrize('ko', [2, 4])
def test_config2_parameters_against_the_oracle(engine, ko):
    """BASELINE config[2] at its stated parameters on a 1024 x 1024 frame: SEEING = 4 px ->
    r = 10 (21 x 21 kernel), rss = 24, 10 x 10 stamps, ko = 4 (the reference's default,
    zuds/hotpants.py:93: 722 unknowns) and ko = 2 (SURVEY.md 8(d) primary)."""
    data = scene(nx=1024, ny=1024, seed=50, nstars=1200, gradient=0.3, nbad=30)
    d, n, info, rd = compare(engine, data, r=10.0, rss=24.0, nsx=10, nsy=10, ko=ko, bgo=0, **COMMON)
    assert info['ncoeff'] == 1 + 48 * (ko + 1) * (ko + 2) // 2 + 1
    assert info['nstamps_total'] == 100 and info['status'] == 0
    assert abs(info['kernel_sum'] - 1.3) < 2e-3


@pytest.mark.parametrize('seed', [21, 33, 47])
def test_both_stamp_search_kernels_pick_the_same_substamps(engine, monkeypatch, seed):
    """``k_hp_cells_reg`` (cells of at most 12 288 px, pixels in registers) and ``k_hp_cells`` (larger cells,
    pixels in global memory) sum a cell's mean and sigma in different orders; the greedy substamp picks hang on a
    threshold made of them.  The same cells through both kernels (``ZM_CELLS_FORM=global``): the same centres - seen
    through everything that follows from them: stamp counts, rounds, kernel sum, every pixel of both products."""
    data = scene(nx=900, ny=840, seed=seed, nstars=900, gradient=0.2, nbad=40)
    kw = dict(r=6.0, rss=15.0, nsx=8, nsy=8, nrx=2, nry=2, ko=2, bgo=1, **COMMON)
    d0, n0, i0 = engine.subtract(*data, **kw)
    monkeypatch.setenv('ZM_CELLS_FORM', 'global')
    d1, n1, i1 = engine.subtract(*data, **kw)
    monkeypatch.delenv('ZM_CELLS_FORM')
    assert i0['status'] == 0 and i0['nstamps_total'] > 100
    for k in ('nstamps_total', 'nstamps_used', 'niter', 'ncoeff', 'kernel_sum', 'chi2', 'nmasked'):
        assert i0[k] == i1[k], k
    assert np.array_equal(d0, d1) and np.array_equal(n0, n1)


def test_barrier_timeout_is_retried_on_the_safe_path_and_reported(engine, monkeypatch, tmp_path):
    """a barrier of the fused factorisation that gives up (its workgroups were
    not all resident) used to look like a singular fit.  Now the time-outs are counted apart from bad
    pivots, the whole fit is repeated on the one-workgroup form (which waits for nobody) and the
    summary says so.  ZM_CHOL_SPIN_LIMIT=0 makes every barrier of the first attempt give up at its
    first unsuccessful poll; the retry must give the bits of an undisturbed run."""
    z = pkg()
    data = scene(nx=640, ny=600, seed=21, nstars=400)
    kw = dict(r=5.0, rss=12.0, nsx=5, nsy=5, nrx=2, nry=2, ko=2, bgo=0, **COMMON)
    d0, n0, i0 = engine.subtract(*data, **kw)
    assert i0['status'] == 0 and i0['retries'] == 0 and i0['nunsolved'] == 0
    monkeypatch.setenv('ZM_CHOL_SPIN_LIMIT', '0')
    d1, n1, i1 = engine.subtract(*data, **kw)
    monkeypatch.delenv('ZM_CHOL_SPIN_LIMIT')
    assert i1['retries'] == 1 and i1['status'] == 0 and i1['nunsolved'] == 0
    assert np.array_equal(d0, d1) and np.array_equal(n0, n1)
    for k in ('nstamps_total', 'nstamps_used', 'niter', 'ncoeff', 'kernel_sum', 'chi2', 'nmasked'):
        assert i0[k] == i1[k], k
    # the summary lands in the product header (hotpants -hki writes its kernel information there)
    cards = z.hotpants.info_cards(i1)
    assert cards['ZMSTATUS'] == 0 and cards['ZMRETRY'] == 1 and cards['ZMUNSOLV'] == 0


def test_unsolved_region_warns_and_is_flagged(engine):
    """A region whose stamps are all unusable: status bit ZM_HP_UNSOLVED, nunsolved counts it, its
    pixels carry the fill value, the object layer warns (hotpants.warn_unsolved)."""
    z = pkg()
    sci, srms, ref, rrms, bpm = scene(nx=640, ny=600, seed=22, nstars=400)
    bpm = bpm.copy()
    bpm[:300, :320] = 1                                   # region (0, 0) of a 2 x 2 layout: all bad
    d, n, info = engine.subtract(sci, srms, ref, rrms, bpm, r=5.0, rss=12.0, nsx=4, nsy=4, nrx=2, nry=2,
                                 ko=1, bgo=0, **COMMON)
    assert info['status'] == z._lib.HP_UNSOLVED and info['nunsolved'] == 1 and info['retries'] == 0
    assert np.all(d[:300, :320] == np.float32(1e-30))
    assert (d[300:, 320:] != np.float32(1e-30)).mean() > 0.9
    with pytest.warns(RuntimeWarning, match='1 region'):
        z.hotpants.warn_unsolved(info, 'test')


@pytest.mark.parametrize('case', [
    dict(nx=384, ny=352, seed=1, nstars=120, kw=dict(r=5.0, rss=12.0, nsx=4, nsy=4, ko=0, bgo=0)),               # 50 unknowns
    dict(nx=540, ny=510, seed=3, nstars=400, kw=dict(r=4.0, rss=9.0, nsx=3, nsy=3, nrx=3, nry=3, ko=1, bgo=0)),   # 146, 9 regions
    dict(nx=448, ny=416, seed=5, nstars=220, kw=dict(r=6.0, rss=11.0, nsx=5, nsy=5, ko=2, bgo=1)),               # 292
    dict(nx=1024, ny=1024, seed=50, nstars=1200, kw=dict(r=10.0, rss=24.0, nsx=10, nsy=10, nrx=3, nry=3, ko=4, bgo=0)),  # 722, 9 regions
    dict(nx=448, ny=416, seed=14, nstars=400, kw=dict(r=3.0, rss=7.0, nsx=8, nsy=8, ko=5, bgo=0)),               # 1010
])
def test_throughput_form_of_the_factorisation_gives_the_same_bits(engine, monkeypatch, case):
    """The kernel fit's factorisation has several forms: k_chol_df (the default where its tiles fit: LDS-resident
    tiles, hand-over flags, the Jacobi scaling folded into its tile load), k_chol_fused (many workgroups per
    region, region barriers; ZM_CHOL_FORM=lat) and k_chol_tp (one workgroup per region: cheap in CU-time; what a
    context uses when zm_ctx_set_share >= 2 and what a fit is repeated on after a time-out).  Same arithmetic,
    operation for operation: every product of the subtraction is bit-identical."""
    data = scene(nx=case['nx'], ny=case['ny'], seed=case['seed'], nstars=case['nstars'], gradient=0.3)
    kw = dict(case['kw'], **COMMON)
    monkeypatch.setenv('ZM_CHOL_FORM', 'lat')
    d0, n0, i0 = engine.subtract(*data, **kw)
    monkeypatch.setenv('ZM_CHOL_FORM', 'tp')
    d1, n1, i1 = engine.subtract(*data, **kw)
    # ... and the data-flow form (k_chol_df: resident tiles, flags instead of barriers; sizes whose tiles do not
    # fit three per workgroup run k_chol_fused under this switch)
    monkeypatch.setenv('ZM_CHOL_FORM', 'df')
    d3, n3, i3 = engine.subtract(*data, **kw)
    monkeypatch.delenv('ZM_CHOL_FORM')
    d4, n4, i4 = engine.subtract(*data, **kw)            # the default choice
    for d, n, i in ((d1, n1, i1), (d3, n3, i3), (d4, n4, i4)):
        assert i0['status'] == 0 and i['status'] == 0 and i0['retries'] == 0 and i['retries'] == 0
        assert np.array_equal(d0, d) and np.array_equal(n0, n)
        for k in ('nstamps_total', 'nstamps_used', 'niter', 'ncoeff', 'kernel_sum', 'chi2', 'nmasked'):
            assert i0[k] == i[k], k


@pytest.mark.parametrize('r', [5.0, 7.0, 10.0, 17.0, 20.0])
def test_the_two_forms_of_the_convolution_agree(engine, monkeypatch, r):
    """Round 4: the convolution of the template runs one wave per kernel block with the taps as scalar operands
    (k_hp_kbasis + k_hp_ktable + k_hp_apply_w) where a block fills most of a wave; ZM_APPLY_FORM=tile is the
    workgroup-per-six-blocks kernel of rounds 1 - 3.  The block kernels are the same sums in another order (the
    spatial terms outside), so the fp32 taps differ in the last bit here and there: the products agree to a few
    1e-7 of the template, the fill pattern and the fit summary exactly."""
    data = scene(nx=700, ny=660, seed=31, nstars=500, gradient=0.3)
    kw = dict(r=r, rss=2.4 * r, nsx=5 if r <= 10 else 2, nsy=5 if r <= 10 else 2, nrx=2, nry=2, ko=2, bgo=0, **COMMON)
    monkeypatch.setenv('ZM_APPLY_FORM', 'wave')          # (the default only where a block fills most of a wave)
    d0, n0, i0 = engine.subtract(*data, **kw)
    monkeypatch.setenv('ZM_APPLY_FORM', 'tile')
    d1, n1, i1 = engine.subtract(*data, **kw)
    monkeypatch.delenv('ZM_APPLY_FORM')
    assert i0 == i1
    fill = np.float32(1e-30)
    assert np.array_equal(d0 == fill, d1 == fill)
    scale = float(np.abs(data[2]).max())
    assert np.abs(d0 - d1).max() <= 2e-6 * scale
    assert np.abs(n0 - n1).max() <= 2e-6 * float(np.abs(n1[np.isfinite(n1)]).max())


def test_two_engines_side_by_side_without_a_pool(engine):
    """two contexts that subtract at the same time without having declared it
    (zm_ctx_set_share) used to size the many-workgroup factorisation to the whole GPU each.  A context
    that finds another one fitting on its device now takes the one-workgroup-per-region form by itself;
    whatever the interleaving (a collision of the first launches ends in a counted retry), the products
    are those of a lone run."""
    from concurrent.futures import ThreadPoolExecutor
    z = pkg()
    data = scene(nx=640, ny=600, seed=23, nstars=400, gradient=0.2)
    kw = dict(r=5.0, rss=12.0, nsx=5, nsy=5, nrx=2, nry=2, ko=2, bgo=0, **COMMON)
    d0, n0, i0 = engine.subtract(*data, **kw)
    assert i0['status'] == 0 and i0['retries'] == 0
    engines = [z.Engine(0), z.Engine(0)]

    def work(e):
        return [e.subtract(*data, **kw) for _ in range(6)]

    try:
        with ThreadPoolExecutor(2) as ex:
            res = list(ex.map(work, engines))
    finally:
        for e in engines:
            e.close()
    for runs in res:
        for d, n, info in runs:
            assert info['status'] == 0 and info['nunsolved'] == 0
            assert np.array_equal(d, d0) and np.array_equal(n, n0)
            for k in ('nstamps_total', 'nstamps_used', 'niter', 'ncoeff', 'kernel_sum', 'chi2'):
                assert info[k] == i0[k], k
