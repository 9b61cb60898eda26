"""SWarp's own edge and mask conventions as options (``zm_ctx_set_conventions``,
``Engine.set_conventions``): EDGE_TRUNCATE - the interpolation kernel truncated at the frame edge
(zuds/astromatic/makecoadd/default.swarp:42-67) - and MASKRES_LANCZOS_ROUND - integer masks interpolated and rounded, as
SWarp does with mask.swarp (zuds/astromatic/makecoadd/mask.swarp:25, zuds/swarp.py:141-152) - against the oracle's
restatement of both (oracle/resample.py, ``edge=``, ``mask_resample=``).  The defaults stay the conventions of DESIGN.md
section 2 and stay bit-identical to what they were."""
import numpy as np
import pytest

from oracle import combine as ocombine
from oracle import resample as ores
from util import assert_close_masked, pkg, synth, to_oracle_wcs

pytestmark = pytest.mark.gpu


@pytest.fixture
def conv(engine):
    yield engine
    engine.set_conventions('zero', 'or')


def oracle(f, wout, kind=ores.LANCZOS3, edge='zero', mres='or', debug=None, with_wgt=True):
    onx, ony = wout.naxis
    px, py = ores.positions(to_oracle_wcs(wout), to_oracle_wcs(f['wcs']), onx, ony)
    o = ores.resample(f['img'], f['wgt'] if with_wgt else None, px, py, kind, f['flxscale'], f['mask'], edge=edge,
                      mask_resample=mres, debug=debug)
    return o + (ores.coverage(px, py, f['img'].shape[1], f['img'].shape[0], kind, edge),)


def frame(s, nx, ny, seed, dx, dy, rot=0.0, nbad=30):
    w = s.ztf_wcs(nx, ny, dx=dx, dy=dy, rot_deg=rot, tpv=True)
    f = s.make_frame(nx, ny, seed, w, nstars=25, nbad=nbad)
    rng = np.random.default_rng(seed)
    f['mask'][rng.integers(0, ny, 200), rng.integers(0, nx, 200)] |= rng.choice([1, 2, 256, 2048], 200).astype(f['mask'].dtype)
    f['mask'][40:52, 60:75] |= 64
    f['mask'][:3, :] |= 4                               # flagged rows at the very edge: the rim sees them
    return f


@pytest.mark.parametrize('kernel,kind', [('LANCZOS3', ores.LANCZOS3), ('BILINEAR', ores.BILINEAR)])
@pytest.mark.parametrize('shift', [(2.37, -1.61, 0.35), (0.0, 0.0, 0.0), (3.0, -2.0, 0.0), (-7.4, 5.2, -0.2)])
def test_edge_truncate_single_frame(conv, kernel, kind, shift):
    s = synth()
    base = s.ztf_wcs(210, 170, tpv=True)
    f = frame(s, 200, 180, 21, *shift)
    conv.set_conventions(edge='truncate')
    assert conv.query('edge') == 1 and conv.query('mask_resample') == 0
    o, w, m = conv.resample(f['img'], f['wcs'], base, wgt=f['wgt'], mask=f['mask'], kernel=kernel, fscale=f['flxscale'])
    ro, rw, rm, cov = oracle(f, base, kind, edge='truncate')
    ro0, rw0, rm0, cov0 = oracle(f, base, kind, edge='zero')
    fractional = any(abs(v - round(v)) > 1e-3 for v in shift[:2]) or shift[2] != 0.0
    assert cov.sum() > cov0.sum() if fractional else cov.sum() >= cov0.sum()      # the option covers more (the rim) ...
    # (a position within float rounding of the half-pixel line may fall on either side)
    assert ((w > 0) != (rw > 0)).sum() <= 4
    both = (w > 0) & (rw > 0)
    assert_close_masked(o[both], ro[both], 3e-5, 2e-3, f'{kernel} truncated values')
    assert_close_masked(w[both], rw[both], 5e-5, 0, f'{kernel} truncated weights')
    assert (m != rm).sum() <= 8, int((m != rm).sum())
    rim = cov & ~cov0
    if fractional:
        assert rim.sum() > (300 if kind == ores.LANCZOS3 else 100) and (w[rim] > 0).mean() > 0.5
    # the default is what it was, bit for bit, and the interior does not know about the option
    conv.set_conventions('zero', 'or')
    o0, w0, m0 = conv.resample(f['img'], f['wcs'], base, wgt=f['wgt'], mask=f['mask'], kernel=kernel, fscale=f['flxscale'])
    assert np.array_equal(w0 > 0, rw0 > 0) and np.array_equal(m0, rm0)
    inner = cov0 & (w0 > 0)
    # (a rim pixel of the option may sit where the default has a footprint touching the edge: compare well inside)
    ys, xs = np.nonzero(inner)
    core = np.zeros_like(inner)
    core[ys, xs] = True
    core[:8] = core[-8:] = False
    core[:, :8] = core[:, -8:] = False
    px, py = ores.positions(to_oracle_wcs(base), to_oracle_wcs(f['wcs']), 210, 170)
    deep = core & (px > 6) & (px < 200 - 7) & (py > 6) & (py < 180 - 7)
    assert deep.sum() > 10000 and np.array_equal(o[deep], o0[deep]) and np.array_equal(w[deep], w0[deep])


def test_mask_lanczos_round_single_frame_and_mask_only(conv):
    s = synth()
    base = s.ztf_wcs(210, 170, tpv=True)
    f = frame(s, 200, 180, 22, 1.3, 2.6, 0.15)
    for edge in ('zero', 'truncate'):
        conv.set_conventions(edge=edge, mask_resample='lanczos_round')
        o, w, m = conv.resample(f['img'], f['wcs'], base, wgt=f['wgt'], mask=f['mask'], fscale=f['flxscale'])
        dbg = {}
        ro, rw, rm, cov = oracle(f, base, edge=edge, mres='lanczos_round', debug=dbg)
        flt = dbg['mask_float']
        # the device evaluates fp32 table taps, the oracle exact ones: the integers agree except where the interpolated
        # value sits within their difference of a half
        assert np.abs(m - flt)[cov].max() <= 0.5 + 2e-3 * (1 + np.abs(flt[cov]).max() * 1e-3)
        assert (m != rm)[cov].mean() < 2e-3 and not m[~cov].any()
        assert (m < 0).any() and (rm < 0).any()          # ringing: what interpolating a bit mask does
        # the image does not depend on the mask rule
        conv.set_conventions(edge=edge)
        o2, w2, m2 = conv.resample(f['img'], f['wcs'], base, wgt=f['wgt'], mask=f['mask'], fscale=f['flxscale'])
        assert np.array_equal(o, o2) and np.array_equal(w, w2) and not np.array_equal(m, m2)
        # a mask on its own (run_align of a mask image, zuds/swarp.py:141-152)
        conv.set_conventions(edge=edge, mask_resample='lanczos_round')
        _, _, mm = conv.resample(None, f['wcs'], base, mask=f['mask'])
        assert np.array_equal(mm, m)
    # a constant mask stays that constant where the whole footprint is on the frame (unit-sum taps)
    const = dict(f, mask=np.full_like(f['mask'], 256))
    conv.set_conventions('zero', 'lanczos_round')
    _, _, mc = conv.resample(None, f['wcs'], base, mask=const['mask'])
    _, _, _, cov = oracle(const, base)
    assert (mc[cov] == 256).all() and not mc[~cov].any()


@pytest.mark.parametrize('combine', ['CLIPPED', 'WEIGHTED'])
def test_coadd_under_both_options(conv, combine):
    z, s = pkg(), synth()
    frames = [frame(s, 200, 180, 40 + i, 3.1 * i - 2.0, -2.3 * i + 1.0, 0.1 * i) for i in range(4)]
    wout = conv.autogrid([f['wcs'] for f in frames])
    onx, ony = wout.naxis
    p = z.coadd_params(combine=combine, mask_combine='AND', subtract_back=False, rescale_weights=False)
    default = conv.coadd(frames, wout, p)
    assert conv.query('fused_form') in (1, 2)
    for edge, mres in (('truncate', 'or'), ('zero', 'lanczos_round'), ('truncate', 'lanczos_round')):
        conv.set_conventions(edge, mres)
        img, wgt, msk, mw = conv.coadd(frames, wout, p)
        vals, wgts, masks, covs, flts = [], [], [], [], []
        for f in frames:
            dbg = {}
            o, w, m, cov = oracle(f, wout, edge=edge, mres=mres, debug=dbg)
            vals.append(o)
            wgts.append(w)
            masks.append(m)
            covs.append(cov)
            flts.append(dbg.get('mask_float'))
        ref, refw, _ = ocombine.combine(np.array(vals), np.array(wgts), combine)
        assert ((wgt > 0) != (refw > 0)).mean() < 2e-3
        both = (wgt > 0) & (refw > 0)
        assert_close_masked(img[both], ref[both], 1e-4, 2e-3, f'{combine} {edge} {mres}', max_bad_frac=2e-4)
        assert_close_masked(wgt[both], refw[both], 2e-4, 0, f'{combine} {edge} {mres} weights', max_bad_frac=2e-4)
        rmsk, rcov = ocombine.combine_masks(np.array(masks), np.array(covs), 'AND')
        # (the lattice-interpolated positions are good to ~1e-3 px: a pixel that close to the half-pixel line may be
        # covered on one side only)
        assert ((mw > 0) != (rcov > 0)).mean() < 2e-3
        if mres == 'or':
            assert (msk != rmsk).mean() < 2e-3
        else:
            assert (msk != rmsk).mean() < 5e-3           # (rounding at halves, then an AND over four frames)
        if edge == 'truncate':
            assert (wgt > 0).sum() > (default[1] > 0).sum()
    conv.set_conventions('zero', 'or')
    again = conv.coadd(frames, wout, p)
    for a, b in zip(default, again):
        assert np.array_equal(a, b)


def test_conventions_argument_checks(conv):
    z = pkg()
    with pytest.raises(ValueError):
        conv.set_conventions(edge='wrap')
    with pytest.raises(ValueError):
        conv.set_conventions(mask_resample='nearest')
    assert conv.L.zm_ctx_set_conventions(conv.ctx, 7, 0) != 0 and b'edge rule' in conv.L.zm_last_error()
    assert conv.L.zm_ctx_set_conventions(conv.ctx, 0, 3) != 0 and b'mask rule' in conv.L.zm_last_error()
    assert conv.query('edge') == 0 and conv.query('mask_resample') == 0
    with pytest.raises(z._lib.ZMError):
        conv.query('no such item')
