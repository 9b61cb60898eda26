"""The fused resample -> WEIGHTED / AVERAGE coadd kernel (frames looped inside the output
tile, running sums in registers: k_coadd_fused) against the materialised path it replaces
(resampled stack in HBM + k_combine_sum, selected with ZM_COADD_FUSED=0): bit-identical
coadd, weight, mask coadd and coverage - the sums run in the same order with the same
operations - for full coadds and for the partial sums of a multi-GPU stack, on interior
tiles (fast items), edge tiles, aligned grids (delta kernels), rotated / rescaled frames
whose footprint exceeds the LDS tile, masks with bits above 15, frames without masks and
frames that miss the grid altogether.  The materialised path itself is pinned to the oracle
in test_coadd_gpu.py / test_resample_gpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

from util import pkg, synth

pytestmark = pytest.mark.gpu


def run_both(engine, frames, wout, p, want_mask=True):
    out = {}
    for mode in ('0', '1'):
        os.environ['ZM_COADD_FUSED'] = mode
        try:
            out[mode] = engine.coadd(frames, wout, p, want_mask=want_mask)
        finally:
            os.environ.pop('ZM_COADD_FUSED', None)
    return out['0'], out['1']


def assert_same(a, b):
    for x, y, name in zip(a, b, ('img', 'wgt', 'mask', 'mask coverage')):
        assert (x is None) == (y is None), name
        if x is not None:
            assert np.array_equal(x, y, equal_nan=True), \
                f'{name}: {(x != y).sum()} of {x.size} pixels differ'


def stack(n, nx, ny, seed, dither=8.0, rot=0.1, tpv=True, nbad=200, scale_jitter=0.0):
    s = synth()
    base = s.ztf_wcs(nx, ny, tpv=tpv)
    rng = np.random.default_rng(seed)
    xs, ys = rng.uniform(5, nx - 5, 60), rng.uniform(5, ny - 5, 60)
    fl = np.exp(rng.uniform(np.log(2e3), np.log(5e4), 60))
    ra, dec = base.all_pix2world(xs, ys, 0)
    frames = []
    for i in range(n):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-dither, dither), dy=rng.uniform(-dither, dither),
                      rot_deg=rng.uniform(-rot, rot), tpv=tpv)
        if scale_jitter:
            w.cd = np.asarray(w.cd) * (1.0 + rng.uniform(-scale_jitter, scale_jitter))
        frames.append(s.make_frame(nx, ny, seed + i, w, star_sky=(ra, dec, fl), sky=rng.uniform(100, 300),
                                   magzp=rng.uniform(25.5, 26.5), nbad=nbad))
    return frames, base


@pytest.mark.parametrize('kind', ['WEIGHTED', 'AVERAGE'])
@pytest.mark.parametrize('mask_kind', ['AND', 'OR'])
def test_fused_equals_materialised_with_background_and_rescale(engine, kind, mask_kind):
    z = pkg()
    frames, wout = stack(6, 700, 650, 100)
    p = z.coadd_params(combine=kind, mask_combine=mask_kind, subtract_back=True, rescale_weights=True,
                       back_size=128)
    a, b = run_both(engine, frames, wout, p)
    assert_same(a, b)
    assert (b[1] > 0).mean() > 0.9 and (b[3] == 0).any()
    if mask_kind == 'OR':
        assert (b[2] != 0).any()


def test_union_grid_edge_tiles_and_frames_that_miss_tiles(engine):
    z = pkg()
    frames, _ = stack(5, 520, 480, 200, dither=60.0, rot=0.3)
    wout = engine.autogrid([f['wcs'] for f in frames])
    assert wout.naxis[0] > 560
    p = z.coadd_params(combine='WEIGHTED', subtract_back=False, rescale_weights=False)
    a, b = run_both(engine, frames, wout, p)
    assert_same(a, b)
    # a frame entirely off the grid contributes nothing and breaks nothing
    s = synth()
    far = dict(frames[0], wcs=s.ztf_wcs(520, 480, dx=5000.0, dy=-4000.0))
    a, b = run_both(engine, frames + [far], wout, p)
    assert_same(a, b)


def test_aligned_grids_take_the_delta_kernel_path(engine):
    z = pkg()
    s = synth()
    frames, base = stack(3, 400, 380, 300, dither=0.0, rot=0.0)
    frames[1]['wcs'] = s.ztf_wcs(400, 380, dx=-7.0, dy=3.0)           # integer shift
    frames[2]['wcs'] = s.ztf_wcs(400, 380, dx=2.5, dy=0.0)            # delta in y only
    p = z.coadd_params(combine='WEIGHTED', subtract_back=False, rescale_weights=False)
    a, b = run_both(engine, frames, base, p)
    assert_same(a, b)
    inner = (slice(12, -12), slice(12, -12))
    assert (b[1][inner] > 0).mean() > 0.97


def test_large_footprints_fall_back_to_the_global_gather(engine):
    z = pkg()
    frames, base = stack(3, 420, 400, 400, rot=0.2)
    for f, k in zip(frames, (1.0, 2.6, 0.45)):                  # finer / coarser output sampling
        f['wcs'].cd = np.asarray(f['wcs'].cd) * k
    frames[1]['wcs'] = synth().ztf_wcs(420, 400, rot_deg=33.0)
    p = z.coadd_params(combine='WEIGHTED', subtract_back=False, rescale_weights=False)
    a, b = run_both(engine, frames, base, p)
    assert_same(a, b)


def test_high_mask_bits_missing_masks_and_no_masks(engine):
    z = pkg()
    frames, base = stack(4, 450, 430, 500)
    frames[0]['mask'][100:140, 200:260] |= 1 << 16
    frames[1]['mask'][300:310, 50:90] |= (1 << 17) | 0xffff
    frames[2]['mask'][20:23, 400:403] = 0xffff
    p = z.coadd_params(combine='WEIGHTED', mask_combine='OR', subtract_back=False, rescale_weights=False)
    a, b = run_both(engine, frames, base, p)
    assert_same(a, b)
    assert (b[2] >> 16).any()
    some = [dict(f) for f in frames]
    some[1]['mask'] = None                                       # this frame does not enter the mask coadd
    a, b = run_both(engine, some, base, p)
    assert_same(a, b)
    none = [dict(f, mask=None) for f in frames]
    a, b = run_both(engine, none, base, p, want_mask=False)
    assert_same(a, b)


def test_partial_sums_and_single_frame(engine):
    """zm_coadd_dev(partial = 1): S1, S0 and the mask with its -1 markers, as the multi-GPU
    reduce consumes them; and a stack of one."""
    import torch
    z = pkg()
    dmod = __import__('importlib').import_module('zuds-pipeline_amd.device')
    frames, base = stack(4, 500, 470, 600)
    p = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True)
    res = {}
    for mode in ('0', '1'):
        os.environ['ZM_COADD_FUSED'] = mode
        try:
            dc = dmod.DeviceCoadd(base, p, device=0, engine=engine, want_mask=True)
            dfr = dmod.DeviceFrames(frames, dc.device)
            dc.run(dfr, partial=True)
            dc.stream.synchronize()
            res[mode] = [t.cpu().numpy() for t in (dc.img, dc.wgt, dc.mask)]
        finally:
            os.environ.pop('ZM_COADD_FUSED', None)
            engine.set_stream(0)
    for x, y in zip(res['0'], res['1']):
        assert np.array_equal(x, y)
    assert (res['1'][2] == -1).any() and (res['1'][2] != -1).any()
    a, b = run_both(engine, frames[:1], base, z.coadd_params(combine='AVERAGE', subtract_back=False,
                                                             rescale_weights=False))
    assert_same(a, b)


@pytest.mark.parametrize('n', [1, 2, 3, 5])
def test_tile_queue_with_short_stacks_and_many_tiles(engine, n):
    """The persistent kernel takes tiles from a queue two items ahead: with one or two frames per
    tile that is one or two tiles ahead (ring of tile ordinals), on a grid with more tiles than
    resident workgroups (1600 x 1600: 1250 tiles against 768)."""
    z = pkg()
    frames, base = stack(n, 1600, 1600, 700 + n, nbad=2000)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='OR', subtract_back=False, rescale_weights=False)
    a, b = run_both(engine, frames, base, p)
    assert_same(a, b)
    assert (b[1] > 0).mean() > 0.95


def test_fullsize_stack_fused_equals_materialised(engine):
    """BASELINE config[1] geometry at full size (3072 x 3072, TPV, +-15 px, +-0.1 deg), 4 frames."""
    z = pkg()
    s = synth()
    nx = ny = 3072
    rng = np.random.default_rng(7)
    base = s.ztf_wcs(nx, ny, tpv=True)
    frames = []
    for i in range(4):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-15, 15), dy=rng.uniform(-15, 15), rot_deg=rng.uniform(-0.1, 0.1))
        img = rng.normal(200.0, 6.0, (ny, nx)).astype(np.float32)
        mask = np.zeros((ny, nx), np.int32)
        bad = rng.integers(0, nx * ny, 9000)
        mask.ravel()[bad] = rng.choice([1, 256, 2, 2048], bad.size)
        wgt = np.where((mask & 198589) > 0, 0.0, 1.0 / 36.0).astype(np.float32)
        frames.append(dict(img=img, wgt=wgt, mask=mask, wcs=w, flxscale=0.3 + 0.05 * i))
    p = z.coadd_params(combine='WEIGHTED', subtract_back=True, rescale_weights=True)
    a, b = run_both(engine, frames, base, p)
    assert_same(a, b)
    assert (b[1] > 0).mean() > 0.98


@pytest.mark.parametrize('kind', ['CLIPPED', 'MEDIAN'])
def test_stack_mode_feeds_the_clipped_and_median_coadds(engine, kind):
    """CLIPPED / MEDIAN stacks are resampled by the same kernel in STACK mode (samples stored instead of
    summed, the mask coadd still folded in registers): the coadd, its weight, the mask coadd and the
    coverage equal those of the k_resample path (ZM_COADD_FUSED=0) bit for bit - interior, edge and
    off-grid tiles, a frame without mask, a frame that misses the grid."""
    z = pkg()
    s = synth()
    frames, _ = stack(6, 520, 480, 900, dither=40.0, rot=0.3)
    frames[2]['mask'] = None
    frames.append(dict(frames[0], wcs=s.ztf_wcs(520, 480, dx=5000.0, dy=-4000.0)))
    wout = engine.autogrid([f['wcs'] for f in frames[:6]])
    p = z.coadd_params(combine=kind, mask_combine='AND', subtract_back=True, rescale_weights=True, back_size=64)
    a, b = run_both(engine, frames, wout, p)
    assert_same(a, b)
    assert (b[1] > 0).mean() > 0.5 and (b[3] == 0).any()


def test_stack_mode_stack_and_partial_mask_equal_k_resample(engine):
    """zm_resample_stack_dev: every sample of the resident stack and the partial mask coadd (-1 markers),
    aligned grids (delta kernels) and a rotated frame whose footprint exceeds the LDS tile included."""
    import importlib
    z = pkg()
    s = synth()
    par = importlib.import_module('zuds-pipeline_amd.parallel')
    frames, base = stack(4, 450, 430, 950)
    frames[1]['wcs'] = s.ztf_wcs(450, 430, dx=-7.0, dy=3.0)            # integer shift: delta kernels
    frames[3]['wcs'] = s.ztf_wcs(450, 430, rot_deg=25.0)
    frames[0]['mask'][100:140, 200:260] |= 1 << 16
    p = z.coadd_params(combine='CLIPPED', mask_combine='OR', subtract_back=False, rescale_weights=False)
    res = {}
    for mode in ('0', '1'):
        os.environ['ZM_COADD_FUSED'] = mode
        try:
            be = par.HipBackend(base, p, device=0, engine=engine)
            st = be.resample_stack(frames, want_mask=True)
            be.stream.synchronize()
            res[mode] = (st.cpu().numpy(), be.partial_mask.cpu().numpy())
        finally:
            os.environ.pop('ZM_COADD_FUSED', None)
            engine.set_stream(0)
    assert np.array_equal(res['0'][0], res['1'][0])
    assert np.array_equal(res['0'][1], res['1'][1])
    assert (res['1'][1] == -1).any() and (res['1'][0][..., 1] > 0).mean() > 0.5


@pytest.mark.parametrize('kind', ['WEIGHTED', 'CLIPPED'])
def test_ragged_stack_with_odd_widths(engine, kind):
    """Frames of different sizes in one stack, widths that are not multiples of 4 (no vector staging of the
    mask box: such frames' edge items go through the generic code), one frame without weights."""
    z = pkg()
    s = synth()
    rng = np.random.default_rng(77)
    base = s.ztf_wcs(420, 400, tpv=True)
    frames = []
    for i, (nx, ny) in enumerate([(333, 290), (400, 410), (257, 300), (420, 400), (391, 377)]):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-20, 20), dy=rng.uniform(-20, 20), rot_deg=rng.uniform(-0.3, 0.3), tpv=True)
        frames.append(s.make_frame(nx, ny, 1300 + i, w, nstars=30, nbad=200))
    frames[3]['wgt'] = None
    p = z.coadd_params(combine=kind, mask_combine='OR', subtract_back=True, rescale_weights=True, back_size=64)
    a, b = run_both(engine, frames, base, p)
    assert_same(a, b)
    assert (b[1] > 0).mean() > 0.6


def test_differential_fuzz_against_the_k_resample_path(engine):
    """80 random stacks (depth, sizes, rotations up to 40 degrees, scale changes, integer and fractional
    dithers, masks with and without high bits, missing masks / weights, every combine and mask-combine
    type, backgrounds on / off) through tools/fuzz_coadd.py: fused and k_resample paths agree bit for bit
    (3 000 cases were run once when the STACK form went in: no mismatch)."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location(
        'fuzz_coadd', pathlib.Path(__file__).resolve().parent.parent / 'tools' / 'fuzz_coadd.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(80, 2026, eng=engine, verbose=False) == 0


DEV_ONLY = ('ZM_FF_FORK', 'ZM_FF_YIELD', 'ZM_FF_PROF', 'ZM_FF_DEAL', 'ZM_FF_PRIO')


@pytest.mark.parametrize('variant', [{'ZM_FF_RAW': '0'}, {'ZM_FF_FORM': 'dma'}, {'ZM_FF_FORM': 'dma', 'ZM_FF_RAW': '0'},
                                     {'share': 2}, {'share': 2, 'ZM_FF_FORM': 'dma'},
                                     {'ZM_FF_FORK': '0'}, {'ZM_FF_YIELD': '1', 'ZM_FF_FORM': 'dma'},
                                     {'ZM_FF_PROF': '1', 'ZM_FF_DEAL': '0'}, {'ZM_FF_PROF': '1', 'ZM_FF_DEAL': '2', 'ZM_FF_PRIO': '10'}])
def test_kernel_variants_behind_the_switches(engine, monkeypatch, variant):
    """The library ships two fused kernels (the owner-staged one - in-place prep, one barrier per item - wherever the
    footprints fit its fixed slot; the LDS-DMA staged one elsewhere or with ZM_FF_FORM=dma; round 6 removed the
    register-staged one) and two ways to feed them (raw planes prepped in the kernel, the default, and planes
    prepped ahead: ZM_FF_RAW=0, also what frames without 16-byte rows or with large footprints take); the yield mode
    of a context that shares the GPU (zm_ctx_set_share >= 2: workgroups retire after two tiles).  A developer build
    (-DZM_DEV) adds instances with phase clocks and staging deals behind ZM_FF_PROF / ZM_FF_DEAL / ZM_FF_PRIO /
    ZM_FF_YIELD / ZM_FF_FORK, which the shipped library does not read.  Every combination gives the bits of the
    materialised path, on interior and edge tiles."""
    if any(k in DEV_ONLY for k in variant) and not engine.query('dev_build'):
        pytest.skip('developer switch: the library was not built with -DZM_DEV')
    variant = dict(variant)
    share = variant.pop('share', 1)
    engine.set_share(share)
    try:
        _variant_body(engine, monkeypatch, variant, 'dma' if variant.get('ZM_FF_FORM') == 'dma' else 'own')
    finally:
        engine.set_share(1)


def _variant_body(engine, monkeypatch, variant, form):
    z = pkg()
    for k, v in variant.items():
        monkeypatch.setenv(k, v)
    frames, wout = stack(5, 700, 650, 310)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='OR', subtract_back=True, rescale_weights=True, back_size=128)
    a, b = run_both(engine, frames, wout, p)
    assert_same(a, b)
    # which kernel the launcher chose is the context's to say (zm_ctx_query; ADVICE r5: bench.py labels its roofline
    # with it): near-unit scale footprints fit the owner-staged slots
    assert engine.query('fused_form') == {'own': 2, 'dma': 1}[form]
    frames, _ = stack(4, 520, 480, 320, dither=50.0, rot=0.3)
    wout = engine.autogrid([f['wcs'] for f in frames])
    p = z.coadd_params(combine='AVERAGE', mask_combine='AND', subtract_back=False, rescale_weights=False)
    a, b = run_both(engine, frames, wout, p)
    assert_same(a, b)
