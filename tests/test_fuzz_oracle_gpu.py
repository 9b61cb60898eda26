"""Randomised parity of zm_coadd with the oracle: geometry the hand-picked cases do not visit - rotations of
tens of degrees, pixel scale ratios from 0.7 to 1.5, TAN and TPV, frames larger and smaller than the grid,
integer shifts (delta kernels on one or both axes) - at the tolerances of test_coadd_gpu.py."""
import os

import numpy as np
import pytest

from test_coadd_gpu import oracle_coadd
from util import assert_close_masked, pkg, synth

pytestmark = pytest.mark.gpu


def random_case(rng, seed):
    s = synth()
    onx, ony = int(rng.integers(110, 230)), int(rng.integers(110, 230))
    tpv = bool(rng.integers(0, 2))
    base = s.ztf_wcs(onx, ony, tpv=tpv)
    frames = []
    for i in range(int(rng.integers(2, 6))):
        nx, ny = (onx, ony) if rng.random() < 0.5 else (int(rng.integers(100, 260)), int(rng.integers(100, 260)))
        rot = rng.uniform(-35, 35) if rng.random() < 0.3 else rng.uniform(-0.4, 0.4)
        if rng.random() < 0.2:
            dx, dy, rot = float(rng.integers(-9, 9)), float(rng.integers(-9, 9)), 0.0
            if rng.random() < 0.5:
                dy += 0.37                                     # delta kernel along x only
        else:
            dx, dy = rng.uniform(-25, 25, 2)
        w = s.ztf_wcs(nx, ny, dx=float(dx), dy=float(dy), rot_deg=float(rot), tpv=tpv)
        if rng.random() < 0.3:
            w.cd = np.asarray(w.cd) * float(rng.uniform(0.7, 1.5))
        frames.append(s.make_frame(nx, ny, seed * 100 + i, w, nstars=15, nbad=int(rng.integers(0, 120)),
                                   magzp=float(rng.uniform(25.5, 26.5))))
    return frames, base


@pytest.mark.parametrize('seed', range(int(os.environ.get('ZM_FUZZ_SEEDS', '10'))))
def test_random_geometry_matches_the_oracle(engine, seed):
    z = pkg()
    rng = np.random.default_rng(4200 + seed)
    frames, wout = random_case(rng, 4200 + seed)
    kind = ['WEIGHTED', 'CLIPPED', 'MEDIAN', 'AVERAGE'][seed % 4]
    p = z.coadd_params(combine=kind, subtract_back=False, rescale_weights=False)
    g_img, g_wgt, g_msk, g_mw = engine.coadd(frames, wout, p)
    r_img, r_wgt, r_msk, vals, wgts = oracle_coadd(frames, wout, kind)
    gv, rv = g_wgt > 0, r_wgt > 0
    # (a footprint that grazes the frame edge, a fraction within 1e-5 of the snap rule, a sample on the clip
    # boundary change a pixel discretely: a few in ten thousand)
    assert (gv != rv).mean() < 5e-4
    both = gv & rv
    assert both.mean() > 0.3
    assert_close_masked(g_img[both], r_img[both], 3e-5, 3e-5 * 5.0, kind, max_bad_frac=5e-4)
    assert_close_masked(g_wgt[both], r_wgt[both], 1e-4, 0, kind + ' weight', max_bad_frac=5e-4)
    assert (g_msk != r_msk).mean() < 5e-4


@pytest.mark.parametrize('seed', range(int(os.environ.get('ZM_FUZZ_SEEDS', '6'))))
def test_random_subtractions_match_the_oracle(engine, seed):
    """Random scene, kernel half width, regions, cells and orders through zm_subtract and the hotpants
    restatement: fill pattern, stamp counts, rounds, kernel sum, difference and noise at the tolerances
    of test_subtract_gpu.py."""
    from test_subtract_gpu import COMMON, compare, scene
    rng = np.random.default_rng(7700 + seed)
    nx, ny = int(rng.integers(230, 330)), int(rng.integers(230, 330))
    data = scene(nx=nx, ny=ny, seed=7700 + seed, nstars=int(nx * ny / 700), ksig=float(rng.uniform(0.7, 1.3)),
                 scale=float(rng.uniform(0.8, 1.5)), bg=float(rng.uniform(0, 30)),
                 gradient=float(rng.choice([0.0, 0.0, 0.3])), nbad=int(rng.integers(0, 8)))
    hwk = int(rng.integers(3, 7))
    kw = dict(COMMON, r=float(hwk), rss=float(hwk + int(rng.integers(4, 9))), nrx=int(rng.integers(1, 3)),
              nry=int(rng.integers(1, 3)), nsx=int(rng.integers(2, 5)), nsy=int(rng.integers(2, 5)),
              ko=int(rng.integers(0, 3)), bgo=int(rng.integers(0, 2)))
    # a well-posed fit: three stamps per spatial term of the kernel at least (with fewer usable stamps
    # than terms the normal matrix is singular up to the ridge, and two solvers that agree at the stamps
    # differ by counts in between - seeds 16 and 19 of the first run)
    while (kw['ko'] + 1) * (kw['ko'] + 2) // 2 * 3 > kw['nsx'] * kw['nsy']:
        kw['ko'] -= 1
    d, n, info, rd = compare(engine, data, **kw)
    assert info['status'] == 0


@pytest.mark.parametrize('seed', range(int(os.environ.get('ZM_FUZZ_SEEDS', '8'))))
def test_random_backgrounds_match_the_oracle(engine, seed):
    """Random frame sizes (ragged against the mesh), mesh sizes, sky gradients, masked blocks (whole meshes
    among them) and weight maps that vary inside a mesh, through zm_background and the SExtractor
    restatement, at the tolerances of test_background_gpu.py."""
    from oracle import background as oback
    s = synth()
    rng = np.random.default_rng(9100 + seed)
    mesh = int(rng.choice([16, 32, 64, 128]))
    nx, ny = int(rng.integers(2 * mesh + 3, 5 * mesh + 40)), int(rng.integers(2 * mesh + 3, 5 * mesh + 40))
    f = s.make_frame(nx, ny, 9100 + seed, s.tan_wcs(nx, ny), sky=float(rng.uniform(50, 400)), noise=float(rng.uniform(2, 9)),
                     nstars=int(nx * ny / 4000) + 3, nbad=int(rng.integers(0, 300)))
    yy, xx = np.mgrid[0:ny, 0:nx]
    img = (f['img'] + rng.uniform(-0.05, 0.05) * xx + rng.uniform(-0.05, 0.05) * yy
           + rng.uniform(0, 6) * np.sin(xx / rng.uniform(40, 200))).astype(np.float32)
    wgt = f['wgt'].copy()
    for _ in range(int(rng.integers(0, 4))):
        x0, y0 = int(rng.integers(0, nx - 8)), int(rng.integers(0, ny - 8))
        wgt[y0:y0 + int(rng.integers(4, 2 * mesh)), x0:x0 + int(rng.integers(4, 2 * mesh))] = 0
    if seed % 3 == 0:
        wgt = (wgt * (1.0 + 0.3 * np.sin(xx / 37.0) * np.cos(yy / 23.0))).astype(np.float32)
    if seed % 4 == 3:
        wgt = None
    bkg, rms, sub, stats = engine.background(img, wgt, mesh=mesh)
    r_bkg, r_rms, r_mean, r_sig, _, _ = oback.background(img.astype(np.float64),
                                                         None if wgt is None else wgt.astype(np.float64), mesh)
    assert_close_masked(bkg, r_bkg, 2e-5, 1e-3, 'background')
    assert_close_masked(rms, r_rms, 1e-4, 1e-4, 'background rms')
    np.testing.assert_allclose(sub, img - bkg, atol=1e-4)
    assert abs(stats[0] - r_mean) < 2e-3 and abs(stats[1] - r_sig) < 1e-3


@pytest.mark.parametrize('seed', range(int(os.environ.get('ZM_FUZZ_SEEDS', '12'))))
def test_random_median_mad_is_numpy(engine, seed):
    """quick_background_estimate (zuds/utils.py:32-53) on random sizes and value distributions - heavy
    ties, negative values, denormals, huge dynamic range, one or two unmasked pixels - equals np.median
    of the float32 values exactly (radix select on the bit patterns)."""
    rng = np.random.default_rng(5300 + seed)
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 1000, 4097, 100003, 640 * 611]))
    kind = seed % 6
    if kind == 0:
        a = rng.normal(150, 8, n)
    elif kind == 1:
        a = rng.integers(-3, 4, n).astype(np.float64)                 # ties
    elif kind == 2:
        a = np.exp(rng.uniform(-80, 80, n)) * rng.choice([-1, 1], n)    # dynamic range, both signs
    elif kind == 3:
        a = rng.normal(0, 1e-41, n)                                   # denormals
    elif kind == 4:
        a = np.full(n, 42.5)
    else:
        a = rng.standard_cauchy(n)
    a = a.astype(np.float32)
    mask = (rng.random(n) < rng.uniform(0, 0.6)).astype(np.int32) * 256
    if mask.all():
        mask[int(rng.integers(0, n))] = 0
    med, mad = engine.median_mad(a, mask)
    pix = a[mask == 0]                      # float32, as image.data of the reference: numpy's median stays in float32
    with np.errstate(over='ignore', invalid='ignore'):
        rmed = np.median(pix)
        rmad = 1.4826 * np.median(np.abs(pix - rmed))
    assert med == float(rmed), (n, kind, med, rmed)
    if np.isfinite(rmad):
        assert abs(mad - rmad) <= 1e-6 * max(abs(rmad), 1e-30), (n, kind, mad, rmad)
