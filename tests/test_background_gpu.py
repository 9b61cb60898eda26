"""Parity of the mesh background kernels with the oracle (SExtractor back.c)."""
import numpy as np
import pytest

from oracle import background as oback
from util import assert_close_masked, pkg, synth

pytestmark = pytest.mark.gpu


def make(nx, ny, seed, grad=True, nstars=60):
    s = synth()
    f = s.make_frame(nx, ny, seed, s.tan_wcs(nx, ny), sky=180.0, noise=6.0,
                     nstars=nstars, nbad=200)
    if grad:
        yy, xx = np.mgrid[0:ny, 0:nx]
        f['img'] = (f['img'] + 0.03 * xx - 0.015 * yy
                    + 4 * np.sin(xx / 90.0)).astype(np.float32)
    return f


@pytest.mark.parametrize('shape,mesh', [((512, 512), 128), ((300, 280), 64),
                                         ((257, 130), 64), ((100, 90), 128),
                                         ((560, 540), 16),    # 35 x 34 = 1190 meshes: generic filter path
                                         ((600, 560), 256)])  # meshes of more than 16384 px: generic statistics kernel
def test_background_matches_oracle(engine, shape, mesh):
    nx, ny = shape
    f = make(nx, ny, nx + ny)
    bkg, rms, sub, stats = engine.background(f['img'], f['wgt'], mesh=mesh)
    r_bkg, r_rms, r_mean, r_sig, _, _ = oback.background(
        f['img'].astype(np.float64), f['wgt'].astype(np.float64), mesh)
    # tolerance: 1e-5 relative on ~180 counts; fp32 node storage + fp32 spline
    assert_close_masked(bkg, r_bkg, 2e-5, 1e-3, 'background')
    assert_close_masked(rms, r_rms, 1e-4, 1e-4, 'background rms')
    np.testing.assert_allclose(sub, f['img'] - bkg, atol=1e-4)
    assert abs(stats[0] - r_mean) < 2e-3 and abs(stats[1] - r_sig) < 1e-3


@pytest.mark.parametrize('fsize', [1, 5, 7])
@pytest.mark.parametrize('shape,mesh', [((512, 480), 64), ((560, 540), 16), ((300, 100), 128)])
def test_other_filter_sizes_match_the_oracle(engine, shape, mesh, fsize):
    """BACK_FILTERSIZE other than the reference's 3 (``sextractor.conf:70``): the window median by rank counting on
    the staged maps (round 6: no per-thread window), clipped windows at the borders included (300 x 100 at 128: a
    3 x 1 mesh map, every window clipped); a mesh masked out so that the fill runs in front of the filter."""
    nx, ny = shape
    f = make(nx, ny, 7 * fsize + mesh)
    w = f['wgt'].copy()
    w[:mesh, :mesh] = 0
    bkg, rms, _, stats = engine.background(f['img'], w, mesh=mesh, filtersize=fsize)
    r_bkg, r_rms, r_mean, r_sig, _, _ = oback.background(f['img'].astype(np.float64), w.astype(np.float64), mesh, fsize)
    assert_close_masked(bkg, r_bkg, 2e-5, 1e-3, 'background')
    assert_close_masked(rms, r_rms, 1e-4, 1e-4, 'background rms')
    assert abs(stats[0] - r_mean) < 2e-3 and abs(stats[1] - r_sig) < 1e-3


def test_no_weight_map_and_flat_image(engine):
    img = np.full((256, 256), 42.0, dtype=np.float32)
    bkg, rms, sub, stats = engine.background(img, None, mesh=64)
    np.testing.assert_allclose(bkg, 42.0, atol=1e-4)
    np.testing.assert_allclose(rms, 0.0, atol=1e-6)
    assert abs(stats[0] - 42.0) < 1e-4


def test_bad_meshes_are_filled_from_neighbours(engine):
    f = make(384, 384, 5, grad=False)
    w = f['wgt'].copy()
    w[128:256, 128:256] = 0          # the centre mesh is entirely masked
    bkg, rms, _, _ = engine.background(f['img'], w, mesh=128)
    r_bkg, r_rms, *_ = oback.background(f['img'].astype(np.float64), w.astype(np.float64), 128)
    assert_close_masked(bkg, r_bkg, 2e-5, 1e-3, 'filled background')
    assert abs(np.median(bkg[128:256, 128:256]) - 180.0) < 2.0
