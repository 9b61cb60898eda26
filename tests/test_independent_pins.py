"""Independent pins for the interpolation kernels: Pillow (Lanczos-3) and scipy (bilinear).

Neither SWarp nor its sources exist here, so the resampling oracle (`oracle/resample.py`) is a restatement of the
published kernel - 6 taps sinc(d) sinc(d / 3) per axis, normalised to unit sum (`RESAMPLING_TYPE LANCZOS3`,
zuds/astromatic/makecoadd/default.swarp:51).  Pillow (`Image.resize(..., Image.LANCZOS, box=...)`, mode "F") implements
the SAME published kernel independently (libImaging/Resample.c: support 3, coefficients normalised to unit sum,
separable, double accumulators) and is installed on the build container and the GPU box.  For a pure translation by a
fraction of a pixel at unit scale the two must agree in the interior of the frame (they differ, by design, at the
edges - Pillow truncates and renormalises the kernel there, the oracle follows the conventions of DESIGN.md section 2 -
and within 1e-5 of an integer offset, where SWarp and the oracle switch to a delta kernel).  This pins the tap
values, their normalisation and the separable form; it does not pin SWarp's edge and snap conventions.
"""
import numpy as np
import pytest

from oracle import resample as oresample

PIL = pytest.importorskip('PIL')
from PIL import Image  # noqa: E402

SHIFTS = [(0.3, 0.7), (0.5, 0.5), (0.123, 0.9), (0.75, 0.25), (0.999, 0.001), (0.05, 0.62)]
MARGIN = 4           # output pixels from the edge of the box that are compared nowhere (Pillow's truncated kernels)
PAD = 6              # the box sits this far inside the input


def scene(nx=211, ny=187, seed=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
    img = 120.0 + 0.05 * xx - 0.03 * yy + rng.normal(0.0, 4.0, (ny, nx))
    for _ in range(60):
        x0, y0, f = rng.uniform(5, nx - 5), rng.uniform(5, ny - 5), np.exp(rng.uniform(np.log(2e2), np.log(3e4)))
        s = rng.uniform(0.8, 1.6)
        img += f / (2 * np.pi * s * s) * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * s * s))
    return img.astype(np.float32)


def pillow_shift(img, sx, sy):
    """Output pixel (x, y) samples the input at (x + PAD + sx, y + PAD + sy) (0-based pixel centres)."""
    ny, nx = img.shape
    ow, oh = nx - 2 * PAD - 2, ny - 2 * PAD - 2
    box = (PAD + sx, PAD + sy, PAD + sx + ow, PAD + sy + oh)
    out = Image.fromarray(img, mode='F').resize((ow, oh), Image.LANCZOS, box=box)
    return np.asarray(out, dtype=np.float64), ow, oh


@pytest.mark.parametrize('sx, sy', SHIFTS)
def test_oracle_lanczos3_translation_equals_pillow(sx, sy):
    img = scene()
    ref, ow, oh = pillow_shift(img, sx, sy)
    yo, xo = np.mgrid[0:oh, 0:ow].astype(np.float64)
    out, wgt, _ = oresample.resample(img, None, xo + PAD + sx, yo + PAD + sy, oresample.LANCZOS3)
    inner = (slice(MARGIN, oh - MARGIN), slice(MARGIN, ow - MARGIN))
    assert np.all(wgt[inner] > 0)
    # Pillow stores a float32 row pass between its two passes and a float32 result: 3e-7 of the peak
    tol = 3e-7 * float(np.abs(img).max())
    err = np.abs(out[inner] - ref[inner]).max()
    assert err <= tol, f'oracle vs Pillow: {err:.3e} > {tol:.3e}'


@pytest.mark.gpu
@pytest.mark.parametrize('sx, sy', SHIFTS[:4])
def test_hip_lanczos3_translation_equals_pillow(engine, sx, sy):
    """The HIP resampler (zm_resample through the C-ABI) against Pillow directly: a TAN frame dithered by (sx, sy)
    pixels onto the undithered grid.  Tolerance: the kernel's own (fp32 taps from the table, fp32 sums): 2e-5 of the
    local scale, as against the oracle (tests/test_resample_gpu.py)."""
    from util import synth
    s = synth()
    img = scene()
    ny, nx = img.shape
    ref, ow, oh = pillow_shift(img, sx, sy)
    # input pixel = output pixel + (PAD + s): CRPIX of the input grid is larger by that much
    win = s.tan_wcs(nx, ny, dx=PAD + sx, dy=PAD + sy)
    wout = s.tan_wcs(nx, ny)
    wout = type(wout)(wout.crpix, wout.crval, wout.cd, None, None, (ow, oh))
    got, gw, _ = engine.resample(img, win, wout)
    inner = (slice(MARGIN, oh - MARGIN), slice(MARGIN, ow - MARGIN))
    assert np.all(gw[inner] > 0)
    scale = float(np.std(img))
    err = np.abs(got[inner].astype(np.float64) - ref[inner])
    lim = 2e-5 * np.abs(ref[inner]) + 2e-5 * scale
    assert (err <= lim).all(), f'HIP vs Pillow: worst excess {np.max(err - lim):.3e}'


def test_oracle_bilinear_at_arbitrary_positions_equals_scipy():
    """`RESAMPLING_TYPE BILINEAR` (north_star's second kernel) against scipy.ndimage.map_coordinates(order=1), an
    independent implementation, at the positions of a rotated, rescaled, dithered map - the interpolation at arbitrary
    positions, not only translations."""
    from scipy import ndimage
    img = scene(173, 161, seed=8).astype(np.float64)
    ny, nx = img.shape
    yo, xo = np.mgrid[0:ny, 0:nx].astype(np.float64)
    c, s = np.cos(np.deg2rad(7.0)), np.sin(np.deg2rad(7.0))
    px = 1.03 * (c * (xo - nx / 2) - s * (yo - ny / 2)) + nx / 2 + 0.37
    py = 1.03 * (s * (xo - nx / 2) + c * (yo - ny / 2)) + ny / 2 - 0.81
    out, wgt, _ = oresample.resample(img, None, px, py, oresample.BILINEAR)
    ref = ndimage.map_coordinates(img, [py, px], order=1, mode='constant', cval=0.0)
    inside = (px >= 1) & (px <= nx - 2) & (py >= 1) & (py <= ny - 2)
    assert inside.mean() > 0.5 and np.all(wgt[inside] > 0)
    err = np.abs(out[inside] - ref[inside]).max()
    assert err <= 1e-9 * float(np.abs(img).max()), f'oracle vs scipy: {err:.3e}'


def test_oracle_background_expansion_equals_scipy_natural_splines():
    """The mesh map -> full-resolution background step (SExtractor's natural bicubic spline, `BACK_SIZE 128`,
    zuds/astromatic/sextractor.conf:67-72; oracle/background.py::expand) against scipy.interpolate.CubicSpline with
    natural end conditions: along y per mesh column at the pixel rows, then along x per image row, extrapolating the end
    intervals beyond the outermost mesh centres as the oracle (and SExtractor) do."""
    from scipy.interpolate import CubicSpline
    from oracle import background as obk
    rng = np.random.default_rng(5)
    nby, nbx, mesh = 7, 9, 32
    ny, nx = nby * mesh - 5, nbx * mesh - 11                     # (ragged last meshes)
    nodes = 100.0 + rng.normal(0.0, 3.0, (nby, nbx)) + np.linspace(0, 8, nbx)[None, :]
    got = obk.expand(nodes, nx, ny, mesh)
    ty = (np.arange(ny) + 0.5) / mesh - 0.5
    tx = (np.arange(nx) + 0.5) / mesh - 0.5
    rows = CubicSpline(np.arange(nby), nodes, axis=0, bc_type='natural', extrapolate=True)(ty)        # (ny, nbx)
    ref = CubicSpline(np.arange(nbx), rows, axis=1, bc_type='natural', extrapolate=True)(tx)          # (ny, nx)
    err = np.abs(got - ref).max()
    assert err <= 1e-10 * np.abs(nodes).max(), f'oracle vs scipy natural spline: {err:.3e}'
