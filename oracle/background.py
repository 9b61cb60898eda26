"""Mesh background / background-RMS restatement (oracle; test infrastructure).

Operator definition from the reference: SWarp ``SUBTRACT_BACK Y``,
``BACK_TYPE AUTO``, ``BACK_FILTERSIZE 3``, ``BACK_FILTTHRESH 0``
(``zuds/astromatic/makecoadd/default.swarp:77-88``) with ``-BACK_SIZE 128``
(``zuds/swarp.py:69``, ``zuds/constants.py:4``); SExtractor ``-BACK_SIZE 128``,
``BACK_FILTERSIZE 3`` with check-images ``-BACKGROUND`` / ``BACKGROUND_RMS`` /
``BACKGROUND`` (``zuds/sextractor.py:21-26,74``,
``zuds/astromatic/sextractor.conf:67-72``); weight map ``MAP_WEIGHT`` with
``WEIGHT_THRESH 1e-30`` (``zuds/sextractor.py:80-100``).

Arithmetic: the published SExtractor/SWarp ``back.c`` algorithm (Bertin & Arnouts
1996, section 7.1): per mesh 2-sigma pre-clip, quantised histogram over +-5 sigma,
iterated +-3 sigma clipping around the histogram median, mode = 2.5 med - 1.5
mean when |mean - med| < 0.3 sigma else the median; bad meshes (< 50 % good
pixels) filled from the nearest good ones; 3x3 median filter of the mesh maps;
natural bicubic spline through mesh centres back to full resolution.
"""
import numpy as np

BIG = 1e30
QUANTIF_NSIGMA = 5
QUANTIF_NMAXLEVELS = 4096
QUANTIF_AMIN = 4
BACK_MINGOODFRAC = 0.5
WEIGHT_THRESH = 1e-30
EPS = 1e-4


def mesh_histogram_stats(pix):
    """Steps ``backstat`` + ``backhisto`` for one mesh.

    pix: 1-D float64 array of the *valid* pixels of the mesh.  Returns a dict
    with mean, sigma (2-sigma clipped), qzero, qscale, nlevels, histo, or None.
    """
    n = pix.size
    if n == 0:
        return None
    mean = pix.mean()
    sig = pix.var()
    sig = np.sqrt(sig) if sig > 0 else 0.0
    lcut = mean - 2.0 * sig
    hcut = mean + 2.0 * sig
    sel = pix[(pix >= lcut) & (pix <= hcut)]
    npix = sel.size
    if npix == 0:
        return None
    mean = sel.mean()
    sig = sel.var()
    sig = np.sqrt(sig) if sig > 0 else 0.0
    step = np.sqrt(2.0 / np.pi) * QUANTIF_NSIGMA / QUANTIF_AMIN
    nlevels = int(step * npix + 1)
    if nlevels > QUANTIF_NMAXLEVELS:
        nlevels = QUANTIF_NMAXLEVELS
    # SExtractor keeps mean, sigma, qscale, qzero and the bin offset in floats and
    # bins with (int)(pix / qscale + cste): float arithmetic, truncation toward 0
    f32 = np.float32
    mean32, sig32 = f32(mean), f32(sig)
    qscale = f32(2.0 * QUANTIF_NSIGMA * np.float64(sig32) / nlevels) if sig32 > 0 else f32(1.0)
    qzero = f32(np.float64(mean32) - QUANTIF_NSIGMA * np.float64(sig32))
    cste = f32(0.499999 - np.float64(qzero / qscale))
    b = np.trunc(pix.astype(f32) / qscale + cste).astype(np.int64)
    b = b[(b >= 0) & (b < nlevels)]
    histo = np.bincount(b, minlength=nlevels).astype(np.int64)
    return dict(mean=float(mean32), sigma=float(sig32), qzero=float(qzero), qscale=float(qscale),
                nlevels=nlevels, histo=histo)


def histogram_median_walk(histo, lcut, hcut):
    """The two-pointer median walk of ``backguess`` (sequential, authoritative)."""
    lowsum = highsum = 0
    lo = lcut
    hi = hcut
    for _ in range(lcut, hcut + 1):
        if lowsum < highsum:
            lowsum += int(histo[lo])
            lo += 1
        else:
            highsum += int(histo[hi])
            hi -= 1
    if hi < 0:
        return 0.0
    a = int(histo[lo]) if lo < len(histo) else 0
    b = int(histo[hi])
    den = 2.0 * max(a, b)
    frac = (highsum - lowsum) / den if den > 0 else 0.0
    return hi + 0.5 + frac


def backguess(st):
    """Iterated +-3 sigma clipping on the mesh histogram -> (mode, sigma)."""
    histo = st['histo']
    nlm1 = st['nlevels'] - 1
    lcut, hcut = 0, nlm1
    sig = 10.0 * nlm1
    sig1 = 1.0
    mea = med = st['mean']
    idx = np.arange(st['nlevels'], dtype=np.float64)
    n = 100
    while n > 0 and sig >= 0.1 and abs(sig / sig1 - 1.0) > EPS:
        n -= 1
        sig1 = sig
        h = histo[lcut:hcut + 1].astype(np.float64)
        ii = idx[lcut:hcut + 1]
        s = h.sum()
        mea = (h * ii).sum()
        sig = (h * ii * ii).sum()
        med = histogram_median_walk(histo, lcut, hcut)
        if s > 0:
            mea /= s
            sig = sig / s - mea * mea
        sig = np.sqrt(sig) if sig > 0 else 0.0
        ft = med - 3.0 * sig
        lcut = int(ft + 0.5) if ft > 0 else 0
        ft = med + 3.0 * sig
        hcut = (int(ft + 0.5) if ft > 0 else int(ft - 0.5)) if ft < nlm1 else nlm1
    qz, qs = st['qzero'], st['qscale']
    if sig > 0:
        if abs((mea - med) / sig) < 0.3:
            mode = qz + (2.5 * med - 1.5 * mea) * qs
        else:
            mode = qz + med * qs
    else:
        mode = qz + mea * qs
    return mode, sig * qs


def fqmedian(v):
    v = np.sort(np.asarray(v, dtype=np.float64))
    n = v.size
    if n == 0:
        return 0.0
    return v[n // 2] if n & 1 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def mesh_maps(img, wgt=None, mesh=128):
    """Raw per-mesh (mode, sigma) maps; bad meshes carry -BIG."""
    img = np.asarray(img, dtype=np.float64)
    ny, nx = img.shape
    nbx = (nx - 1) // mesh + 1
    nby = (ny - 1) // mesh + 1
    back = np.full((nby, nbx), -BIG)
    sigm = np.full((nby, nbx), -BIG)
    for j in range(nby):
        for i in range(nbx):
            y0, y1 = j * mesh, min((j + 1) * mesh, ny)
            x0, x1 = i * mesh, min((i + 1) * mesh, nx)
            p = img[y0:y1, x0:x1].ravel()
            ok = p > -BIG
            if wgt is not None:
                ok &= (wgt[y0:y1, x0:x1].ravel() > WEIGHT_THRESH)
            p = p[ok]
            area = (y1 - y0) * (x1 - x0)
            if p.size < area * BACK_MINGOODFRAC:
                continue
            st = mesh_histogram_stats(p)
            if st is None:
                continue
            npix2 = int(st['histo'].sum())
            if npix2 == 0:
                continue
            back[j, i], sigm[j, i] = backguess(st)
    return back, sigm


def filter_maps(back, sigm, fsize=3):
    """Fill bad meshes from the nearest good ones, then fsize x fsize median."""
    nby, nbx = back.shape
    b2 = back.copy()
    s2 = sigm.copy()
    good = back > -BIG
    gy, gx = np.nonzero(good)
    if gy.size:
        for j in range(nby):
            for i in range(nbx):
                if good[j, i]:
                    continue
                d2 = (gx - i) ** 2 + (gy - j) ** 2
                m = d2 == d2.min()
                b2[j, i] = back[gy[m], gx[m]].mean()
                s2[j, i] = sigm[gy[m], gx[m]].mean()
    else:
        b2[:] = 0.0
        s2[:] = 1.0
    hb = fsize // 2
    bo = b2.copy()
    so = s2.copy()
    if fsize > 1:
        for j in range(nby):
            for i in range(nbx):
                ys = slice(max(j - hb, 0), min(j + hb, nby - 1) + 1)
                xs = slice(max(i - hb, 0), min(i + hb, nbx - 1) + 1)
                bo[j, i] = fqmedian(b2[ys, xs].ravel())
                so[j, i] = fqmedian(s2[ys, xs].ravel())
    return bo, so


def spline_derivs(a):
    """Natural cubic spline second derivatives / 6 along axis 0 (unit spacing)."""
    n = a.shape[0]
    d = np.zeros_like(a)
    if n < 3:
        return d
    u = np.zeros_like(a)
    for y in range(1, n - 1):
        temp = -1.0 / (d[y - 1] + 4.0)
        d[y] = temp
        u[y] = temp * (u[y - 1] - 6.0 * (a[y + 1] + a[y - 1] - 2.0 * a[y]))
    d[n - 1] = 0.0
    for y in range(n - 2, 0, -1):
        d[y] = d[y] * d[y + 1] + u[y]
    d[0] = 0.0
    return d / 6.0


def _spline_eval_axis(nodes, derivs, npix, mesh):
    """Evaluate along axis 0 at the centres of ``npix`` pixels."""
    n = nodes.shape[0]
    t = (np.arange(npix, dtype=np.float64) + 0.5) / mesh - 0.5
    if n < 2:
        return np.repeat(nodes[:1], npix, axis=0)
    i0 = np.floor(t).astype(np.int64)
    i0 = np.clip(i0, 0, n - 2)
    dy = t - i0
    dy1 = 1.0 - dy
    cdy = dy * dy * dy - dy
    cdy1 = dy1 * dy1 * dy1 - dy1
    shp = (npix,) + (1,) * (nodes.ndim - 1)
    return (dy1.reshape(shp) * nodes[i0] + dy.reshape(shp) * nodes[i0 + 1]
            + cdy1.reshape(shp) * derivs[i0] + cdy.reshape(shp) * derivs[i0 + 1])


def expand(nodes, nx, ny, mesh=128):
    """Bicubic-spline a mesh map (nby, nbx) up to (ny, nx): spline along y per
    node column, then along x per image row (``backline``)."""
    dyy = spline_derivs(nodes)                       # (nby, nbx)
    rows = _spline_eval_axis(nodes, dyy, ny, mesh)   # (ny, nbx)
    rt = rows.T                                      # (nbx, ny)
    dxx = spline_derivs(rt)
    return _spline_eval_axis(rt, dxx, nx, mesh).T    # (ny, nx)


def background(img, wgt=None, mesh=128, fsize=3):
    """Full-resolution background and background-RMS images plus the global
    (backmean, backsig) = medians of the filtered mesh maps."""
    ny, nx = img.shape
    back, sigm = mesh_maps(img, wgt, mesh)
    bo, so = filter_maps(back, sigm, fsize)
    bkg = expand(bo, nx, ny, mesh)
    rms = expand(so, nx, ny, mesh)
    return bkg, rms, fqmedian(bo.ravel()), fqmedian(so.ravel()), bo, so


def quick_background_estimate(data, mask):
    """Restatement of ``zuds/utils.py:32-53`` without ``nsamp``: (median, 1.4826 * MAD) of the pixels whose mask is
    0, in the reference's arithmetic - the median and the absolute deviations stay in the image's dtype (float32
    for every frame on this path), the factor multiplies a float32 scalar as a float64 (numpy 1.x, which the
    reference ran on).  PINNED: tests/golden/reference_python.json holds the outputs of the reference's own
    function on 18 arrays (tests/golden/make_reference_python_golden.py); tests/test_golden.py compares bit for
    bit.  An all-masked frame gives (nan, nan), as there."""
    pix = np.asarray(data)[np.asarray(mask) == 0]
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            bkg = np.median(pix)
            mad = np.median(np.abs(pix - bkg))
    return bkg, 1.4826 * float(mad)
