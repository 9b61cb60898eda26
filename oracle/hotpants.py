"""Alard-Lupton difference imaging, hotpants-style (oracle; test infrastructure).

Operator definition from the reference (``zuds/hotpants.py:77-93``):
``hotpants -inim sci -tmplim ref -outim D -c t -n i -hki -tu 5e3 -iu 5e3 -tl .. -il ..
-r 2.5 SEEING -rss 6 SEEING -tni ref.rms -ini sci.rms -imi bpm -oni D.rms
-fin sqrt(50000) -nsx NAXIS1/100/nreg -nsy .. -nrx nreg -nry nreg -bgo 0 -ko 4 -v 0``;
masked output pixels carry 1e-30 (``zuds/subtraction.py:170-171``).

hotpants itself is an un-vendored submodule (``.gitmodules:1-3``), so the
arithmetic below restates the published algorithm (Alard & Lupton 1998; Alard
2000; hotpants 5.1.11 ``alard.c`` / ``functions.c`` as documented) with every
convention made explicit:

* valid pixel: bpm == 0, il <= I <= iu, tl <= T <= tu, finite.
* regions nrx x nry; stamps nsx x nsy per region (integer cell size, remainder
  unused); integer half widths hwk = int(r), hwss = int(rss).
* substamp centres (on the template, -c t): greedy brightest-first among pixels
  of the stamp cell with T >= sky + ft sig (3-pass 3-sigma clipped mean / std of
  the cell's valid pixels), whose (2 (hwss + hwk) + 1)^2 box is inside the image
  and free of invalid pixels; a chosen centre excludes its (2 hwss + 1)^2 box;
  ties -> lowest (y, x); at most nss per stamp.
* basis: Gaussians sigma_g x polynomials u^a v^b (a + b <= deg_g), ordered
  g, a, b; even-even terms normalised to unit sum and (all but the first) minus
  the first term, so the kernel sum is the first coefficient (spatially
  constant).
* vectors: true convolution W_n(x) = sum_u K_n(u) T(x - u) over the substamp;
  background terms x^i y^j (i + j <= bgo) per pixel in region-normalised
  coordinates; kernel coefficients are polynomials (order ko) of the substamp
  centre.
* unweighted least squares; Jacobi-scaled normal matrix plus a 1e-10 ridge on its
  unit diagonal (a small kernel cannot carry all 49 basis terms); Cholesky in
  float64.
* stamp merit m_s = sum (I - M)^2 / (Npix mean(sI^2 + sT^2)); reject
  m_s > mean + ks std (3-pass 3-sigma clipped moments); a rejected stamp moves to
  its next substamp; at most 8 rounds.
* apply: kernel re-evaluated at the centre of each (2 hwk + 1)^2 output block;
  D = I - (T (x) K + bg); noise = sqrt(sI^2 + sT^2 (x) K^2); -n t divides both by
  the kernel sum; outputs within hwk of an invalid pixel or of the frame edge are
  filled with fi / fin.
"""
import numpy as np

DEFAULTS = dict(tu=5e3, tl=0.0, iu=5e3, il=0.0, r=10.0, rss=15.0,
                fin=np.sqrt(50000.0), fi=1e-30, nsx=10, nsy=10, nrx=1, nry=1,
                ko=4, bgo=0, nss=3, normalize=0, ft=20.0, ks=2.0,
                deg=(6, 4, 2), sigma=(0.7, 1.5, 3.0))
MAX_ROUNDS = 8
RIDGE = 1e-10


def params(**kw):
    p = dict(DEFAULTS)
    p.update(kw)
    return p


def poly_terms(order):
    return [(i, j) for i in range(order + 1) for j in range(order + 1 - i)]


def basis_1d(hwk, degs, sigmas):
    """1-D filters f[g][a] (length 2 hwk + 1) and the 2-D term list (g, a, b)."""
    u = np.arange(-hwk, hwk + 1, dtype=np.float64)
    f = []
    terms = []
    for g, (deg, sig) in enumerate(zip(degs, sigmas)):
        ga = np.exp(-u * u / (2.0 * sig * sig))
        f.append([ga * u ** a for a in range(deg + 1)])
        for a in range(deg + 1):
            for b in range(deg + 1 - a):
                terms.append((g, a, b))
    return f, terms


def basis_2d(hwk, degs, sigmas):
    """Kernel basis K_n[v, u] (n, 2 hwk + 1, 2 hwk + 1), hotpants normalisation."""
    f, terms = basis_1d(hwk, degs, sigmas)
    ks = []
    for n, (g, a, b) in enumerate(terms):
        fx = f[g][a]
        fy = f[g][b]
        if a % 2 == 0 and b % 2 == 0:
            fx = fx / fx.sum()
            fy = fy / fy.sum()
        k = np.outer(fy, fx)
        if a % 2 == 0 and b % 2 == 0 and n > 0:
            k = k - ks[0]
        ks.append(k)
    return np.array(ks), terms


def convolve_true(img, k):
    """C(y, x) = sum_{v,u} K[v, u] T(y - v', x - u'), 'valid' region only."""
    from scipy.signal import correlate2d
    return correlate2d(img, k[::-1, ::-1], mode='valid')


def clipped_moments(v, nsig=3.0, passes=3):
    v = np.asarray(v, dtype=np.float64)
    if v.size == 0:
        return 0.0, 0.0
    m, s = v.mean(), v.std()
    for _ in range(passes):
        sel = v[np.abs(v - m) <= nsig * s]
        if sel.size == 0:
            break
        m, s = sel.mean(), sel.std()
    return m, s


def valid_mask(sci, ref, bpm, p):
    ok = np.isfinite(sci) & np.isfinite(ref)
    ok &= (sci >= p['il']) & (sci <= p['iu']) & (ref >= p['tl']) & (ref <= p['tu'])
    if bpm is not None:
        ok &= (np.asarray(bpm) == 0)
    return ok


def box_any(bad, hw):
    """True where any ``bad`` pixel lies within the (2 hw + 1)^2 box (clipped)."""
    ny, nx = bad.shape
    c = np.zeros((ny + 1, nx + 1), dtype=np.int64)
    c[1:, 1:] = np.cumsum(np.cumsum(bad.astype(np.int64), axis=0), axis=1)
    y0 = np.clip(np.arange(ny) - hw, 0, ny)
    y1 = np.clip(np.arange(ny) + hw + 1, 0, ny)
    x0 = np.clip(np.arange(nx) - hw, 0, nx)
    x1 = np.clip(np.arange(nx) + hw + 1, 0, nx)
    s = (c[y1][:, x1] - c[y0][:, x1] - c[y1][:, x0] + c[y0][:, x0])
    return s > 0


def regions(nx, ny, nrx, nry):
    out = []
    for ry in range(nry):
        for rx in range(nrx):
            x0 = rx * (nx // nrx)
            x1 = nx if rx == nrx - 1 else (rx + 1) * (nx // nrx)
            y0 = ry * (ny // nry)
            y1 = ny if ry == nry - 1 else (ry + 1) * (ny // nry)
            out.append((x0, x1, y0, y1))
    return out


def find_substamps(ref, ok, reg, p):
    """Per stamp cell: list of up to nss centres (x, y), brightest first."""
    ny, nx = ref.shape
    hwk, hwss = int(p['r']), int(p['rss'])
    hw = hwk + hwss
    x0, x1, y0, y1 = reg
    dirty = box_any(~ok, hw)
    inside = np.zeros_like(ok)
    inside[hw:ny - hw, hw:nx - hw] = True
    elig_all = ~dirty & inside
    cw = (x1 - x0) // p['nsx']
    ch = (y1 - y0) // p['nsy']
    stamps = []
    for sy in range(p['nsy']):
        for sx in range(p['nsx']):
            cx0, cy0 = x0 + sx * cw, y0 + sy * ch
            cell = (slice(cy0, cy0 + ch), slice(cx0, cx0 + cw))
            vals = ref[cell][ok[cell]]
            sky, sig = clipped_moments(vals)
            thr = sky + p['ft'] * sig
            t = ref[cell].astype(np.float64)
            el = elig_all[cell] & (t >= thr)
            centres = []
            for _ in range(p['nss']):
                if not el.any():
                    break
                tv = np.where(el, t, -np.inf)
                j = int(np.argmax(tv))          # first occurrence = lowest (y, x)
                yy, xx = divmod(j, cw)
                centres.append((cx0 + xx, cy0 + yy))
                el[max(yy - hwss, 0):yy + hwss + 1, max(xx - hwss, 0):xx + hwss + 1] = False
            stamps.append(centres)
    return stamps


def substamp_system(sci, ref, svar, tvar, cx, cy, basis, reg, p):
    """Extended vectors E (nc + nbg, npix) of one substamp and its Gram pieces."""
    hwk, hwss = int(p['r']), int(p['rss'])
    hw = hwk + hwss
    patch = ref[cy - hw:cy + hw + 1, cx - hw:cx + hw + 1].astype(np.float64)
    W = np.array([convolve_true(patch, k).ravel() for k in basis])
    x0, x1, y0, y1 = reg
    xc, hx = x0 + (x1 - x0) / 2.0, (x1 - x0) / 2.0
    yc, hy = y0 + (y1 - y0) / 2.0, (y1 - y0) / 2.0
    yy, xx = np.mgrid[cy - hwss:cy + hwss + 1, cx - hwss:cx + hwss + 1]
    xf = ((xx - xc) / hx).ravel()
    yf = ((yy - yc) / hy).ravel()
    B = np.array([xf ** i * yf ** j for (i, j) in poly_terms(p['bgo'])])
    E = np.concatenate([W, B], axis=0)
    I = sci[cy - hwss:cy + hwss + 1, cx - hwss:cx + hwss + 1].astype(np.float64).ravel()
    v = (svar[cy - hwss:cy + hwss + 1, cx - hwss:cx + hwss + 1].astype(np.float64)
         + tvar[cy - hwss:cy + hwss + 1, cx - hwss:cx + hwss + 1].astype(np.float64))
    return dict(Q=E @ E.T, b=E @ I, ii=float(I @ I), vbar=float(v.mean()),
                npix=I.size, fx=(cx - xc) / hx, fy=(cy - yc) / hy, cx=cx, cy=cy)


def expand_design(st, nc, nbg, kterms):
    """Map per-stamp extended index -> global unknowns with spatial weights.

    Returns (idx, wts): for extended vector e, the global columns it feeds and
    the weights phi_p(stamp)."""
    phi = np.array([st['fx'] ** i * st['fy'] ** j for (i, j) in kterms])
    nkp = len(kterms)
    cols, wts, src = [0], [1.0], [0]
    for n in range(1, nc):
        for pidx in range(nkp):
            cols.append(1 + (n - 1) * nkp + pidx)
            wts.append(phi[pidx])
            src.append(n)
    for q in range(nbg):
        cols.append(1 + (nc - 1) * nkp + q)
        wts.append(1.0)
        src.append(nc + q)
    return np.array(cols), np.array(wts), np.array(src)


def solve_region(systems, nc, nbg, ko):
    """Accumulate and solve the global normal equations of one region."""
    kterms = poly_terms(ko)
    nunk = 1 + (nc - 1) * len(kterms) + nbg
    A = np.zeros((nunk, nunk))
    rhs = np.zeros(nunk)
    for st in systems:
        cols, wts, src = expand_design(st, nc, nbg, kterms)
        Qe = st['Q'][np.ix_(src, src)] * np.outer(wts, wts)
        A[np.ix_(cols, cols)] += Qe
        rhs[cols] += wts * st['b'][src]
    d = np.sqrt(np.where(np.diag(A) > 0, np.diag(A), 1.0))
    As = A / np.outer(d, d)
    As[np.diag_indices_from(As)] += RIDGE     # keeps a rank-deficient basis solvable
    L = np.linalg.cholesky(As)
    y = np.linalg.solve(L, rhs / d)
    x = np.linalg.solve(L.T, y) / d
    return x, kterms


def stamp_merit(st, x, nc, nbg, kterms):
    cols, wts, src = expand_design(st, nc, nbg, kterms)
    c = np.zeros(nc + nbg)
    np.add.at(c, src, wts * x[cols])
    ss = st['ii'] - 2.0 * c @ st['b'] + c @ st['Q'] @ c
    return ss / (st['npix'] * st['vbar'])


def fit_region(sci, ref, svar, tvar, ok, reg, basis, p):
    """Kernel solution of one region: (x, kterms, info) or None if unsolvable."""
    nc = basis.shape[0]
    nbg = len(poly_terms(p['bgo']))
    cands = find_substamps(ref, ok, reg, p)
    ntotal = sum(1 for c in cands if c)
    active = [0 if c else -1 for c in cands]
    cache = {}

    def system(si):
        key = (si, active[si])
        if key not in cache:
            cx, cy = cands[si][active[si]]
            cache[key] = substamp_system(sci, ref, svar, tvar, cx, cy, basis, reg, p)
        return cache[key]

    x = kterms = None
    merits = []
    rounds = 0
    nfit = 0
    for rounds in range(1, MAX_ROUNDS + 1):
        live = [si for si in range(len(cands)) if active[si] >= 0]
        nunk = 1 + (nc - 1) * len(poly_terms(p['ko'])) + nbg
        if len(live) == 0:
            return None
        systems = [system(si) for si in live]
        nfit = len(systems)
        fitted = list(live)
        x, kterms = solve_region(systems, nc, nbg, p['ko'])
        merits = np.array([stamp_merit(st, x, nc, nbg, kterms) for st in systems])
        m, s = clipped_moments(merits)
        rej = [si for si, mm in zip(live, merits) if mm > m + p['ks'] * s]
        if not rej:
            break
        for si in rej:
            active[si] += 1
            if active[si] >= len(cands[si]):
                active[si] = -1
    # the solution in force is the one of the last solve: report its stamps
    info = dict(nstamps_total=ntotal, nstamps_used=nfit, niter=rounds,
                ncoeff=len(x), kernel_sum=float(x[0]),
                chi2=float(np.mean(merits)) if len(merits) else 0.0,
                fitted=fitted)
    return x, kterms, info


def kernel_at(x, kterms, basis, fx, fy):
    nc = basis.shape[0]
    nkp = len(kterms)
    phi = np.array([fx ** i * fy ** j for (i, j) in kterms])
    c = np.empty(nc)
    c[0] = x[0]
    c[1:] = x[1:1 + (nc - 1) * nkp].reshape(nc - 1, nkp) @ phi
    return np.tensordot(c, basis, axes=1)


def subtract(sci, ref, sci_rms, ref_rms, bpm, **kw):
    """Full difference image.  Returns (diff, noise, info)."""
    p = params(**kw)
    sci = np.asarray(sci, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    svar = np.asarray(sci_rms, dtype=np.float64) ** 2
    tvar = np.asarray(ref_rms, dtype=np.float64) ** 2
    ny, nx = sci.shape
    hwk = int(p['r'])
    ok = valid_mask(sci, ref, bpm, p)
    basis, _ = basis_2d(hwk, p['deg'], p['sigma'])
    nc = basis.shape[0]
    nbg = len(poly_terms(p['bgo']))
    bgt = poly_terms(p['bgo'])
    diff = np.full((ny, nx), p['fi'])
    noise = np.full((ny, nx), p['fin'])
    outbad = box_any(~ok, hwk)
    outbad[:hwk] = True
    outbad[ny - hwk:] = True
    outbad[:, :hwk] = True
    outbad[:, nx - hwk:] = True
    infos = []
    step = 2 * hwk + 1
    refz = np.where(np.isfinite(ref), ref, 0.0)
    tvz = np.where(np.isfinite(tvar), tvar, 0.0)
    for reg in regions(nx, ny, p['nrx'], p['nry']):
        fit = fit_region(sci, ref, svar, tvar, ok, reg, basis, p)
        if fit is None:
            infos.append(None)
            continue
        x, kterms, info = fit
        infos.append(info)
        x0, x1, y0, y1 = reg
        xc, hx = x0 + (x1 - x0) / 2.0, (x1 - x0) / 2.0
        yc, hy = y0 + (y1 - y0) / 2.0, (y1 - y0) / 2.0
        bgc = x[1 + (nc - 1) * len(kterms):]
        norm = 1.0 / x[0] if p['normalize'] else 1.0
        # blocks are anchored at the region origin; the kernel is evaluated at the
        # nominal block centre even when the block is clipped by the frame edge
        for gy in range(y0, y1, step):
            for gx in range(x0, x1, step):
                by, bx = max(gy, hwk), max(gx, hwk)
                ey, ex = min(gy + step, y1, ny - hwk), min(gx + step, x1, nx - hwk)
                if ey <= by or ex <= bx:
                    continue
                cyb, cxb = gy + hwk, gx + hwk
                K = kernel_at(x, kterms, basis, (cxb - xc) / hx, (cyb - yc) / hy)
                pt = refz[by - hwk:ey + hwk, bx - hwk:ex + hwk]
                pv = tvz[by - hwk:ey + hwk, bx - hwk:ex + hwk]
                conv = convolve_true(pt, K)
                cvar = convolve_true(pv, K * K)
                yy, xx = np.mgrid[by:ey, bx:ex]
                xf, yf = (xx - xc) / hx, (yy - yc) / hy
                bg = sum(bgc[q] * xf ** i * yf ** j for q, (i, j) in enumerate(bgt))
                d = (sci[by:ey, bx:ex] - conv - bg) * norm
                nz = np.sqrt(np.maximum(svar[by:ey, bx:ex] + cvar, 0.0)) * abs(norm)
                good = ~outbad[by:ey, bx:ex]
                diff[by:ey, bx:ex] = np.where(good, d, p['fi'])
                noise[by:ey, bx:ex] = np.where(good, nz, p['fin'])
    return diff, noise, dict(regions=infos, nmasked=int((diff == p['fi']).sum()))
