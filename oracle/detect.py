"""Pixel-only seeing estimate and detection cuts (oracle; test infrastructure).

The reference measures the seeing as the median SExtractor ``FWHM_IMAGE`` of catalog sources
matched to Gaia stars (``zuds/seeing.py:10-118``) and filters subtraction candidates with
SExtractor columns plus three pixel tests (``zuds/filterobjects.py:57-195``).  Catalogs and
the Gaia / Kowalski queries are out of reach offline; what is restated
here is the pixel arithmetic, as chosen conventions:

* stars = isolated local maxima: strictly the largest pixel of the (2 iso + 1)^2 box around
  them (ties to the first pixel in raster order), ``lo < peak < hi`` (hi = half of SATURATE:
  unsaturated), no bad or NaN pixel in the box, ``border`` pixels away from the edges; the
  ``nmax`` brightest are used (sorted by peak, then y, then x);
* FWHM = 2.3548 sigma from adaptive Gaussian-weighted second moments: weight
  ``exp(-r^2 / 2 s_w^2)`` about the current centroid, measured ``m2 = sum(w I r^2) / (2 sum(w I))``,
  de-weighted ``sigma^2 = 1 / (1 / m2 - 1 / s_w^2)``, ``s_w^2 <- sigma^2`` until it settles
  (exact for a Gaussian star whatever the start);
* seeing = median of the finite FWHMs (``np.nanmedian`` as ``zuds/seeing.py:113``);
* negative-pixel cut exactly as ``filterobjects.py:155-195`` except that cutouts are clipped at
  the frame edges (numpy's negative slice indices wrap around there).
"""
import numpy as np


def find_stars(img, bad, lo, hi, iso=5, border=12, nmax=300):
    img = np.asarray(img, dtype=np.float32)
    ny, nx = img.shape
    ok = (img > lo) & (img < hi)
    ok[:border] = ok[-border:] = False
    ok[:, :border] = ok[:, -border:] = False
    ys, xs = np.nonzero(ok)
    out = []
    for y, x in zip(ys, xs):
        v = img[y, x]
        box = img[y - iso:y + iso + 1, x - iso:x + iso + 1]
        if np.isnan(box).any():
            continue
        if bad is not None and bad[y - iso:y + iso + 1, x - iso:x + iso + 1].any():
            continue
        flat = box.ravel()
        c = flat.size // 2
        if (flat[:c] >= v).any() or (flat[c + 1:] > v).any():
            continue
        out.append((x, y, v))
    out.sort(key=lambda t: (-t[2], t[1], t[0]))
    return out[:nmax], len(out)


def star_fwhm(img, x, y, half=10, maxit=25):
    img = np.asarray(img, dtype=np.float64)
    ny, nx = img.shape
    j0, j1 = max(y - half, 0), min(y + half + 1, ny)
    i0, i1 = max(x - half, 0), min(x + half + 1, nx)
    jj, ii = np.mgrid[j0:j1, i0:i1]
    I = img[j0:j1, i0:i1]
    cx, cy, s2 = float(x), float(y), 4.0
    out = np.nan
    for _ in range(maxit):
        w = np.exp(-0.5 * ((ii - cx) ** 2 + (jj - cy) ** 2) / s2) * I
        s0 = w.sum()
        if not s0 > 0:
            break
        cx, cy = (w * ii).sum() / s0, (w * jj).sum() / s0
        w = np.exp(-0.5 * ((ii - cx) ** 2 + (jj - cy) ** 2) / s2) * I
        t0 = w.sum()
        if not t0 > 0:
            break
        m2 = 0.5 * ((w * (ii - cx) ** 2).sum() + (w * (jj - cy) ** 2).sum()) / t0
        if not m2 > 0:
            break
        inv = 1.0 / m2 - 1.0 / s2
        if not inv > 0:
            break
        ns2 = 1.0 / inv
        done = abs(ns2 - s2) <= 1e-8 * ns2
        s2 = ns2
        out = 2.3548200450309493 * np.sqrt(s2)
        if done:
            break
    return out, cx, cy


def seeing(img, bad, lo, hi, iso=5, border=12, nmax=300, half=10):
    stars, _ = find_stars(img, bad, lo, hi, iso, border, nmax)
    f = np.array([star_fwhm(img, x, y, half)[0] for x, y, _ in stars])
    return float(np.nanmedian(f)) if np.isfinite(f).any() else np.nan


def negpix(img, x, y, med, sig, half=5):
    img = np.asarray(img, dtype=np.float32)
    ny, nx = img.shape
    s = (img - np.float32(med)) / np.float32(sig)
    out = np.zeros(len(x), dtype=np.int32)
    for k, (xs, ys) in enumerate(zip(x, y)):
        xc, yc = int(np.round(xs)) - 1, int(np.round(ys)) - 1
        hit = False
        for j in range(max(yc - half, 0), min(yc + half, ny - 1) + 1):
            for i in range(max(xc - half, 0), min(xc + half, nx - 1) + 1):
                if not s[j, i] < -5:
                    continue
                j0, j1 = max(j - 1, yc - half - 1, 0), min(j + 1, yc + half + 1, ny - 1)
                i0, i1 = max(i - 1, xc - half - 1, 0), min(i + 1, xc + half + 1, nx - 1)
                if (s[j0:j1 + 1, i0:i1 + 1] > 5).any():
                    hit = True
                    break
            if hit:
                break
        out[k] = int(hit)
    return out
