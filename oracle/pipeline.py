"""End-to-end restatement of the two product constructors (oracle; test
infrastructure): ``_coadd_from_images`` (``zuds/coadd.py:25-236``) and
``Subtraction.from_images`` + ``prepare_hotpants`` (``zuds/subtraction.py:57-226``,
``zuds/hotpants.py:15-95``), on arrays instead of FITS files."""
import numpy as np

from . import background as oback
from . import combine as ocombine
from . import grid as ogrid
from . import hotpants as ohp
from . import resample as ores

BKG_VAL = 150.0                       # zuds/constants.py:23
BIG_RMS = np.sqrt(50000.0)            # zuds/constants.py:3
BAD_SUM = 198589                      # zuds/constants.py:45-46


def rescaled_weight(img, wgt, mesh=128):
    """RESCALE_WEIGHTS Y: scale the variance map so that its level equals the
    measured background variance."""
    bkg, rms, bmean, bsig, _, _ = oback.background(img, wgt, mesh)
    with np.errstate(divide='ignore'):
        var = np.where(wgt > 1e-30, 1.0 / np.where(wgt > 0, wgt, 1), 0.0)
    vb, vs = oback.mesh_maps(var, wgt, mesh)
    vbf, _ = oback.filter_maps(vb, vs, 3)
    level = oback.fqmedian(vbf.ravel())
    scale = bsig * bsig / level if (level > 0 and bsig > 0) else 1.0
    return bkg, wgt / scale


def coadd_from_images(frames, addbkg=True, combine='CLIPPED', mesh=128):
    """frames: dicts img, wgt, mask (int), wcs (oracle WCS), magzp."""
    wout = ogrid.autogrid([f['wcs'] for f in frames])
    onx, ony = wout.naxis
    vals, wgts, masks, cov = [], [], [], []
    for f in frames:
        img = f['img'].astype(np.float64)
        wgt = f['wgt'].astype(np.float64)
        bkg, wgt = rescaled_weight(img, wgt, mesh)
        px, py = ores.positions(wout, f['wcs'], onx, ony)
        fs = ores.flux_scale(f['wcs'], wout, 10 ** (-0.4 * (f['magzp'] - 25.0)))
        o, w, m = ores.resample(img - bkg, wgt, px, py, ores.LANCZOS3, fs, f['mask'])
        vals.append(o)
        wgts.append(w)
        masks.append(m)
        nx, ny = f['wcs'].naxis
        cov.append(ores.coverage(px, py, nx, ny))
    img, wgt, _ = ocombine.combine(np.array(vals), np.array(wgts), combine)
    msk, mcov = ocombine.combine_masks(np.array(masks), np.array(cov), 'AND')
    msk = msk + np.where(mcov == 0, 2 ** 16, 0)         # zuds/mask.py:26-33
    if addbkg:
        img = img + BKG_VAL
    return dict(img=img, wgt=wgt, mask=msk, wcs=wout)


def align(img, win, wout, mask=None, flxscale=1.0):
    onx, ony = wout.naxis
    px, py = ores.positions(wout, win, onx, ony)
    fs = ores.flux_scale(win, wout, flxscale)
    return ores.resample(img, None, px, py, ores.LANCZOS3, fs, mask)


def subtract_from_images(sci, ref, seeing, nreg_side=3, subtract_back=True,
                         hotpants_kws=None):
    """sci: dict img, rms, wgt (for the background), mask, wcs; ref: dict img, wgt,
    mask, wcs (a coadd).  Returns dict diff, noise, mask."""
    hotpants_kws = dict(hotpants_kws or {})
    ws, wr = sci['wcs'], ref['wcs']
    nx, ny = ws.naxis
    # ref.aligned_to(sci): image without weights, mask with OR and no bit 16
    ref_al, _, refmask_al = align(ref['img'].astype(np.float64), wr, ws,
                                  ref['mask'].astype(np.int64))
    badpix = refmask_al | sci['mask'].astype(np.int64)
    bpm = (badpix & BAD_SUM) > 0
    simg = sci['img'].astype(np.float64)
    if subtract_back:
        bkg, _, _, _, _, _ = oback.background(simg, sci['wgt'].astype(np.float64), 128)
        scim = (simg - bkg).astype(np.float32) + np.float32(BKG_VAL)
    else:
        scim = simg.astype(np.float32)
    # reference rms map: 1 / sqrt(w), BIG_RMS where bad (zuds/image.py:173-208)
    rbad = ((ref['mask'].astype(np.int64) & BAD_SUM) > 0)
    with np.errstate(divide='ignore'):
        rrms = np.where(rbad | ~(ref['wgt'] > 0), BIG_RMS, 1.0 / np.sqrt(np.where(ref['wgt'] > 0, ref['wgt'], 1)))
    rrms_al, _, _ = align(rrms.astype(np.float32).astype(np.float64), wr, ws)
    s = scim[sci['mask'] == 0]
    scibkg = np.median(s)
    scistd = 1.4826 * np.median(np.abs(s - scibkg))
    r32 = ref_al.astype(np.float32)
    r = r32[refmask_al == 0]
    refbkg = np.median(r)
    refstd = 1.4826 * np.median(np.abs(r - refbkg))
    kw = dict(tu=5e3, iu=5e3, tl=float(refbkg) - 10 * float(refstd),
              il=float(scibkg) - 10 * float(scistd), r=2.5 * seeing, rss=6.0 * seeing,
              fin=BIG_RMS, nsx=max(int(nx / 100.0 / nreg_side), 1),
              nsy=max(int(ny / 100.0 / nreg_side), 1), nrx=nreg_side, nry=nreg_side,
              bgo=0, ko=4)
    kw.update(hotpants_kws)
    diff, noise, info = ohp.subtract(scim, r32, sci['rms'].astype(np.float32),
                                     rrms_al.astype(np.float32), bpm.astype(np.uint8), **kw)
    mask = badpix | np.where(diff.astype(np.float32) == np.float32(1e-30), 2 ** 17, 0)
    return dict(diff=diff, noise=noise, mask=mask, info=info, ref_al=ref_al, scim=scim)
