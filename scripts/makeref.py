#!/usr/bin/env python
"""Make the reference images: the driver of the reference's ``scripts/makeref.py``,
database-free.

usage: makeref.py dirs.txt min_date max_date version

``dirs.txt`` lists directories; each is searched for ``ztf*sciimg.fits`` (masks as
``*mskimg.fits`` beside them).  The selection is the reference's
(``scripts/makeref.py:57-78``), read from the header cards its ORM columns are filled from
(``zuds/image.py:481-491``): observation date (``OBSJD``) inside [min_date, max_date],
1.7 < ``SEEING`` < 2.5, 19.2 < ``MAGLIM`` < 22, ``INFOBITS`` == 0; the 50 deepest frames by
``MAGLIM``; at least 14 of them or the directory is skipped.  The coadd
``ref.<field>_c<ccd>_q<quadrant>_<filter>.<version>.fits`` (+ ``.weight.fits``, ``.mask.fits``)
is written into the directory by ``ReferenceImage.from_images`` - one ``zm_coadd`` call where
the reference ran SWarp twice.  A directory whose reference exists is skipped (resume by name).
The reference's catalog / archive / database steps that follow are out of scope here.
"""
import os
import sys
import time
from pathlib import Path

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zuds_amd as zuds  # noqa: E402

MIN_FRAMES, MAX_FRAMES = 14, 50


def obsdate(sci):
    """Observation date as a pandas Timestamp (the ORM's ``obsdate`` column) from OBSJD."""
    return pd.to_datetime(float(sci.header['OBSJD']), unit='D', origin='julian')


def select(directory, min_date, max_date):
    """The frames of ``directory`` that may go into a reference, deepest first."""
    ok = []
    for fn in sorted(Path(directory).glob('ztf*sciimg.fits')):
        # each file is read once to see that it is whole, then dropped again and re-mapped, as the
        # reference does (scripts/makeref.py:45-63): a directory of several hundred candidates must not
        # sit in host memory until the 50 deepest are chosen - from_images loads those when it runs
        try:
            sci = zuds.ScienceImage.from_file(f'{fn}')
            sci.load()
        except Exception:
            print(f'bad: File {fn.name} is corrupted, skipping...', flush=True)
            continue
        sci.clear()
        sci.map_to_local_file(f'{fn}')
        maskname = f'{fn}'.replace('sciimg', 'mskimg')
        try:
            sci.mask_image = zuds.MaskImage.from_file(maskname)
            sci.mask_image.load()
        except Exception:
            print(f'bad: File {os.path.basename(maskname)} is corrupted, skipping...', flush=True)
            continue
        sci.mask_image.clear()
        sci.mask_image.map_to_local_file(maskname)
        h = sci.header
        try:
            c1 = min_date <= obsdate(sci) <= max_date
            c2 = 1.7 < float(h['SEEING']) < 2.5
            c3 = 19.2 < float(h['MAGLIM']) < 22.
            c4 = int(h['INFOBITS']) == 0
        except KeyError as exc:
            print(f'bad: File {fn.name} has no {exc} card, skipping...', flush=True)
            continue
        if c1 and c2 and c3 and c4:
            ok.append(sci)
    return sorted(ok, key=lambda i: float(i.header['MAGLIM']), reverse=True)[:MAX_FRAMES]


def make_one(directory, min_date, max_date, version, tmpdir='./tmp'):
    t_start = time.time()
    top = select(directory, min_date, max_date)
    if len(top) == 0:
        print(f'Not enough images ({len(top)} < {MIN_FRAMES}) to make reference '
              f'for {directory}. Skipping...')
        return None
    first = top[0]
    coaddname = os.path.join(directory, f'ref.{int(first.field):06d}_c{int(first.ccdid):02d}'
                                        f'_q{int(first.qid)}_{zuds.fid_map[int(first.fid)]}.{version}.fits')
    if len(top) < MIN_FRAMES:
        print(f'Not enough images ({len(top)} < {MIN_FRAMES}) to make reference '
              f'{coaddname}. Skipping...')
        return None
    if os.path.exists(coaddname):
        print(f'{os.path.basename(coaddname)} exists, skipping', flush=True)
        return None
    try:
        coadd = zuds.ReferenceImage.from_images(top, coaddname, data_product=True,
                                                nthreads=zuds.get_nthreads(), tmpdir=tmpdir)
        coadd.version = version
    except TypeError as e:
        print(e, [t.basename for t in top], coaddname)
        return None
    t_stop = time.time()
    print(f'it took {t_stop - t_start} sec to make {coaddname}.', flush=True)
    return coadd


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 4:
        sys.exit(__doc__)
    infile, version = argv[0], argv[3]
    min_date, max_date = pd.to_datetime(argv[1]), pd.to_datetime(argv[2])
    zuds.init_db()
    made = []
    for d in zuds.get_my_share_of_work(infile):
        coadd = make_one(str(d), min_date, max_date, version)
        if coadd is not None:
            made.append(coadd.local_path)
    return made


if __name__ == '__main__':
    main()
