#!/usr/bin/env python
"""Make subtractions: the driver of the reference's ``scripts/dosub.py``
(``do_one``), database-free.

usage: dosub.py images.txt ref.fits
images.txt lists science image paths (masks as ``*mskimg.fits``; a ``.weight.fits``
or ``.rms.fits`` sibling is used when present, else the mesh background RMS map).
``ref.fits`` needs ``ref.mask.fits`` and ``ref.weight.fits`` next to it.
"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zuds_amd as zuds

zuds.init_db()


class PredecessorError(Exception):
    pass


def do_one(fn, sciclass, subclass, refname, tmpdir='/tmp'):
    tstart = time.time()
    sstart = time.time()
    sci = sciclass.from_file(fn)
    maskname = fn.replace('sciimg', 'mskimg') if 'sciimg' in fn else fn.replace('.fits', '.mask.fits')
    sci.mask_image = zuds.MaskImage.from_file(maskname)
    weightname = fn.replace('.fits', '.weight.fits')
    rmsname = fn.replace('.fits', '.rms.fits')
    if os.path.exists(weightname):
        sci._weightimg = zuds.FITSImage.from_file(weightname)
    elif os.path.exists(rmsname):
        sci._rmsimg = zuds.FITSImage.from_file(rmsname)
    else:
        if sciclass == zuds.ScienceImage:
            _ = sci.rms_image       # mesh BACKGROUND_RMS map (dosub.py:42-44)
        else:
            raise RuntimeError(f'Cannot produce a subtraction for {fn},'
                               f' the image has no weightmap or rms map.')
    sstop = time.time()
    print(f'sci: {sstop - sstart:.2f} sec to load  {sci.basename}', flush=True)

    if not os.path.exists(refname):
        raise RuntimeError(f'Ref {refname} does not exist. Skipping...')
    rstart = time.time()
    ref = zuds.ReferenceImage.from_file(refname, load_others=False)
    ref.mask_image = zuds.MaskImage.from_file(refname.replace('.fits', '.mask.fits'))
    ref._weightimg = zuds.FITSImage.from_file(refname.replace('.fits', '.weight.fits'))
    rstop = time.time()
    print(f'ref: {rstop - rstart:.2f} sec to load ref for {sci.basename}', flush=True)

    outname = zuds.sub_name(sci.local_path, ref.local_path)
    if os.path.exists(outname):     # checkpoint by name (dosub.py:85-94)
        raise PredecessorError(f'{os.path.basename(outname)} already has a predecessor')

    substart = time.time()
    sub = subclass.from_images(sci, ref, data_product=False, tmpdir=tmpdir, refined=True)
    substop = time.time()
    print(f'sub: {substop - substart:.2f} sec to make {sub.basename}', flush=True)

    cleanstart = time.time()
    sci.unmap()
    cleanstop = time.time()
    tstop = time.time()
    print(f'clean: took {cleanstop - cleanstart} sec to clean up after {sub.basename}"', flush=True)
    print(f'took {tstop - tstart} sec to make "{sub.basename}"', flush=True)
    return sub


if __name__ == '__main__':
    infile = sys.argv[1]
    refname = sys.argv[2]
    subclass = zuds.SingleEpochSubtraction
    sciclass = zuds.ScienceImage
    imgs = zuds.get_my_share_of_work(infile)
    for fn in imgs:
        try:
            sub = do_one(str(fn), sciclass, subclass, refname)
        except Exception:
            traceback.print_exception(*sys.exc_info())
            continue
