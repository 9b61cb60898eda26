#!/usr/bin/env python
"""Build science coadds for a list of jobs on the GPU engine.

    dostack.py jobs.csv            (one process)
    torchrun --nproc-per-node N dostack.py jobs.csv     (jobs dealt over N ranks)

Plays the part of the reference's ``scripts/dostack.py`` with its database
queries replaced by a job file: every row of ``jobs.csv`` has ``target`` (science
image paths joined by ';' -- each with its ``*mskimg.fits`` beside it) and the
bin edges ``left`` / ``right`` that go into the coadd's file name.  A job whose
output already exists is skipped, which is how an interrupted run resumes.
"""
import argparse
import os
import sys
import time

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zuds_amd as zuds  # noqa: E402


def load_inputs(target):
    """ScienceImages (mask attached) of one job."""
    images = []
    for path in (p.strip() for p in str(target).split(';')):
        sci = zuds.ScienceImage.from_file(path)
        sci.mask_image = zuds.MaskImage.from_file(path.replace('sciimg', 'mskimg'))
        images.append(sci)
    zuds.ensure_images_have_the_same_properties(images, zuds.GROUP_PROPERTIES)
    return images


def coadd_name(first, left, right):
    """<field>_<ccd>_<quadrant>_<filter>_<left>_<right>.coadd.fits next to the inputs."""
    stem = '_'.join([f'{int(first.field):06d}', f'c{int(first.ccdid):02d}',
                     f'q{int(first.qid)}', zuds.fid_map[int(first.fid)],
                     str(left), str(right)])
    return os.path.join(os.path.dirname(first.local_path), stem + '.coadd.fits')


def run_job(job, tmpdir):
    t0 = time.time()
    images = load_inputs(job['target'])
    outname = coadd_name(images[0], job['left'], job['right'])
    if os.path.exists(outname):
        return None
    t1 = time.time()
    print(f'load: {t1 - t0:.2f} sec to load input images for {outname}', flush=True)
    try:
        stack = zuds.ScienceCoadd.from_images(images, outfile_name=outname,
                                              data_product=False, tmpdir=tmpdir,
                                              nthreads=zuds.get_nthreads())
    except Exception as exc:        # one bad job must not end the night's run
        print(exc, [im.basename for im in images], flush=True)
        return None
    stack.binleft, stack.binright = job['left'], job['right']
    t2 = time.time()
    print(f'stack: {t2 - t1:.2f} sec to make {stack.basename}', flush=True)
    for frame in (*images, stack):
        frame.unmap()
    t3 = time.time()
    print(f'clean: took {t3 - t2:.2f} sec to clean up after {stack.basename}', flush=True)
    print(f'took {t3 - t0:.2f} sec to make "{stack.basename}"', flush=True)
    return stack.basename


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('jobs')
    ap.add_argument('--tmpdir', default='/tmp')
    args = ap.parse_args(argv)
    zuds.init_db()
    share = zuds.get_my_share_of_work(args.jobs, reader=pd.read_csv)
    table = share if isinstance(share, pd.DataFrame) else pd.DataFrame(list(share))
    for _, row in table.iterrows():
        run_job(row, args.tmpdir)
    return 0


if __name__ == '__main__':
    sys.exit(main())
