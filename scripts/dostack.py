#!/usr/bin/env python
"""Make science coadds: the driver loop of the reference's ``scripts/dostack.py``
with the database lookups replaced by paths in the job file.

usage: dostack.py jobs.csv
jobs.csv columns: ``target`` (';'-separated science image paths, masks next to them
as ``*mskimg.fits``), ``left``, ``right`` (bin edges used in the output name).
Launch under ``torchrun`` to shard the job list over ranks / GPUs
(``zuds.get_my_share_of_work``, reference: MPI scatter in ``zuds/mpi.py:36-64``).
"""
import os
import sys
import time

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zuds_amd as zuds

zuds.init_db()

infile = sys.argv[1]
jobs = zuds.get_my_share_of_work(infile, reader=pd.read_csv)
if not isinstance(jobs, pd.DataFrame):
    jobs = pd.DataFrame(list(jobs))

for _, job in jobs.iterrows():
    tstart = time.time()
    sstart = time.time()
    images = []
    for path in str(job['target']).split(';'):
        image = zuds.ScienceImage.from_file(path.strip())
        image.mask_image = zuds.MaskImage.from_file(path.strip().replace('sciimg', 'mskimg'))
        images.append(image)
    zuds.ensure_images_have_the_same_properties(images, zuds.GROUP_PROPERTIES)

    field = f'{int(images[0].field):06d}'
    ccdid = f'c{int(images[0].ccdid):02d}'
    qid = f'q{int(images[0].qid)}'
    fid = f'{zuds.fid_map[int(images[0].fid)]}'
    basename = f'{field}_{ccdid}_{qid}_{fid}_{job["left"]}_{job["right"]}.coadd.fits'
    outname = os.path.join(os.path.dirname(images[0].local_path), basename)
    sstop = time.time()
    if os.path.exists(outname):      # checkpoint by name (dostack.py:44-49)
        continue
    print(f'load: {sstop - sstart:.2f} sec to load input images for {outname}', flush=True)

    stackstart = time.time()
    try:
        stack = zuds.ScienceCoadd.from_images(images, outfile_name=outname, data_product=False,
                                              tmpdir='/tmp', nthreads=zuds.get_nthreads())
    except Exception as e:
        print(e, [i.basename for i in images], flush=True)
        continue
    stack.binleft = job['left']
    stack.binright = job['right']
    stackstop = time.time()
    print(f'stack: {stackstop - stackstart:.2f} sec to make {stack.basename}', flush=True)

    cleanstart = time.time()
    for sci in images + [stack]:
        sci.unmap()
    cleanstop = time.time()
    tstop = time.time()
    print(f'clean: took {cleanstop - cleanstart} sec to clean up after {stack.basename}"',
          flush=True)
    print(f'took {tstop - tstart} sec to make "{stack.basename}"', flush=True)
