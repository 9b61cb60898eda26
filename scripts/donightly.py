#!/usr/bin/env python
"""Nightly subtractions + forced photometry: the job of the reference's
``scripts/donightly.py`` (per image: ``dosub.do_one``) followed by ``scripts/dophot.py``
(``raw_aperture_photometry`` at known sky positions), database-free, with the kernel fits of B
subtractions as one batch of launches on each of J lanes (``nightly.SubtractionPool(J, batch=B)``;
``--fit-batch 0``: J separate subtractions in flight).

usage: donightly.py images.txt ref.fits [positions.txt] [--jobs J] [--fit-batch B] [--batch FRAMES] [--nreg-side N]

* ``images.txt``: science image paths (``*sciimg.fits``; the mask is ``*mskimg.fits``; a
  ``.weight.fits`` sibling is required: 1 / rms^2, 0 on bad pixels).  The list is sharded over
  ranks as ``zuds.get_my_share_of_work`` does (RANK / WORLD_SIZE, one rank per GPU).
* ``ref.fits`` with ``ref.mask.fits`` and ``ref.weight.fits`` next to it.
* ``positions.txt``: ``ra dec`` per line (degrees); forced r = 3 px apertures are measured on
  every difference image and written to ``<sub>.phot.txt``
  (columns of the reference's photometry table: ra dec flux fluxerr flags zp obsjd).

Products per image, with the reference's names: ``sub.<sci>_<ref>.fits``, ``.rms.fits``,
``.mask.fits`` next to the science image.  An image whose subtraction exists is skipped
(the reference's checkpoint by name, ``scripts/dosub.py:85-94``)."""
import argparse
import importlib
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zuds_amd as zuds

zuds.init_db()


def load_reference(io, refname):
    """The reference planes in HBM: image, rms (1 / sqrt(w), BIG_RMS where bad: zuds/image.py:173-208)
    and mask."""
    import torch
    img, hdr = io.load(refname, 'f32')
    wgt, _ = io.load(refname.replace('.fits', '.weight.fits'), 'f32')
    mask, _ = io.load(refname.replace('.fits', '.mask.fits'), 'i32')
    eng = io.engine
    rms = torch.empty_like(img)
    bad = torch.empty(mask.shape, dtype=torch.uint8, device=mask.device)
    with torch.cuda.stream(io.stream):
        zuds._lib.check(eng.L.zm_mask_bad_dev(eng.ctx, mask.data_ptr(), None, zuds.BAD_SUM, mask.numel(),
                                              None, bad.data_ptr()))
        zuds._lib.check(eng.L.zm_rms_from_weight_dev(eng.ctx, wgt.data_ptr(), bad.data_ptr(), wgt.numel(),
                                                     float(zuds.BIG_RMS), rms.data_ptr()))
    io.stream.synchronize()
    return dict(img=img, rms=rms, mask=mask, wcs=zuds.WCS.from_header(hdr),
                flxscale=float(hdr.get('FLXSCALE', 1.0)), header=hdr, path=refname)


def science_files(fn):
    """The three files of a science frame, as the ring / the device loader take them."""
    return [(fn, 'f32'), (fn.replace('sciimg', 'mskimg'), 'i32'), (fn.replace('.fits', '.weight.fits'), 'f32')]


def finish_science(io, fn, img, hdr, mask, wgt):
    """Behind the loads, on ``io.stream`` (not waited for): rms = 1 / sqrt(w), BIG_RMS where the mask is bad or the
    pixel saturated (zuds/image.py:173-208)."""
    import torch
    if 'SEEING' not in hdr:
        raise RuntimeError(f'{fn}: no SEEING card (run estimate_seeing on the frame first)')
    eng = io.engine
    eng.set_stream(io.stream.cuda_stream)
    with torch.cuda.stream(io.stream):
        rms = torch.empty_like(img)
        bad = torch.empty(mask.shape, dtype=torch.uint8, device=mask.device)
        zuds._lib.check(eng.L.zm_mask_bad_dev(eng.ctx, mask.data_ptr(), None, zuds.BAD_SUM, mask.numel(),
                                              None, bad.data_ptr()))
        zuds._lib.check(eng.L.zm_rms_from_weight_dev(eng.ctx, wgt.data_ptr(), bad.data_ptr(), wgt.numel(),
                                                     float(zuds.BIG_RMS), rms.data_ptr()))
        if 'SATURATE' in hdr:         # zuds/image.py:203-204
            rms = torch.where(img >= 0.9 * float(hdr['SATURATE']), torch.full_like(rms, float(zuds.BIG_RMS)), rms)
    return dict(img=img, rms=rms, mask=mask, wgt=wgt, wcs=zuds.WCS.from_header(hdr),
                seeing=float(hdr['SEEING']), header=hdr, path=fn)


def load_science(io, fn):
    """One frame, serially (the ring of ``run_night`` loads a batch ahead instead)."""
    (img, hdr), (mask, _), (wgt, _) = (io.load(p, k) for p, k in science_files(fn))
    sci = finish_science(io, fn, img, hdr, mask, wgt)
    io.stream.synchronize()
    return sci


def write_products(io, sci, ref, res):
    """``io``: anything with ``save(path, tensor, header)`` - the ring (files written behind the caller's back,
    ``flush`` waits for them) or a ``FITSDeviceIO`` (each file written before the call returns)."""
    out = zuds.sub_name(sci['path'], ref['path'])
    hdr = dict(sci['header'])
    hdr.update(zuds.hotpants.info_cards(res['info']))    # KSUM00, NSTAMPS, ZMSTATUS, ZMUNSOLV, ZMRETRY
    io.save(out, res['diff'], hdr)
    io.save(out.replace('.fits', '.rms.fits'), res['noise'], hdr)
    mh = dict(sci['header'])
    mh['BIT17'] = 17
    io.save(out.replace('.fits', '.mask.fits'), res['mask'], mh)
    if 'phot' in res:
        p = res['phot']
        zp = float(hdr.get('MAGZP', 0.0)) + float(hdr.get(zuds.APER_KEY, 0.0))
        jd = float(hdr.get('OBSJD', 0.0))
        ra, dec = sci['radec']
        # (np.savetxt's bytes - '# ' + header, one '%'-formatted row per line - without its per-row overhead: 1.6 -> 0.3 ms
        # per table, which at 2 ms per subtraction is host time that matters)
        fmt = '%.8f %.8f %.6e %.6e %d %.5f %.6f\n'
        rows = zip(ra.tolist(), dec.tolist(), np.asarray(p['flux'], dtype=np.float64).tolist(),
                   np.asarray(p['fluxerr'], dtype=np.float64).tolist(), np.asarray(p['flags']).tolist(),
                   [zp] * ra.size, [jd] * ra.size)
        with open(out.replace('.fits', '.phot.txt'), 'w') as fh:
            fh.write('# ra dec flux fluxerr flags zp obsjd\n' + ''.join([fmt % r for r in rows]))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('infile')
    ap.add_argument('refname')
    ap.add_argument('positions', nargs='?')
    ap.add_argument('--jobs', type=int, default=3,
                    help='lanes (host thread, context, stream) working on the GPU at the same time; with --fit-batch 0: '
                         'subtractions in flight')
    ap.add_argument('--nreg-side', type=int, default=3)
    ap.add_argument('--batch', type=int, default=36, help='science frames resident at a time')
    ap.add_argument('--fit-batch', type=int, default=12,
                    help='kernel fits per launch chain (SubtractionPool(jobs, batch=N): --jobs lanes whose N fits run '
                         'as one batch, zm_subtract_batch_dev; measured on 32 subtractions of 3072^2: three lanes of '
                         '11 = 2.0 - 2.1 ms each whether the frames share one seeing or fall into three seeing '
                         'groups, two lanes of 16 1.95 / 2.8, 8 - 16 separate chains 3.0); 0: one chain per job')
    args = ap.parse_args(argv)

    nightly = importlib.import_module('zuds-pipeline_amd.nightly')
    device = importlib.import_module('zuds-pipeline_amd.device')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    imgs = [str(f) for f in zuds.get_my_share_of_work(args.infile)]
    radec = None
    if args.positions:
        t = np.atleast_2d(np.loadtxt(args.positions))
        radec = (t[:, 0].copy(), t[:, 1].copy())

    io = device.FITSDeviceIO(local, engine=zuds.Engine(local))
    ref = load_reference(io, args.refname)
    pool = nightly.SubtractionPool(args.jobs, device=local, batch=args.fit_batch)
    ring = importlib.import_module('zuds-pipeline_amd.fitsring').FITSRing(local)
    try:
        return run_night(imgs, ref, pool, io, ring, radec, batch=args.batch, nreg_side=args.nreg_side)
    finally:
        pool.close()
        ring.close()


def run_night(imgs, ref, pool, io, ring, radec=None, batch=36, nreg_side=3):
    """The images of this rank against one reference.  The files of batch b + 1 are read, sent and decoded by the
    ring (fitsring.FITSRing: reader threads, copy stream) while the pool subtracts batch b; the products of batch b
    are encoded on the device, copied back on a third stream and written by the ring's writer threads while
    batch b + 1 runs.  Returns the paths of the difference images, in order."""
    nightly = importlib.import_module('zuds-pipeline_amd.nightly')
    refname = ref['path']
    chunks = [imgs[b0:b0 + batch] for b0 in range(0, len(imgs), batch)]

    def ask(b):
        todo = []
        for fn in chunks[b]:
            if os.path.exists(zuds.sub_name(fn, refname)):
                print(f'{os.path.basename(fn)}: subtraction exists, skipping', flush=True)
            else:
                todo.append(fn)
        wanted = [w for fn in todo for w in science_files(fn)]
        return todo, (ring.prefetch(wanted, [k == 'f32' and i % 3 == 0 for i, (_, k) in enumerate(wanted)],
                                    return_exceptions=True) if wanted else None)
    from concurrent.futures import ThreadPoolExecutor

    def finish(scis, results):
        # headers, photometry tables and the hand-over of the planes to the ring: host work of ~3 ms per image, done
        # on a thread of its own while the pool is at the next batch
        out = []
        for sci, res in zip(scis, results):
            if 'error' in res:
                # the job raised: no products, the night goes on
                print(f'{os.path.basename(sci["path"])}: subtraction failed: {res["error"]}', flush=True)
                continue
            if res['info']['status'] != 0:
                # some regions of the fit have no solution: their pixels carry the fill value and
                # bit 17; the products are written with ZMSTATUS / ZMUNSOLV in their headers
                print(f'{os.path.basename(sci["path"])}: {res["info"]["nunsolved"]} region(s) of the '
                      f'kernel fit unsolved (status {res["info"]["status"]}, '
                      f'{res["info"]["nstamps_used"]} stamps)', flush=True)
            out.append(write_products(ring, sci, ref, res))
        return out
    def prepare(todo, ticket):
        # the decoded planes of a batch -> its jobs (rms maps enqueued on io.stream and waited for): host work of
        # ~0.8 ms per frame, done for batch b + 1 on a thread of its own while the pool is at batch b
        t0 = time.time()
        loaded = ticket.result(io.stream) if ticket is not None else []
        t1 = time.time()
        scis, jobs = [], []
        for k, fn in enumerate(todo):
            trio = loaded[3 * k:3 * k + 3]
            try:
                for item in trio:
                    if isinstance(item, BaseException):
                        raise item
                (img, hdr), (mask, _), (wgt, _) = trio
                sci = finish_science(io, fn, img, hdr, mask, wgt)
            except Exception:
                # (the reference's drivers: try / except per image, scripts/dosub.py:205-213)
                traceback.print_exception(*sys.exc_info())
                continue
            sci['radec'] = radec
            scis.append(sci)
            jobs.append(nightly.SubtractionJob(sci, ref, radec=radec, nreg_side=nreg_side, tag=fn))
        io.stream.synchronize()                  # (the rms maps; the pool's lanes read them on their own streams)
        return scis, jobs, (1e3 * (t1 - t0), 1e3 * (time.time() - t1))
    finisher = ThreadPoolExecutor(1, thread_name_prefix='zmnight-fin')
    prep = ThreadPoolExecutor(1, thread_name_prefix='zmnight-prep')
    pending = []
    try:
        # two batches ahead in files (the ring reads b + 2 while b + 1 is prepared and b subtracted), one in jobs
        asked = [ask(b) for b in range(min(2, len(chunks)))]
        ready = prep.submit(prepare, *asked[0]) if chunks else None
        for b in range(len(chunks)):
            t0 = time.time()
            scis, jobs, (ms_files, ms_prep) = ready.result()
            if b + 2 < len(chunks):
                asked.append(ask(b + 2))
            ready = prep.submit(prepare, *asked[b + 1]) if b + 1 < len(chunks) else None
            t3 = time.time()
            results = pool.map(jobs, sync=False)
            pending.append(finisher.submit(finish, scis, results))
            if jobs:
                print(f'took {time.time() - t0:.2f} sec to make {len(jobs)} subtractions', flush=True)
            if os.environ.get('ZM_NIGHT_TRACE'):
                print(f'  batch {b}: waited {1e3 * (t3 - t0):.1f} ms for its jobs (files {ms_files:.1f}, rms maps {ms_prep:.1f} on the '
                      f'preparing thread), pool {1e3 * (time.time() - t3):.1f}', flush=True)
        t0 = time.time()
        done = [out for f in pending for out in f.result()]
        t1 = time.time()
    finally:
        prep.shutdown(wait=True)
        finisher.shutdown(wait=True)
    ring.flush()                                 # every product is on disk when this returns
    if os.environ.get('ZM_NIGHT_TRACE'):
        print(f'  finisher {1e3 * (t1 - t0):.1f} ms behind the last batch, files on disk {1e3 * (time.time() - t1):.1f} ms '
              f'later', flush=True)
    return done


if __name__ == '__main__':
    main()
