"""Many subtractions on one GPU at a time (BASELINE config 5: 64 epochs x 4 quadrants,
32 subtractions per GPU, forced photometry on every difference image).

The reference runs one process per subtraction job, 64 per node
(``nersc/controller.py:101``; per job ``scripts/donightly.py:21-40`` ->
``scripts/dosub.py:do_one`` -> ``SingleEpochSubtraction.from_images``, then
``scripts/dophot.py:94-156`` -> ``raw_aperture_photometry``).  One device-resident
subtraction is latency-bound - 13 launches of the kernel-fit solver with 23 serial block
steps each - and leaves most of the GPU idle, so this module runs J of them concurrently:
J engines (contexts, each with its own stream and scratch) driven by J host threads; the
ctypes calls release the GIL.  ``Engine.set_share(J)`` makes every engine size the solver's
resident grid to 1 / J of the CUs.  Results do not depend on J (tests/test_nightly_gpu.py).

``SubtractionPool(J, batch=B)`` (round 4) is the other shape of the same work: the hardware runs at most six
kernels of J streams side by side, so beyond six chains a job's latency sets the rate.  With ``batch`` the kernel
fits of B jobs are ONE chain of launches with the job as a grid dimension (``zm_subtract_batch_dev``: 9 regions x B
jobs of the one-workgroup factorisation fill 9 B compute units for the time one factorisation takes); what fills the
GPU by itself - alignment, backgrounds, masks, stamp search, the convolution - is enqueued job after job around
it.  J is then the number of such lanes (threads, engines, streams) working on different batches at the same time,
so that one lane's throughput kernels run beside another lane's fit.  Same products, bit for bit.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from ._lib import check
from .constants import APERTURE_RADIUS
from .engine import Engine

__all__ = ['SubtractionPool', 'SubtractionJob']

# Streams share the HIP runtime's pool of hardware queues (4 by default): with more streams than
# queues, kernels of different streams queue up behind each other instead of running side by side
# (measured: two workers never overlapped).  The runtime reads this at the first HIP call of the
# process, so import this module before anything touches the GPU.  8, not more, and only for the
# one-process-per-GPU pattern of this module: with several processes on a card the hardware queues
# are oversubscribed and time-sliced (two processes with 16 each: 6 x slower).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


class SubtractionJob(object):
    """One science frame against one reference, all planes as torch tensors on the device.

    sci / ref: dicts with ``img``, ``rms``, ``mask`` (int32), ``wcs``; sci also ``wgt`` (for the
    mesh background, may be None) and ``seeing`` (FWHM in pixels); ref optionally ``flxscale``.
    ``radec``: optional (ra, dec) arrays for forced photometry on the difference image."""

    def __init__(self, sci, ref, radec=None, nreg_side=3, hotpants_kws=None, tag=None):
        self.sci, self.ref, self.radec = sci, ref, radec
        self.nreg_side, self.hotpants_kws, self.tag = nreg_side, hotpants_kws, tag


class _Worker(object):
    """An engine, its stream and the device planes of one subtraction chain."""

    def __init__(self, device, share):
        import torch
        self.torch = torch
        self.stream = torch.cuda.Stream(torch.device('cuda', device))
        self.engine = Engine(device, stream=self.stream.cuda_stream)
        self.engine.set_share(share)
        self.device = device
        self.chain = None
        self.key = None

    def release(self):
        """Drop the device planes (``SubtractionPool.close``)."""
        self.chain, self.key = None, None

    def subtract(self, job, keep=True):
        from .device import DeviceSubtraction
        torch = self.torch
        sci, ref = job.sci, job.ref
        # The device planes are reused between jobs of one frame size; the WCS structs are
        # rebuilt for EVERY job (a few hundred bytes of host work).  They were once cached under
        # id(wcs): ids are only unique among live objects, and a driver that builds fresh WCS
        # objects per batch gets the same addresses back for different frames.
        shape = tuple(sci['img'].shape)
        if self.chain is not None and self.key != shape:
            self.chain = None                           # frees the planes of another frame size first
        if self.chain is None:
            self.chain = DeviceSubtraction(sci['wcs'], ref['wcs'], device=self.device,
                                           engine=self.engine, stream=self.stream)
            self.key = shape
        else:
            self.chain.wsci = _lib.wcs_struct(sci['wcs'])
            self.chain.wref = _lib.wcs_struct(ref['wcs'])
        ch = self.chain
        # (round 6: the call returns when the fit's last round has been seen; the photometry's host work - sky to
        # pixel positions, its buffers - and its launches are made while the convolution runs, the fit summary is
        # read behind the synchronisation)
        diff, noise, mask = ch.run(sci['img'], sci['rms'], sci['mask'], sci.get('wgt'), ref['img'],
                                   ref['rms'], ref['mask'], seeing=float(sci['seeing']),
                                   nreg_side=job.nreg_side, hotpants_kws=job.hotpants_kws,
                                   ref_flxscale=float(ref.get('flxscale', 1.0)), wait=False)
        out = _collect(self, ch, job, None, diff, noise, mask, keep)
        self.stream.synchronize()
        info = ch.result()
        out['info'] = {k: getattr(info, k) for k, _ in info._fields_}
        return _settle(out)


def _collect(w, ch, job, info, diff, noise, mask, keep):
    """What a finished chain hands back: the fit summary, the forced photometry on the planes that are
    still in HBM (scripts/dophot.py:131-133) and, with ``keep``, copies of the three products.  Enqueued on
    the worker's stream; the caller synchronises."""
    torch, sci = w.torch, job.sci
    out = dict(tag=job.tag, info=None if info is None else {k: getattr(info, k) for k, _ in info._fields_})
    if job.radec is not None:
        ra, dec = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in job.radec)
        x, y = sci['wcs'].all_world2pix(ra, dec, 0)
        n = x.size
        with torch.cuda.stream(w.stream):
            pos = torch.from_numpy(np.ascontiguousarray(np.stack([x, y]))).to(ch.device, non_blocking=True)
            res = torch.empty((2, n), dtype=torch.float64, device=ch.device)
            flg = torch.empty(n, dtype=torch.int32, device=ch.device)
            ny_, nx_ = ch.shape
            check(w.engine.L.zm_aperture_photometry_dev(
                w.engine.ctx, diff.data_ptr(), noise.data_ptr(), mask.data_ptr(), nx_, ny_, n,
                pos[0].data_ptr(), pos[1].data_ptr(), float(APERTURE_RADIUS), res[0].data_ptr(),
                res[1].data_ptr(), flg.data_ptr()), 'zm_aperture_photometry_dev')
            # (copies into pinned memory, not waited for here: a lane enqueues the next job's work first;
            # _settle turns them into arrays once the stream has been synchronised)
            res_h = torch.empty((2, n), dtype=torch.float64, pin_memory=True)
            flg_h = torch.empty(n, dtype=torch.int32, pin_memory=True)
            res_h.copy_(res, non_blocking=True)
            flg_h.copy_(flg, non_blocking=True)
        out['phot'] = dict(x=x, y=y, _pending=(res_h, flg_h, res, flg, pos))
    if keep:
        with torch.cuda.stream(w.stream):
            out['diff'], out['noise'], out['mask'] = diff.clone(), noise.clone(), mask.clone()
    return out


def _settle(out):
    """After the stream synchronisation: the photometry of ``_collect`` as numpy arrays."""
    ph = out.get('phot')
    if ph and '_pending' in ph:
        res_h, flg_h = ph.pop('_pending')[:2]
        ph.update(flux=res_h[0].numpy().copy(), fluxerr=res_h[1].numpy().copy(), flags=flg_h.numpy().copy())
    return out


def _fit_key(job):
    """Jobs whose kernel fits have the same shape (frame size, half widths, basis, orders, regions, stamps,
    thresholds) can share a batch; data limits and fill values may differ (zm_subtract_batch_dev)."""
    from .hotpants import job_params
    ny, nx = (int(v) for v in job.sci['img'].shape)
    p = job_params(float(job.sci['seeing']), nx, ny, job.nreg_side, 0.0, 0.0, job.hotpants_kws)
    for k in ('tu', 'tl', 'iu', 'il', 'fin', 'fi'):
        p.pop(k, None)
    p['r'], p['rss'] = int(p['r']), int(p['rss'])
    return (ny, nx) + tuple(sorted((k, tuple(v) if isinstance(v, list) else v) for k, v in p.items()))


def _batches(keys, batch):
    """Job indices -> batches: jobs of one fit key, in job order, at most ``batch`` of them; a group is cut into the
    fewest batches that hold it, of equal size up to one job (32 jobs at batch 14 are 11 + 11 + 10, not 14 + 14 + 4:
    the factorisations of a batch take the time of one whatever its size, so the largest batch sets the pace)."""
    groups = {}
    for i, key in enumerate(keys):
        groups.setdefault(key, []).append(i)
    chunks = []
    for idx in groups.values():
        nb = -(-len(idx) // batch)
        q, r = divmod(len(idx), nb)
        k = 0
        for b in range(nb):
            n = q + (1 if b < r else 0)
            chunks.append(idx[k:k + n])
            k += n
    return chunks


class _BatchLane(object):
    """An engine, its stream and up to ``batch`` subtraction chains whose kernel fits run as one batch."""

    def __init__(self, device, batch, turn=None, share=1):
        import torch
        self.torch = torch
        self.stream = torch.cuda.Stream(torch.device('cuda', device))
        self.engine = Engine(device, stream=self.stream.cuda_stream)
        #  a chunk of ONE job - a fit-key group of size 1, the per-job repeat after a failed batch -
        # goes through zm_subtract_dev: with other lanes at work it must take the one-workgroup-per-region
        # factorisation like the workers of an unbatched pool, not the form that wants the GPU to itself
        if share >= 2:
            self.engine.set_share(share)
        self.device, self.batch = device, batch
        self.chains, self.key = [], None
        # `turn` (a lock shared by the lanes of a pool): one lane at a time runs its preparation - kernels that
        # fill the GPU alone - and holds the lock until the GPU has done it, so that the lanes fall out of step
        # and one lane's preparation meets the other's fit (whose factorisations leave 40 % of the CUs idle)
        # instead of its preparation
        self.turn = turn

    def release(self):
        """Drop the device planes (``SubtractionPool.close``)."""
        self.chains, self.key = [], None

    def subtract(self, jobs, keep=True):
        """``jobs``: at most ``batch`` jobs of one fit key.  Returns their results in order."""
        import ctypes as C
        from .device import DeviceSubtraction
        shape = tuple(jobs[0].sci['img'].shape)
        if self.key != shape:
            self.chains, self.key = [], shape            # frees the planes of another frame size first
        while len(self.chains) < len(jobs):
            self.chains.append(DeviceSubtraction(jobs[0].sci['wcs'], jobs[0].ref['wcs'], device=self.device,
                                                 engine=self.engine, stream=self.stream))
        n = len(jobs)
        arr = (_lib.zm_sub_job * n)()
        infos = (_lib.zm_hp_info * n)()
        held = []
        if self.turn is not None:
            self.turn.acquire()
        try:
            for k, job in enumerate(jobs):
                ch, sci, ref = self.chains[k], job.sci, job.ref
                ch.wsci = _lib.wcs_struct(sci['wcs'])
                ch.wref = _lib.wcs_struct(ref['wcs'])
                scim, p = ch.prepare(sci['img'], sci['rms'], sci['mask'], sci.get('wgt'), ref['img'], ref['rms'],
                                     ref['mask'], seeing=float(sci['seeing']), nreg_side=job.nreg_side,
                                     hotpants_kws=job.hotpants_kws, ref_flxscale=float(ref.get('flxscale', 1.0)))
                arr[k] = ch.job(scim, sci['rms'], p)
                held.append((scim, p))
            if self.turn is not None:
                self.stream.synchronize()
        finally:
            if self.turn is not None:
                self.turn.release()
        ny, nx = shape
        self.engine.set_stream(self.stream.cuda_stream)
        check(self.engine.L.zm_subtract_batch_dev(self.engine.ctx, n, arr, nx, ny, infos), 'zm_subtract_batch_dev')
        outs = []
        for k, job in enumerate(jobs):
            ch = self.chains[k]
            C.memmove(C.byref(ch.info), C.byref(infos[k]), C.sizeof(_lib.zm_hp_info))
            try:
                ch.check_limits()              # (a frame without a valid pixel fails its job, not the batch)
            except _lib.ZMError as exc:
                outs.append(dict(tag=job.tag, error=str(exc)))
                continue
            diff, noise, mask = ch.finish()
            outs.append(_collect(self, ch, job, infos[k], diff, noise, mask, keep))
        self.stream.synchronize()
        return [_settle(o) for o in outs]


class SubtractionPool(object):
    """``njobs`` subtractions in flight on one GPU.  ``batch`` >= 2: ``njobs`` lanes of ``batch`` chains each whose
    kernel fits run as one batch (module docstring); ``batch`` = 0 (default): lanes and batch chosen for ``njobs`` jobs
    in flight; ``batch`` = 1: ``njobs`` separate chains, one host thread each (rounds 3 - 5)."""

    def __init__(self, njobs=8, device=0, batch=0):
        if not 1 <= njobs <= 64:
            raise ValueError('njobs must be in 1 .. 64')
        if not 0 <= batch <= 64:
            raise ValueError('batch must be in 0 .. 64')
        self.njobs, self.device = int(njobs), int(device)
        self.batch = int(batch) if int(batch) >= 2 else 0
        if int(batch) == 0 and self.njobs >= 2:
            # Round 6: `njobs` subtractions in flight without a word about batches are run in the shape that was
            # measured fastest for that many (bench.py --nightly-batches, 32 subtractions of 3072^2, ms per
            # subtraction): separate chains flip every job to the one-workgroup-per-region factorisation (1.1 ms
            # instead of 0.22) and were SLOWER than one worker at njobs = 2 (4.85 against 3.8; BENCH_r05), and so
            # is a batch of two or three (1x2 5.8, 1x3 4.7: a batch pays that factorisation once, but for too few).
            # Up to five in flight: ONE worker on the latency form, job after job (3.7 - 3.85; 2x2 measured 3.4 on one
            # box and 4.2 on another); from six on: two lanes of njobs / 2 batched fits (2x3 2.9, 2x4 2.5, 2x6 2.3);
            # from twelve on three lanes (3x4 2.1).
            # Same products bit for bit whatever the shape (tests/test_nightly_gpu.py).  `batch=1` asks for the
            # separate chains explicitly (A / B, tests).
            if self.njobs <= 5:
                self.njobs = 1
            else:
                lanes = 2 if self.njobs < 12 else 3
                self.batch = -(-self.njobs // lanes)
                self.njobs = lanes
        # share >= 2 selects the one-workgroup-per-region form of the kernel fit's factorisation, which
        # claims nothing for itself (a lone job keeps the many-workgroup form); ZM_POOL_SHARE overrides
        # (developer: with ZM_CHOL_FORM=lat it is the fraction of the CUs each job's resident grid gets)
        self.share = min(max(int(os.environ.get('ZM_POOL_SHARE', self.njobs)), self.njobs), 64)
        _lib.lib()                                   # loaded once, here, not by racing worker threads
        self._local = threading.local()
        self._workers = []
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=self.njobs)
        self._turn = threading.Lock() if (self.batch and self.njobs > 1 and
                                          os.environ.get('ZM_POOL_TURNS', '1') != '0') else None

    def _worker(self):
        w = getattr(self._local, 'w', None)
        if w is None:
            import torch
            torch.cuda.set_device(self.device)
            w = self._local.w = (_BatchLane(self.device, self.batch, self._turn, self.share if self.njobs > 1 else 1) if self.batch
                                 else _Worker(self.device, self.share))
            with self._lock:
                self._workers.append(w)
        return w

    def _run(self, job, keep):
        # one failed job must not end the batch (the reference's drivers wrap every image in
        # try / except, scripts/dosub.py:205-213): the result carries the error instead of products
        try:
            return self._worker().subtract(job, keep=keep)
        except _lib.ZMError as exc:
            return dict(tag=job.tag, error=str(exc))

    def _run_batch(self, jobs, keep):
        try:
            return self._worker().subtract(jobs, keep=keep)
        except _lib.ZMError:
            # a batch that fails as a whole is taken apart: every job on its own, failures per job
            outs = []
            for job in jobs:
                try:
                    outs.extend(self._worker().subtract([job], keep=keep))
                except _lib.ZMError as exc:
                    outs.append(dict(tag=job.tag, error=str(exc)))
            return outs

    def map(self, jobs, keep=True, sync=True):
        """Run every job; results in job order.  ``keep``: clone diff / noise / mask of each job
        out of the worker's planes (off for throughput runs that only want the photometry).  ``sync=False``: the
        caller has already waited for the streams that produced the inputs (scripts/donightly.py: a device-wide
        wait here would also wait for the NEXT batch's copies on the ring's stream)."""
        import torch
        if sync:
            torch.cuda.synchronize(self.device)           # inputs produced on other streams are complete
        jobs = list(jobs)
        if not self.batch:
            return list(self._pool.map(lambda j: self._run(j, keep), jobs))
        chunks = _batches([_fit_key(job) for job in jobs], self.batch)
        results = [None] * len(jobs)
        for idx, outs in zip(chunks, self._pool.map(lambda c: self._run_batch([jobs[i] for i in c], keep), chunks)):
            for i, o in zip(idx, outs):
                results[i] = o
        return results

    def close(self):
        self._pool.shutdown(wait=True)
        for w in self._workers:
            w.release()
            w.engine.close()
        self._workers = []
