"""Device-resident pipelines on torch-allocated HBM buffers.

torch is plumbing here: it owns device memory, the HIP stream and (for N > 1)
the RCCL process group.  All arithmetic runs in libzudsmi through the ``*_dev``
entry points of the C-ABI.

Multi-GPU layout: frames are sharded across ranks; a
WEIGHTED coadd exchanges the two partial-sum planes ``S1 = sum(w v)`` and
``S0 = sum(w)`` with one RCCL all-reduce each; an exact CLIPPED / MEDIAN coadd
exchanges row bands of the resampled stacks with an all-to-all and combines its
own band; subtractions are independent per job.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, wcs_struct
from .engine import coadd_params, get_engine


def _torch():
    import torch
    return torch


class DeviceFrames(object):
    """A set of frames resident in HBM (torch tensors) plus their zm_dframe
    descriptors."""

    def __init__(self, frames, device):
        torch = _torch()
        self.n = len(frames)
        self.tensors = []
        self.arr = (_lib.zm_dframe * self.n)()
        for i, f in enumerate(frames):
            img = self._dev(f['img'], torch.float32, device)
            wgt = self._dev(f.get('wgt'), torch.float32, device)
            # an int16 mask (a ZTF mask as its BITPIX 16 file holds it) stays int16 in HBM: zm_dframe.mask_type
            m = f.get('mask')
            m16 = m is not None and (m.dtype == torch.int16 if isinstance(m, torch.Tensor) else np.asarray(m).dtype == np.int16)
            msk = self._dev(m, torch.int16 if m16 else torch.int32, device)
            self.tensors.append((img, wgt, msk))
            self.arr[i].img = img.data_ptr()
            self.arr[i].wgt = wgt.data_ptr() if wgt is not None else None
            self.arr[i].mask = msk.data_ptr() if msk is not None else None
            self.arr[i].mask_type = _lib.MASKTYPE_I16 if m16 else _lib.MASKTYPE_I32
            self.arr[i].wcs = wcs_struct(f['wcs'])
            self.arr[i].flxscale = float(f.get('flxscale', 1.0))

    @staticmethod
    def _dev(a, dtype, device):
        torch = _torch()
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            return a.to(device=device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


class DeviceCoadd(object):
    """Resample + combine on one GPU, optionally as one shard of a multi-GPU
    stack."""

    def __init__(self, wout, params=None, device=0, engine=None, want_mask=False, stream=None):
        torch = _torch()
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.engine = engine or get_engine(device)
        # a dedicated non-default stream: its handle is never 0, and torch ops
        # (RCCL collectives, copies) issued inside `with torch.cuda.stream(...)`
        # order against the kernels libzudsmi enqueues on the same stream
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        self.engine.set_stream(self.stream.cuda_stream)
        self.params = params or coadd_params()
        self.wout = wcs_struct(wout)
        onx, ony = self.wout.naxis[0], self.wout.naxis[1]
        self.shape = (ony, onx)
        # the two planes of one buffer: the multi-GPU reduce of the partial sums is one collective
        self._planes = torch.empty((2,) + self.shape, dtype=torch.float32, device=self.device)
        self.img, self.wgt = self._planes[0], self._planes[1]
        self.mask = self.mask_wgt = None
        if want_mask:
            self.mask = torch.empty(self.shape, dtype=torch.int32, device=self.device)
            self.mask_wgt = torch.empty(self.shape, dtype=torch.float32, device=self.device)

    def run(self, dframes, partial=False):
        """Enqueue resample + combine of ``dframes`` (a DeviceFrames)."""
        L = self.engine.L
        # several device-side objects may share one engine: each binds it to its own stream
        # before enqueueing, and waits for whatever the caller's current stream produced
        self.engine.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        with self.torch.cuda.stream(self.stream):
            check(L.zm_coadd_dev(self.engine.ctx, dframes.n, dframes.arr,
                                 C.byref(self.wout), C.byref(self.params),
                                 int(bool(partial)), self.img.data_ptr(),
                                 self.wgt.data_ptr(),
                                 self.mask.data_ptr() if self.mask is not None else None,
                                 self.mask_wgt.data_ptr() if self.mask_wgt is not None else None),
                  'zm_coadd_dev')
        return self.img, self.wgt

    def run_sharded_weighted(self, dframes, group=None):
        """This rank's frames -> partial sums -> RCCL all-reduce -> coadd.

        Every rank ends with the full coadd (img, wgt)."""
        import torch.distributed as dist
        if self.params.combine not in (_lib.COMBINE['WEIGHTED'], _lib.COMBINE['AVERAGE']):
            raise ValueError('a sum-reduce coadd needs COMBINE_TYPE WEIGHTED or AVERAGE; '
                             'use run_sharded_exact for CLIPPED / MEDIAN')
        from .parallel import reduce_masks
        L, ctx = self.engine.L, self.engine.ctx
        self.run(dframes, partial=True)
        import os
        from . import parallel
        if os.environ.get('ZM_NATIVE_RCCL') == '1':
            # the same reductions called from inside libzudsmi (csrc/comm.hip) on the engine's stream
            if getattr(self, '_native', None) is None:
                self._native = parallel.NativeComm(self.engine, group)
            self.engine.set_stream(self.stream.cuda_stream)
            self._native.all_reduce_planes(self.img, self.wgt)
            check(L.zm_coadd_finalize_dev(ctx, self.img.data_ptr(), self.wgt.data_ptr(), self.img.numel()),
                  'zm_coadd_finalize_dev')
            if self.mask is not None:
                self._native.reduce_mask(self.mask, self.params.mask_combine, self.mask_wgt)
            return self.img, self.wgt
        with self.torch.cuda.stream(self.stream):
            if dist.is_initialized() and (dist.get_world_size(group) > 1 or parallel.FORCE_COLLECTIVES):
                parallel.all_reduce_planes(self.img, self.wgt, group)
            check(L.zm_coadd_finalize_dev(ctx, self.img.data_ptr(), self.wgt.data_ptr(),
                                          self.img.numel()), 'zm_coadd_finalize_dev')
            if self.mask is not None:
                # the mask coadd of the whole stack: partial masks of all ranks, folded locally
                n, kind = self.mask.numel(), int(self.params.mask_combine)
                reduce_masks(
                    self.mask,
                    lambda acc, m, first: check(L.zm_mask_accum_dev(ctx, acc.data_ptr(), m.data_ptr(),
                                                                    acc.numel(), kind, int(first)),
                                                'zm_mask_accum_dev'),
                    lambda acc: check(L.zm_mask_finalize_dev(
                        ctx, acc.data_ptr(),
                        self.mask_wgt.data_ptr() if self.mask_wgt is not None else None, n),
                        'zm_mask_finalize_dev'),
                    group)
        return self.img, self.wgt


class DeviceSubtraction(object):
    """Device-resident single-epoch subtraction: the chain of
    ``Subtraction.from_images`` (``zuds/subtraction.py:57-226``) and
    ``prepare_hotpants`` (``zuds/hotpants.py:15-95``) without leaving HBM.

    ref_* live on the reference grid ``wref``; sci_* on the science grid ``wsci``.
    """

    def __init__(self, wsci, wref, device=0, engine=None, stream=None, overlap=False):
        from .constants import BAD_SUM, BIG_RMS, BKG_VAL
        torch = _torch()
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.engine = engine or get_engine(device)
        if stream is None:
            stream = torch.cuda.Stream(self.device)
            self.engine.set_stream(stream.cuda_stream)
        self.stream = stream
        self.wsci = wcs_struct(wsci)
        self.wref = wcs_struct(wref)
        nx, ny = self.wsci.naxis[0], self.wsci.naxis[1]
        self.shape = (ny, nx)
        self.n = nx * ny
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.ref_al = torch.empty(self.shape, **f32)      # reference on the science grid
        self.ref_al_w = torch.empty(self.shape, **f32)
        self.refmask_al = torch.empty(self.shape, **i32)
        self.refrms_al = torch.empty(self.shape, **f32)
        self.refrms_al_w = torch.empty(self.shape, **f32)
        self.submask = torch.empty(self.shape, **i32)
        self.bpm = torch.empty(self.shape, dtype=torch.uint8, device=self.device)
        self.scibkgsub = torch.empty(self.shape, **f32)
        self.diff = torch.empty(self.shape, **f32)
        self.noise = torch.empty(self.shape, **f32)
        self.BAD_SUM, self.BIG_RMS, self.BKG_VAL = BAD_SUM, float(BIG_RMS), float(BKG_VAL)
        self.info = _lib.zm_hp_info()
        # overlap (round 6): the mesh background of the science frame (a chain of small launches on ONE frame: ~0.13 ms of
        # latency) runs on a second context and stream beside the alignment of the reference (another ~0.13 ms) instead of
        # behind it - for a lone subtraction, whose latency is the point; the lanes of a pool keep one context per chain
        self.overlap = bool(overlap)
        self._bk_engine = self._bk_stream = None
        self._last_done = None              # event: the last subtraction of this chain has read its inputs
        self._pending = False               # run(wait=False): the fit summary has not been fetched yet

    def run(self, sci, sci_rms, sci_mask, sci_wgt, ref, ref_rms, ref_mask, seeing,
            nreg_side=3, subtract_back=True, hotpants_kws=None, ref_flxscale=1.0,
            ref_rms_flxscale=None, sci_ready=None, wait=True):
        """All arguments are torch tensors on this device; ``seeing`` is the
        science FWHM in pixels (header SEEING); ``ref_flxscale`` the FLXSCALE card of the
        reference (SWarp applies it on resampling, ``swarp.run_align``).
        Returns (diff, noise, submask).

        ``wait=False`` (round 6; ``zm_hp_params.async_info``): return when the fit's last rejection round has been
        seen - the convolution, bit 17 and the fit summary are enqueued on this chain's stream, not waited for, so the
        caller's next launches (the next coadd, the next frame's alignment) are enqueued while the convolution runs.
        The planes returned are then valid for work on this stream; ``result()`` waits for the summary, fills
        ``self.info`` and makes the checks ``run`` otherwise makes before it returns."""
        L, ctx = self.engine.L, self.engine.ctx
        ny, nx = self.shape
        scim, p = self.prepare(sci, sci_rms, sci_mask, sci_wgt, ref, ref_rms, ref_mask, seeing,
                               nreg_side=nreg_side, subtract_back=subtract_back, hotpants_kws=hotpants_kws,
                               ref_flxscale=ref_flxscale, ref_rms_flxscale=ref_rms_flxscale, sci_ready=sci_ready)
        if self._pending:
            self.result()                                 # (one summary buffer per context: the last one is read first)
        p.async_info = 0 if wait else 1
        with self.torch.cuda.stream(self.stream):
            check(L.zm_subtract_dev(ctx, scim.data_ptr(), sci_rms.data_ptr(),
                                    self.ref_al.data_ptr(), self.refrms_al.data_ptr(),
                                    self.bpm.data_ptr(), nx, ny, C.byref(p),
                                    self.diff.data_ptr(), self.noise.data_ptr(),
                                    C.byref(self.info)), 'zm_subtract_dev')
            if self.overlap:
                self._last_done = self.stream.record_event()
        self._pending = bool(self.info.status & _lib.HP_PENDING)
        if not self._pending:
            self.check_limits()
        return self.finish()

    def result(self):
        """Behind ``run(..., wait=False)``: wait for the convolution and the fit summary, fill ``self.info``, raise what
        ``run`` would have raised (a frame without a valid pixel).  Returns ``self.info``; a no-op when nothing is
        pending."""
        if self._pending:
            check(self.engine.L.zm_subtract_info(self.engine.ctx, C.byref(self.info)), 'zm_subtract_info')
            self._pending = False
            self.check_limits()
        return self.info

    def release_overlap(self):
        """Give the second context of ``overlap=True`` back (it is made again on the next run that wants it)."""
        if self._bk_engine is not None:
            self._bk_stream.synchronize()
            self._bk_engine.close()
        self._bk_engine = self._bk_stream = None

    def job(self, scim, sci_rms, p):
        """The ``zm_sub_job`` of a prepared chain (``zm_subtract_batch_dev``: many chains, one fit)."""
        return _lib.zm_sub_job(scim.data_ptr(), sci_rms.data_ptr(), self.ref_al.data_ptr(),
                               self.refrms_al.data_ptr(), self.bpm.data_ptr(), C.pointer(p),
                               self.diff.data_ptr(), self.noise.data_ptr())

    def finish(self):
        """Behind the hotpants step: nothing is left to enqueue - bit 17 where hotpants masked
        (subtraction.py:167-177) was set by the subtraction itself (``prepare``: flag_mask_dev)."""
        return self.diff, self.noise, self.submask

    def prepare(self, sci, sci_rms, sci_mask, sci_wgt, ref, ref_rms, ref_mask, seeing,
                nreg_side=3, subtract_back=True, hotpants_kws=None, ref_flxscale=1.0,
                ref_rms_flxscale=None, sci_ready=None):
        """Everything in front of the hotpants step, enqueued on this chain's stream: alignment of the
        reference and its rms map, bad-pixel map, mesh background, the two background estimates.
        Returns (science frame minus background, the job's ``zm_hp_params``)."""
        from .engine import hp_params
        from .hotpants import job_params
        L, ctx = self.engine.L, self.engine.ctx
        ny, nx = self.shape
        LAN = _lib.RESAMPLE['LANCZOS3']
        fs = C.c_double()
        check(L.zm_flux_scale(C.byref(self.wref), C.byref(self.wsci), float(ref_flxscale),
                              C.byref(fs)), 'zm_flux_scale')
        fs_rms = C.c_double()
        check(L.zm_flux_scale(C.byref(self.wref), C.byref(self.wsci),
                              float(ref_flxscale if ref_rms_flxscale is None else ref_rms_flxscale),
                              C.byref(fs_rms)), 'zm_flux_scale')
        self.engine.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        side = self.overlap and subtract_back
        bk_done = None
        if side:
            if self._bk_engine is None:
                from .engine import Engine
                self._bk_stream = self.torch.cuda.Stream(self.device)
                self._bk_engine = Engine(self.device.index, stream=self._bk_stream.cuda_stream)
            e2 = self._bk_engine
            # What the background waits for: the science planes, and the previous subtraction of this chain (it read
            # scibkgsub).  `sci_ready`: None - everything enqueued on this chain's stream so far (the planes may have
            # been made there); an event - that event; False - the planes are resident (bench.py: the background of the
            # science frame then runs beside whatever the chain's stream still has ahead of the subtraction, e.g. the
            # coadd of the reference it will be subtracted from - which it does not depend on).
            if sci_ready is None:
                self._bk_stream.wait_stream(self.stream)
            else:
                if sci_ready is not False:
                    self._bk_stream.wait_event(sci_ready)
                if self._last_done is not None:
                    self._bk_stream.wait_event(self._last_done)
            with self.torch.cuda.stream(self._bk_stream):
                check(L.zm_background_dev(e2.ctx, sci.data_ptr(), sci_wgt.data_ptr() if sci_wgt is not None else None,
                                          nx, ny, 128, 3, None, None, self.scibkgsub.data_ptr(), None), 'sci background')
                check(L.zm_add_scalar_dev(e2.ctx, self.scibkgsub.data_ptr(), self.BKG_VAL, self.n), 'pedestal')
                bk_done = self._bk_stream.record_event()
        with self.torch.cuda.stream(self.stream):
            # an int16 science mask (a ZTF mask as its file holds it) is widened here, once: the
            # bookkeeping kernels below read int32 words
            if sci_mask.dtype == self.torch.int16:
                if getattr(self, '_scimask32', None) is None:
                    self._scimask32 = self.torch.empty(self.shape, dtype=self.torch.int32, device=self.device)
                check(L.zm_mask_widen_dev(ctx, sci_mask.data_ptr(), self.n, self._scimask32.data_ptr()),
                      'zm_mask_widen_dev')
                sci_mask = self._scimask32
            # ref.aligned_to(sci): image (WEIGHT_TYPE NONE) + mask (OR), fitsfile.py:290-314.
            # The reference aligns a transaction copy whose mask is a plain MaskImageBase
            # (zuds/subtraction.py:94-99), so run_align does NOT add bit 16 to it
            # (zuds/swarp.py:186-191): uncovered pixels of the aligned mask stay 0, and
            # quick_background_estimate(ref) therefore counts them (zuds/hotpants.py:67).
            # Round 6: the reference and its rms map go to the science grid in ONE launch (zm_align_pair_dev: same
            # positions, same taps, the values of the two separate alignments bit for bit; hotpants.py:51 for the rms)
            pair = os.environ.get('ZM_ALIGN_PAIR', '1') != '0' and int(self.engine.query('edge')) == 0 and \
                int(self.engine.query('mask_resample')) == 0
            if pair:
                check(L.zm_align_pair_dev(ctx, ref.data_ptr(), ref_rms.data_ptr(), ref_mask.data_ptr(),
                                          C.byref(self.wref), C.byref(self.wsci), LAN, fs.value, fs_rms.value,
                                          self.ref_al.data_ptr(), self.refrms_al.data_ptr(),
                                          self.refmask_al.data_ptr()), 'align ref + rms')
            else:
                check(L.zm_resample_dev(ctx, ref.data_ptr(), None, ref_mask.data_ptr(),
                                        C.byref(self.wref), C.byref(self.wsci), LAN, fs.value,
                                        self.ref_al.data_ptr(), self.ref_al_w.data_ptr(),
                                        self.refmask_al.data_ptr()), 'align ref')
            # badpix = remapped_refmask | sci mask; boolean bpm (subtraction.py:135-142)
            check(L.zm_mask_bad_dev(ctx, self.refmask_al.data_ptr(), sci_mask.data_ptr(),
                                    self.BAD_SUM, self.n, self.submask.data_ptr(),
                                    self.bpm.data_ptr()), 'submask')
            # scimbkg = sci - mesh background + 150 (hotpants.py:27-32)
            if subtract_back and not side:
                check(L.zm_background_dev(ctx, sci.data_ptr(),
                                          sci_wgt.data_ptr() if sci_wgt is not None else None,
                                          nx, ny, 128, 3, None, None, self.scibkgsub.data_ptr(),
                                          None), 'sci background')
                check(L.zm_add_scalar_dev(ctx, self.scibkgsub.data_ptr(), self.BKG_VAL, self.n),
                      'pedestal')
                scim = self.scibkgsub
            elif subtract_back:
                self.stream.wait_event(bk_done)          # (made beside the alignment, above)
                scim = self.scibkgsub
            else:
                scim = sci
            # ref rms aligned to the science grid (hotpants.py:51)
            if not pair:
                check(L.zm_resample_dev(ctx, ref_rms.data_ptr(), None, None, C.byref(self.wref),
                                        C.byref(self.wsci), LAN, fs_rms.value, self.refrms_al.data_ptr(),
                                        self.refrms_al_w.data_ptr(), None), 'align ref rms')
            # quick_background_estimate x 2 (hotpants.py:65-67).  Round 4: the estimates stay on the device and
            # the subtraction takes its lower limits (median - 10 sigma, hotpants.py:69-72) from there - the fit is
            # enqueued behind them without the host reading them in between; `limits` fetches them afterwards.
            # ZM_HOST_LIMITS=1: the host round trip of rounds 1 - 3 (A / B, tests: the same products bit for bit).
            kws = dict(hotpants_kws or {})
            user_limits = 'il' in kws or 'tl' in kws
            if os.environ.get('ZM_HOST_LIMITS') == '1' or user_limits:
                mm = (C.c_double * 4)()
                check(L.zm_median_mad2_dev(ctx, scim.data_ptr(), sci_mask.data_ptr(),
                                           self.ref_al.data_ptr(), self.refmask_al.data_ptr(),
                                           self.n, mm), 'sci / ref bkg')
                m1, s1, m2, s2 = (float(v) for v in mm)
                self._limits = dict(il=m1 - 10 * s1, tl=m2 - 10 * s2)
                p = hp_params(**job_params(seeing, nx, ny, nreg_side, self._limits['il'],
                                           self._limits['tl'], hotpants_kws))
            else:
                if getattr(self, '_lim_dev', None) is None:
                    self._lim_dev = self.torch.zeros(6, dtype=self.torch.float64, device=self.device)
                check(L.zm_median_mad2_async_dev(ctx, scim.data_ptr(), sci_mask.data_ptr(),
                                                 self.ref_al.data_ptr(), self.refmask_al.data_ptr(),
                                                 self.n, self._lim_dev.data_ptr()), 'sci / ref bkg')
                # (the two sample counts travel to a pinned buffer behind the estimates, so that the call
                # that waits for the fit anyway can refuse a frame without a valid pixel as the host path does)
                if getattr(self, '_lim_host', None) is None:
                    self._lim_host = self.torch.zeros(6, dtype=self.torch.float64).pin_memory()
                self._lim_host.copy_(self._lim_dev, non_blocking=True)
                self._limits = None
                p = hp_params(**job_params(seeing, nx, ny, nreg_side, 0.0, 0.0, hotpants_kws))
                p.limits_dev = self._lim_dev.data_ptr()
                p.limits_nsigma = 10.0
        # bit 17 where hotpants masked (subtraction.py:167-177): enqueued by the subtraction itself, behind its
        # convolution and before it waits for the fit summary (zm_hp_params.flag_mask_dev)
        p.flag_mask_dev = self.submask.data_ptr()
        p.flag_bit = 1 << 17
        self._sci_rms = sci_rms                          # (kept alive until the fit has read it)
        return scim, p

    @property
    def limits(self):
        """``{'il', 'tl'}``: the lower data limits of the last run (``zuds/hotpants.py:65-72``).  When they
        were taken on the device this reads the estimates back (a synchronisation) and raises, as the host
        entry points do, when a frame had no valid pixel."""
        if getattr(self, '_limits', None) is None:
            self.stream.synchronize()
            self.check_limits()
            m1, s1, _, m2, s2, _ = (float(v) for v in self._lim_host)
            self._limits = dict(il=m1 - 10 * s1, tl=m2 - 10 * s2)
        return self._limits

    def check_limits(self):
        """Behind a wait for this chain's fit (``zm_subtract_dev`` / ``zm_subtract_batch_dev`` return with the fit
        summary, which the stream delivers after the estimates): raise, as the host entry point
        ``zm_median_mad2`` does, when the science frame or the aligned reference had no valid pixel - the fit of
        such a job ran on limits of 0 and its products are fill values."""
        if getattr(self, '_limits', None) is None and getattr(self, '_lim_host', None) is not None:
            if not (float(self._lim_host[2]) > 0 and float(self._lim_host[5]) > 0):
                raise _lib.ZMError('zm_median_mad2: every pixel is masked')


# ---------------------------------------------------------------------------
# FITS files <-> HBM without a host-side decode
_KIND = {'f32': 0, 'i32': 1, 'u8': 2, 'i16': 3}


class FITSDeviceIO(object):
    """Reads the data block of a primary HDU straight into pinned memory, sends it over PCIe
    as it lies on disk and decodes it (byte swap, BSCALE / BZERO) on the GPU; the reverse for
    products.  Stands in for the numpy decode inside ``FITSFile.load_data`` / ``save``
    (``zuds/fitsfile.py:69-94,146-206``) when the consumer is the device-resident pipeline.

    Two pinned staging buffers alternate, so the read of file i + 1 overlaps the copy and
    the decode of file i."""

    def __init__(self, device=0, engine=None, stream=None):
        torch = _torch()
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.engine = engine or get_engine(device)
        if stream is None:
            stream = torch.cuda.Stream(self.device)
            self.engine.set_stream(stream.cuda_stream)
        self.stream = stream
        self._pin = [None, None]
        self._busy = [None, None]          # event: the async copy out of the buffer is done
        self._turn = 0

    def _staging(self, nbytes):
        torch = self.torch
        k = self._turn
        self._turn ^= 1
        if self._busy[k] is not None:
            self._busy[k].synchronize()
        if self._pin[k] is None or self._pin[k].numel() < nbytes:
            self._pin[k] = torch.empty(int(nbytes * 1.1) + 4096, dtype=torch.uint8, pin_memory=True)
        return k, self._pin[k]

    def load(self, path, kind='f32'):
        """(tensor on the device, header dict).  kind: 'f32', 'i32', 'u8' or 'i16'; 'mask': 'i16' for a
        BITPIX 16 file without scaling (a ZTF mask), else 'i32'."""
        from . import fits
        torch = self.torch
        hdr, _, off = fits.read_header(path)
        nbytes = abs(int(hdr['BITPIX'])) // 8 * int(np.prod([int(hdr[f'NAXIS{i}'])
                                                             for i in range(1, int(hdr['NAXIS']) + 1)]))
        k, pin = self._staging(nbytes)
        raw, hdr, _, info = fits.read_raw(path, out=pin.numpy())
        if kind == 'mask':
            kind = 'i16' if (info['bitpix'] == 16 and info['bscale'] == 1.0 and info['bzero'] == 0.0) else 'i32'
        dt = {'f32': torch.float32, 'i32': torch.int32, 'u8': torch.uint8, 'i16': torch.int16}[kind]
        out = torch.empty(info['shape'], dtype=dt, device=self.device)
        self.engine.set_stream(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            d_raw = pin[:info['nbytes']].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._busy[k] = ev
            check(self.engine.L.zm_fits_decode_dev(self.engine.ctx, d_raw.data_ptr(), info['bitpix'],
                                                   info['bscale'], info['bzero'], info['count'],
                                                   _KIND[kind], out.data_ptr()), 'zm_fits_decode_dev')
            d_raw.record_stream(self.stream)
        return out, hdr

    def save(self, path, tensor, header=None, comments=None, bitpix=None):
        """Encode on the device, copy back, write.  float32 -> BITPIX -32, int32 -> 32 (or 16
        with ``bitpix=16``), uint8 -> 8.  Blocks until the file is written."""
        from . import fits
        torch = self.torch
        t = tensor.contiguous()
        if t.dtype == torch.float32:
            kind, bp = 0, -32
        elif t.dtype == torch.int32:
            kind, bp = (3, 16) if bitpix == 16 else (1, 32)
        elif t.dtype in (torch.uint8, torch.bool):
            t = t.to(torch.uint8)
            kind, bp = 2, 8
        else:
            raise ValueError(f'cannot write {t.dtype} to FITS from the device')
        nbytes = t.numel() * abs(bp) // 8
        k, pin = self._staging(nbytes)
        self.engine.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            d_raw = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(self.engine.L.zm_fits_encode_dev(self.engine.ctx, t.data_ptr(), kind, t.numel(),
                                                   d_raw.data_ptr()), 'zm_fits_encode_dev')
            pin[:nbytes].copy_(d_raw, non_blocking=True)
        self.stream.synchronize()
        fits.write_raw(path, pin[:nbytes].numpy(), tuple(t.shape), bp, header, comments)

    @property
    def ring(self):
        """The pipelined form of this object (``fitsring.FITSRing``: reader / writer threads, copy and return
        streams of its own), created on first use."""
        if getattr(self, '_ring', None) is None:
            from .fitsring import FITSRing
            self._ring = FITSRing(self.device.index)
        return self._ring

    def save_async(self, path, tensor, header=None, comments=None, bitpix=None):
        """``save`` without the wait: the plane is encoded behind the kernels already on this object's stream and may
        be overwritten as soon as the call returns; copy-back and write happen on the ring's stream and threads.
        Returns a Future of the path; ``flush()`` waits for all of them.  Same bytes as ``save``."""
        self.engine.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        return self.ring.save(path, tensor, header, comments, bitpix, engine=self.engine, stream=self.stream)

    def flush(self):
        if getattr(self, '_ring', None) is not None:
            self._ring.flush()

    def close(self):
        if getattr(self, '_ring', None) is not None:
            self._ring.close()
            self._ring = None

    def load_many(self, wanted, nreaders=4):
        """[(path, kind), ...] -> [(tensor, header), ...] in order.  The files are read by `nreaders`
        threads into a pool of pinned buffers (reads from the page cache scale with threads: one reader moves
        ~11 GB/s, four ~40) while this thread enqueues copy + decode of whatever has arrived, in order."""
        import queue
        from concurrent.futures import ThreadPoolExecutor
        from . import fits
        torch = self.torch
        if getattr(self, '_pool', None) is None:
            self._pool = ThreadPoolExecutor(nreaders)
            self._free = queue.Queue()

        def pin_of(nbytes):
            try:
                while True:
                    buf, ev = self._free.get_nowait()
                    if ev is not None:
                        ev.synchronize()
                    if buf.numel() >= nbytes:
                        return buf
            except queue.Empty:
                pass
            return torch.empty(int(nbytes) + 4096, dtype=torch.uint8, pin_memory=True)
        from collections import deque
        todo, jobs = deque(wanted), deque()

        def submit():
            path, kind = todo.popleft()
            hdr, _, _ = fits.read_header(path)
            nbytes = abs(int(hdr['BITPIX'])) // 8 * int(np.prod([int(hdr[f'NAXIS{i}'])
                                                                 for i in range(1, int(hdr['NAXIS']) + 1)]))
            pin = pin_of(nbytes)
            jobs.append((self._pool.submit(fits.read_raw, path, pin.numpy()), pin, kind))
        out = []
        self.engine.set_stream(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            while todo or jobs:
                while todo and len(jobs) < 2 * nreaders:        # a bounded window of reads in flight (pinned memory)
                    submit()
                fut, pin, kind = jobs.popleft()
                raw, hdr, _, info = fut.result()
                if kind == 'mask':
                    kind = 'i16' if (info['bitpix'] == 16 and info['bscale'] == 1.0 and info['bzero'] == 0.0) else 'i32'
                dt = {'f32': torch.float32, 'i32': torch.int32, 'u8': torch.uint8, 'i16': torch.int16}[kind]
                t = torch.empty(info['shape'], dtype=dt, device=self.device)
                d_raw = pin[:info['nbytes']].to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self._free.put((pin, ev))
                check(self.engine.L.zm_fits_decode_dev(self.engine.ctx, d_raw.data_ptr(), info['bitpix'],
                                                       info['bscale'], info['bzero'], info['count'],
                                                       _KIND[kind], t.data_ptr()), 'zm_fits_decode_dev')
                d_raw.record_stream(self.stream)
                out.append((t, hdr))
        return out

    def load_frames(self, sci_paths, weight_paths=None, mask_paths=None, zp_key='MAGZP'):
        """A DeviceFrames for ``zm_coadd_dev`` from science / weight / mask files
        (``FLXSCALE = 10^(-0.4 (MAGZP - 25))``, ``zuds/swarp.py:31``)."""
        from .wcs import WCS
        wanted, slots = [], []
        for i, sp in enumerate(sci_paths):
            wanted.append((sp, 'f32'))
            slots.append((i, 'img'))
            if weight_paths is not None and weight_paths[i] is not None:
                wanted.append((weight_paths[i], 'f32'))
                slots.append((i, 'wgt'))
            if mask_paths is not None and mask_paths[i] is not None:
                wanted.append((mask_paths[i], 'mask'))
                slots.append((i, 'mask'))
        frames = [dict() for _ in sci_paths]
        for (i, key), (t, hdr) in zip(slots, self.load_many(wanted)):
            frames[i][key] = t
            if key == 'img':
                frames[i].update(wcs=WCS.from_header(hdr), header=hdr,
                                 flxscale=10 ** (-0.4 * (float(hdr.get(zp_key, 25.0)) - 25.0)))
        # consumers on other streams order against the loads through the current stream
        torch = self.torch
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return DeviceFrames(frames, self.device), frames
