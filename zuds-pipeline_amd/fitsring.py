"""FITS files <-> HBM as a pipeline: what a night looks like on the file-inclusive clock.

The reference moves every plane through ``FITSFile.load_data`` / ``save`` on the one thread of its
process (``zuds/fitsfile.py:69-206``; per coadd ``zuds/coadd.py:61-93,165-217``: copies in, products out).
``device.FITSDeviceIO`` already keeps the decode off the host (raw data block -> pinned memory -> PCIe ->
``zm_fits_decode_dev``), but one call at a time: read, compute, write, each waiting for the other.  This
module overlaps the three:

* ``prefetch(wanted)`` returns at once.  ``nreaders`` threads read the data blocks of the NEXT step's files
  into a ring of pinned buffers (``readinto`` releases the GIL; reads out of the page cache scale with
  threads), a feeder thread sends each block over PCIe on the ring's own high-priority copy stream as soon
  as it has arrived and decodes it on a second one (an engine of the ring's own: a ``zm_ctx`` is bound to one stream),
  all while the caller's stream computes the CURRENT step.  ``Ticket.result(stream)`` orders the consumer's
  stream behind the last decode with an event - the host waits only for the files, never for the GPU.
* ``save(path, tensor, ...)`` returns at once as well: the product is encoded into a device buffer of its
  own on the producer's stream (so the producer may overwrite the plane right away), copied back on a third
  stream (PCIe is full duplex) into a pinned buffer, and written by one of ``nwriters`` threads when the
  copy's event has fired.  The bytes are those of ``FITSDeviceIO.save`` / ``fits.write``
  (tests/test_fitsio_gpu.py).  ``flush()`` waits for every file handed over so far and re-raises the
  first error of a reader or writer.

Nothing here computes pixels: torch owns the pinned and device buffers and the streams, libzudsmi the two
kernels.
"""
import os
import queue
import threading
from collections import deque
from concurrent.futures import Future, ThreadPoolExecutor

import numpy as np

from . import _lib, fits
from ._lib import check
from .engine import Engine

__all__ = ['FITSRing', 'Ticket']

_KIND = {'f32': 0, 'i32': 1, 'u8': 2, 'i16': 3}
_GEOM_KEYS = ('SIMPLE', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'NAXIS3', 'BSCALE', 'BZERO')


def _default_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 4)
    return max(2, n)


def scan_header(f, path, full):
    """(header or geometry dict, comments, data offset) from an open file.  ``full=False`` parses only the cards
    that describe the array (weight maps and masks: nobody reads the rest of their headers; parsing them all
    costs more host time per step than the step's kernels take)."""
    if full:
        return fits._read_header(f.read, path)
    geom, nblocks = {}, 0
    while True:
        block = f.read(fits.BLOCK)
        if len(block) < fits.BLOCK:
            raise ValueError(f'{path}: truncated FITS header')
        nblocks += 1
        end = False
        for i in range(0, fits.BLOCK, 80):
            key = block[i:i + 8]
            if key == b'END     ':
                end = True
                break
            k = key.rstrip().decode('ascii', 'replace')
            if k in _GEOM_KEYS and block[i + 8:i + 10] == b'= ':
                geom[k] = fits._parse_value(block[i + 10:i + 80].decode('ascii', 'replace'))[0]
        if end:
            break
    if geom.get('SIMPLE') is not True:
        raise ValueError(f'{path}: not a standard FITS file (SIMPLE != T)')
    return geom, {}, nblocks * fits.BLOCK


def _geometry(path, header):
    naxis = int(header.get('NAXIS', 0))
    if naxis == 0:
        raise ValueError(f'{path}: no image data in the primary HDU')
    shape = tuple(int(header[f'NAXIS{i}']) for i in range(naxis, 0, -1))
    bitpix = int(header['BITPIX'])
    if bitpix not in fits._BITPIX_DTYPE:
        raise ValueError(f'{path}: unsupported BITPIX {bitpix}')
    count = int(np.prod(shape))
    return dict(bitpix=bitpix, shape=shape, count=count, nbytes=count * abs(bitpix) // 8,
                bscale=float(header.get('BSCALE', 1)), bzero=float(header.get('BZERO', 0)))


class _PinPool(object):
    """Pinned staging buffers of one size class, handed out when the copy that last read (or wrote) them is
    done.  The pool grows up to ``limit`` bytes; beyond it ``get`` blocks until a buffer comes back."""

    def __init__(self, torch, limit):
        self.torch, self.limit = torch, int(limit)
        self.total = 0
        self.free = deque()              # (buffer, event or None)
        self.cv = threading.Condition()

    def get(self, nbytes):
        torch = self.torch
        with self.cv:
            while True:
                pick = next((i for i, (b, _) in enumerate(self.free) if b.numel() >= nbytes), None)
                if pick is not None:
                    buf, ev = self.free[pick]
                    del self.free[pick]
                    break
                if self.total == 0 or self.total + nbytes <= self.limit:
                    self.total += nbytes + 4096
                    buf, ev = None, None
                    break
                if self.free:                     # at the limit with buffers that are all too small: trade one in
                    small, _ = self.free.popleft()
                    self.total -= small.numel()
                    continue
                self.cv.wait()
        if buf is None:
            return torch.empty(int(nbytes) + 4096, dtype=torch.uint8, pin_memory=True)
        if ev is not None:
            ev.synchronize()
        return buf

    def put(self, buf, ev=None):
        with self.cv:
            self.free.append((buf, ev))
            self.cv.notify()


class Ticket(object):
    """The files of one ``prefetch``: ``result(stream)`` -> [(tensor, header), ...] in the order asked for."""

    def __init__(self, n):
        self._fut = Future()
        self.n = n
        self.arrived = None              # event on the ring's decode stream: the last decode

    def result(self, stream=None, timeout=None):
        out = self._fut.result(timeout)
        if stream is not None:
            stream.wait_event(self.arrived)
            for t, _ in out:
                t.record_stream(stream)          # (allocated on the decode stream, consumed on this one)
        return out

    def done(self):
        return self._fut.done()


class FITSRing(object):

    def __init__(self, device=0, nreaders=None, nwriters=None, pinned_in=3 << 29, pinned_out=1 << 31):
        import torch
        self.torch = torch
        self.device = torch.device('cuda', device)
        nthr = _default_threads()
        self.nreaders = int(nreaders or min(12, max(4, nthr - 4)))
        self.nwriters = int(nwriters or min(6, max(2, nthr // 3)))      # (5 - 8 writers keep up with the D2H copies; more of
        #  them contend in the page cache: 48 new 38 MB files at 42 GB/s with 8 threads, 20 GB/s with 32 - tools/ring_probe.py)
        # high-priority streams get hardware queues of their own: a copy never waits behind a kernel of the step it
        # is meant to overlap with (bench.py, data_movement_clocks)
        self.cs = torch.cuda.Stream(self.device, priority=-1)          # H2D, nothing else: copies back to back
        self.xs = torch.cuda.Stream(self.device, priority=-1)          # decode, behind each copy's event
        self.ds = torch.cuda.Stream(self.device, priority=-1)          # D2H of encoded products
        self.ceng = Engine(device, stream=self.xs.cuda_stream)      # (contexts on the ring's streams: none of their own)
        self.deng = Engine(device, stream=self.ds.cuda_stream)
        self._readers = ThreadPoolExecutor(self.nreaders, thread_name_prefix='zmfits-r')
        self._writers = ThreadPoolExecutor(self.nwriters, thread_name_prefix='zmfits-w')
        self._pin_in = _PinPool(torch, pinned_in)
        self._pin_out = _PinPool(torch, pinned_out)
        self._tickets = queue.Queue()
        self._feeder = threading.Thread(target=self._feed, name='zmfits-feed', daemon=True)
        self._feeder.start()
        self._pending = deque()          # futures of files being written
        self._lock = threading.Lock()
        self.stats = dict(files_in=0, bytes_in=0, files_out=0, bytes_out=0)

    # -- in ---------------------------------------------------------------------------------
    def prefetch(self, wanted, full_header=None, return_exceptions=False):
        """``wanted``: [(path, kind), ...] with kind 'f32', 'i32', 'u8', 'i16' or 'mask' ('i16' for a BITPIX 16
        file without scaling - a ZTF mask as its file holds it - else 'i32').  ``full_header``: per file, whether
        the whole header is parsed (default: only for 'f32' planes; the others return the array's own cards).
        ``return_exceptions``: a file that cannot be read puts its exception in its place of the result instead of
        failing the ticket (a night goes on without the frame, ``scripts/dosub.py:205-213``)."""
        wanted = list(wanted)
        if full_header is None:
            full_header = [k == 'f32' for _, k in wanted]
        t = Ticket(len(wanted))
        self._tickets.put((t, wanted, list(full_header), bool(return_exceptions)))
        return t

    def _read(self, path, full):
        with open(path, 'rb', buffering=0) as f:
            hdr, _, off = scan_header(f, path, full)
            info = _geometry(path, hdr)
            pin = self._pin_in.get(info['nbytes'])
            try:
                buf = memoryview(pin.numpy())[:info['nbytes']]
                f.seek(off)
                got = 0
                while got < info['nbytes']:
                    k = f.readinto(buf[got:])
                    if not k:
                        raise ValueError(f'{path}: truncated FITS data ({got} of {info["nbytes"]} bytes)')
                    got += k
            except BaseException:
                self._pin_in.put(pin)
                raise
        return pin, hdr, info

    def _feed(self):
        torch = self.torch
        torch.cuda.set_device(self.device)
        while True:
            item = self._tickets.get()
            if item is None:
                return
            ticket, wanted, full, tolerant = item
            futs = []
            try:
                futs = [self._readers.submit(self._read, p, fl) for (p, _), fl in zip(wanted, full)]
                out = []
                for (path, kind), fut in zip(wanted, futs):
                    try:
                        pin, hdr, info = fut.result()
                    except Exception as e:        # noqa
                        if not tolerant:
                            raise
                        out.append(e)
                        continue
                    if kind == 'mask':
                        kind = 'i16' if (info['bitpix'] == 16 and info['bscale'] == 1.0 and info['bzero'] == 0.0) else 'i32'
                    dt = {'f32': torch.float32, 'i32': torch.int32, 'u8': torch.uint8, 'i16': torch.int16}[kind]
                    with torch.cuda.stream(self.cs):
                        d_raw = pin[:info['nbytes']].to(self.device, non_blocking=True)
                        ev = self.cs.record_event()
                    self._pin_in.put(pin, ev)
                    # (the decode on a stream of its own: on the copy stream each kernel would hold up the next copy -
                    # 99 copy -> kernel -> copy hand-overs per step, measured 50 GB/s against 56 for the copies alone)
                    with torch.cuda.stream(self.xs):
                        self.xs.wait_event(ev)
                        t = torch.empty(info['shape'], dtype=dt, device=self.device)
                        check(self.ceng.L.zm_fits_decode_dev(self.ceng.ctx, d_raw.data_ptr(), info['bitpix'],
                                                             info['bscale'], info['bzero'], info['count'],
                                                             _KIND[kind], t.data_ptr()), 'zm_fits_decode_dev')
                        d_raw.record_stream(self.xs)
                    del d_raw
                    out.append((t, hdr))
                    self.stats['files_in'] += 1
                    self.stats['bytes_in'] += info['nbytes']
                ticket.arrived = self.xs.record_event()
                ticket._fut.set_result(out)
            except BaseException as e:            # noqa: handed to whoever asks for the result
                for fut in futs:
                    if not fut.cancel():
                        try:
                            self._pin_in.put(fut.result()[0])
                        except BaseException:     # noqa
                            pass
                ticket._fut.set_exception(e)

    def prefetch_frames(self, sci_paths, weight_paths=None, mask_paths=None, extra=()):
        """The files of a stack (``device.FITSDeviceIO.load_frames``) + ``extra`` [(path, kind), ...] as one
        ticket; ``frames(ticket, stream)`` turns it into a ``DeviceFrames``."""
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
        full = [k == 'img' for _, k in slots] + [True] * len(extra)
        t = self.prefetch(wanted + list(extra), full)
        t.slots, t.nframes = slots, len(sci_paths)
        return t

    def frames(self, ticket, stream, zp_key='MAGZP'):
        """(DeviceFrames, frame dicts, [(tensor, header) of the extras]) of a ``prefetch_frames`` ticket, ordered
        behind the decodes on ``stream`` (``FLXSCALE = 10^(-0.4 (MAGZP - 25))``, ``zuds/swarp.py:31``)."""
        from .device import DeviceFrames
        from .wcs import WCS
        loaded = ticket.result(stream)
        frames = [dict() for _ in range(ticket.nframes)]
        for (i, key), (t, hdr) in zip(ticket.slots, loaded):
            frames[i][key] = t
            if key == 'img':
                frames[i].update(wcs=WCS.from_header(hdr), header=hdr,
                                 flxscale=10 ** (-0.4 * (float(hdr.get(zp_key, 25.0)) - 25.0)))
        return DeviceFrames(frames, self.device), frames, loaded[len(ticket.slots):]

    # -- out --------------------------------------------------------------------------------
    def save(self, path, tensor, header=None, comments=None, bitpix=None, engine=None, stream=None):
        """Hand a product over; returns a Future of the path.  float32 -> BITPIX -32, int32 -> 32 (or 16 with
        ``bitpix=16``), uint8 / bool -> 8: the file ``FITSDeviceIO.save`` writes.

        ``engine`` / ``stream``: where the plane was produced (the engine must be bound to that stream, as the
        device chains keep it); the encode is enqueued there, behind the producer's kernels, so that the plane may
        be reused as soon as this call returns.  Without them the caller's current torch stream is waited for and
        the encode runs on the ring's return stream."""
        torch = self.torch
        if engine is None:
            self.ds.wait_stream(torch.cuda.current_stream(self.device))
            engine, stream = self.deng, self.ds
        with torch.cuda.stream(stream):
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
            hdr_bytes = fits.header_block(tuple(t.shape), bp, header, comments)   # (here: the caller may change the dict)
            pin = self._pin_out.get(nbytes)
            d_raw = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(engine.L.zm_fits_encode_dev(engine.ctx, t.data_ptr(), kind, t.numel(), d_raw.data_ptr()),
                  'zm_fits_encode_dev')
            encoded = stream.record_event()
        with torch.cuda.stream(self.ds):
            self.ds.wait_event(encoded)
            pin[:nbytes].copy_(d_raw, non_blocking=True)
            d_raw.record_stream(self.ds)
            ev = self.ds.record_event()
        fut = self._writers.submit(self._write, path, pin, nbytes, hdr_bytes, ev)
        with self._lock:
            self._pending.append(fut)
        return fut

    def _write(self, path, pin, nbytes, hdr_bytes, ev):
        try:
            ev.synchronize()
            raw = memoryview(pin.numpy())[:nbytes]
            with open(path, 'wb', buffering=0) as f:
                f.write(hdr_bytes)
                f.write(raw)
                f.write(b'\0' * (-nbytes % fits.BLOCK))
            self.stats['files_out'] += 1
            self.stats['bytes_out'] += nbytes
        finally:
            self._pin_out.put(pin)
        return path

    def flush(self):
        """Wait until every product handed to ``save`` so far is on disk; raises the first writer error."""
        err = None
        while True:
            with self._lock:
                if not self._pending:
                    break
                fut = self._pending.popleft()
            try:
                fut.result()
            except BaseException as e:            # noqa
                err = err or e
        if err is not None:
            raise err

    def close(self):
        try:
            self.flush()
        finally:
            self._tickets.put(None)
            self._feeder.join()
            self._readers.shutdown(wait=True)
            self._writers.shutdown(wait=True)
            self.ceng.close()
            self.deng.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
