"""The device route under the object API (round 4).

``Coadd.from_images`` / ``Subtraction.from_images`` are what ``scripts/dostack.py``, ``dosub.py`` and
``makeref.py`` call.  Through the host-pointer C-ABI they spend nine tenths of their time outside the
engine: numpy decodes every FITS block, masks are widened, pageable arrays cross PCIe, products come back,
are byte-swapped and written, then read and written AGAIN by ``save()`` for a few header cards
(the I/O the reference does at ``zuds/coadd.py:61-93,165-217`` and ``zuds/subtraction.py:68-99,151-183``).
This module is the same work on the planes the bench times: raw data blocks go to the GPU as they lie on
disk (several files in flight on reader threads), are decoded there (``zm_fits_decode_dev``), coadded /
subtracted by ``DeviceCoadd`` / ``DeviceSubtraction``, the bookkeeping (bit 16, pedestal, rms maps,
seeing) runs on the device planes, products are encoded on the GPU and every file is written ONCE, with
its final header.  An object whose pixels are already in memory (loaded, modified, derived) contributes
that array instead of its file - there is no separate fallback path for "unmapped" inputs.

``ZM_OBJECT_API=host`` keeps the host-pointer route (the two are compared plane by plane in
``tests/test_object_route_gpu.py``).
"""
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import fits as _fits

_state = {}
NREADERS = 4          # files being read at a time (page-cache reads scale with threads)


def enabled():
    if os.environ.get('ZM_OBJECT_API', 'device') == 'host':
        return False
    try:
        import torch
    except ImportError:                # the host-pointer route needs no torch
        return False
    return torch.cuda.is_available()


class ObjectIO(object):
    """One per process: the engine's stream, pinned staging buffers, reader threads."""

    def __init__(self, device=0):
        import torch
        from .device import FITSDeviceIO
        from .engine import get_engine
        self.torch = torch
        self.engine = get_engine(device)
        self.device = torch.device('cuda', device)
        self.stream = torch.cuda.Stream(self.device)
        self.io = FITSDeviceIO(device, engine=self.engine, stream=self.stream)
        self.pool = ThreadPoolExecutor(NREADERS)
        self._free = queue.Queue()
        self._pins = 0

    # -- inputs -------------------------------------------------------------------------------
    def _pin(self, nbytes):
        """A pinned staging buffer of at least nbytes (a small pool; a buffer is reused once the copy
        out of it has been waited for)."""
        torch = self.torch
        try:
            while True:
                buf, ev = self._free.get_nowait()
                if ev is not None:
                    ev.synchronize()
                if buf.numel() >= nbytes:
                    return buf
                del buf                                   # too small: let it go, allocate a larger one
        except queue.Empty:
            pass
        return torch.empty(int(nbytes) + 4096, dtype=torch.uint8, pin_memory=True)

    def planes(self, wanted):
        """wanted: list of (object, kind) with kind 'f32', 'i32', 'u8' or 'mask' (int16 for a BITPIX 16
        file or an int16 array, else int32).  -> list of device tensors, in order; objects with pixels in
        memory give those, mapped objects their file (raw block, decoded on the device: several files in
        flight on reader threads, ``FITSDeviceIO.load_many``)."""
        torch = self.torch
        out = [None] * len(wanted)
        files = [(i, obj.local_path, kind) for i, (obj, kind) in enumerate(wanted)
                 if obj is not None and '_data' not in obj.__dict__]
        for (i, _, _), (t, _) in zip(files, self.io.load_many([(p, k) for _, p, k in files], NREADERS)):
            out[i] = t
        with torch.cuda.stream(self.stream):
            for i, (obj, kind) in enumerate(wanted):
                if obj is None or '_data' not in obj.__dict__:
                    continue
                a = np.asarray(obj.__dict__['_data'])
                if kind == 'mask':
                    dt = torch.int16 if a.dtype == np.int16 else torch.int32
                else:
                    dt = {'f32': torch.float32, 'i32': torch.int32, 'u8': torch.uint8}[kind]
                if a.dtype == np.bool_:
                    a = a.astype(np.uint8)
                out[i] = torch.from_numpy(np.ascontiguousarray(a)).to(self.device).to(dt)
        return out

    # -- products -----------------------------------------------------------------------------
    def save_all(self, items):
        """items: list of (path, tensor, header, comments[, bitpix]).  Encoded on the device, copied back
        into pinned buffers, written by the reader threads; returns when every file is on disk."""
        torch = self.torch
        from ._lib import check
        self.engine.set_stream(self.stream.cuda_stream)
        staged = []
        with torch.cuda.stream(self.stream):
            for it in items:
                path, t, hdr, com = it[:4]
                bitpix = it[4] if len(it) > 4 else None
                t = t.contiguous()
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
                pin = self._pin(nbytes)
                d_raw = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                check(self.engine.L.zm_fits_encode_dev(self.engine.ctx, t.data_ptr(), kind, t.numel(),
                                                       d_raw.data_ptr()), 'zm_fits_encode_dev')
                pin[:nbytes].copy_(d_raw, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                staged.append((path, pin, nbytes, tuple(t.shape), bp, hdr, com, ev))

        def write(s):
            path, pin, nbytes, shape, bp, hdr, com, ev = s
            ev.synchronize()
            _fits.write_raw(path, pin[:nbytes].numpy(), shape, bp, hdr, com)
            return pin
        for pin in self.pool.map(write, staged):
            self._free.put((pin, None))


def get_io(device=0):
    if device not in _state:
        _state[device] = ObjectIO(device)
    return _state[device]


def written_header(header, comments, shape, bitpix):
    """The (header, comments) an object gets from ``from_file`` after ``fits.write`` wrote these cards for an
    array of this shape: the host route writes a product, re-reads its header, adds cards and saves again -
    the device route reproduces the re-read (value formatting, dropped entries, the structural cards) without
    the file, so that the ONE write it does carries the same cards in the same order."""
    return _fits.parse_header(_fits.header_block(shape, bitpix, header, comments))
