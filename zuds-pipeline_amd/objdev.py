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
        # planes that are already in HBM, by file: what derive_maps made (and wrote) or read on the way - the
        # from_images call that follows takes them from here instead of the file (ZM_OBJDEV_CACHE_GB, default 4: one from_images call of 32 frames)
        self._cache = {}
        self._cache_bytes = 0
        self._cache_cap = int(float(os.environ.get('ZM_OBJDEV_CACHE_GB', '4')) * 1e9)
        self._prestamp = {}                 # (path, kind) -> the file's stamp taken BEFORE it was read

    # -- plane cache --------------------------------------------------------------------------
    @staticmethod
    def _stamp(path):
        st = os.stat(path)
        return (st.st_mtime_ns, st.st_size)

    def cache_put(self, path, kind, tensor):
        """Remember that `tensor` is the decoded data of the file at `path`.  The entry carries the file's stamp
        (mtime in ns, size) from BEFORE the read that produced the plane (`planes` notes it) - a rewrite between the
        read and this call then leaves a stale stamp behind and the entry is never served; a plane this process
        wrote itself (derived maps) carries the stamp of the file as written.  Planes are handed out by reference:
        READ-ONLY for every consumer (the kernels that take them only read; host code must clone before it changes
        one)."""
        try:
            key = (os.path.abspath(path), kind)
            stamp = self._prestamp.pop(key, None) or self._stamp(path)
        except OSError:
            return
        nbytes = tensor.numel() * tensor.element_size()
        old = self._cache.pop(key, None)
        if old is not None:
            self._cache_bytes -= old[2]
        while self._cache and self._cache_bytes + nbytes > self._cache_cap:
            k0 = next(iter(self._cache))                  # oldest first (dicts keep insertion order)
            self._cache_bytes -= self._cache.pop(k0)[2]
        if nbytes <= self._cache_cap:
            self._cache[key] = (tensor, stamp, nbytes)
            self._cache_bytes += nbytes

    def cache_get(self, path, kind):
        key = (os.path.abspath(path), kind)
        hit = self._cache.get(key)
        if hit is None:
            return None
        try:
            if self._stamp(path) != hit[1]:               # the file changed since: not this plane any more
                raise OSError
        except OSError:
            self._cache_bytes -= self._cache.pop(key)[2]
            return None
        return hit[0]

    def cache_clear(self):
        self._prestamp.clear()
        self._cache.clear()
        self._cache_bytes = 0

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
        files = []
        for i, (obj, kind) in enumerate(wanted):
            if obj is None or '_data' in obj.__dict__:
                continue
            hit = self.cache_get(obj.local_path, kind)
            if hit is not None:
                out[i] = hit
            else:
                files.append((i, obj.local_path, kind))
                try:
                    self._prestamp[(os.path.abspath(obj.local_path), kind)] = self._stamp(obj.local_path)
                except OSError:
                    pass
        for (i, p, k), (t, _) in zip(files, self.io.load_many([(p, k) for _, p, k in files], NREADERS)):
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


def can_derive(image):
    """The maps of `image` can be made on the device: it is a file on disk whose pixels nobody holds (or has
    changed) in memory, and so is its mask."""
    m = getattr(image, 'mask_image', None)
    return (enabled() and m is not None and image.ismapped and m.ismapped and
            '_data' not in image.__dict__ and '_data' not in m.__dict__)


def derive_maps(image, want_weight=False):
    """``rms_image`` (and ``weight_image``) of a frame that came without ``.rms.fits`` / ``.weight.fits`` - the way
    ZTF delivers science frames - on the device.  The reference makes them with a SExtractor run
    and numpy (``zuds/image.py:136-208``, ``zuds/sextractor.py:80-96``, called from ``scripts/dosub.py:35-47`` and
    ``zuds/swarp.py:43-51``); the host route here does the same through the host-pointer ``zm_background``.  Here:
    image + mask go to HBM as they lie on disk (6 B per pixel), the false weight map, the mesh background's rms map
    and ``w = 1 / rms^2`` (0 under bad bits and within 10 % of SATURATE) are kernels on those planes, each derived
    map is encoded on the device and written ONCE next to the image (the reference saves them too), and the planes
    stay in the plane cache for the ``from_images`` call that asked for them.  Sets ``image._rmsimg`` (and
    ``_weightimg``) to file-mapped objects, like the host route after its ``save()``."""
    derive_maps_many([image], want_weight)


def derive_maps_many(images, want_weight=False):
    """``derive_maps`` for several frames at once (``Coadd.from_images`` asks every input for its weight map): the
    files of all frames are read by the reader threads while the planes that have arrived are worked on, and all
    derived maps are written by the same threads side by side - frame after frame the reads, kernels and writes of
    one frame would wait for each other (8 frames of 3072^2: 51 ms one by one)."""
    from . import _lib
    from .constants import BAD_SUM, BKG_BOX_SIZE, MASK_BORDER
    from .image import FITSImage
    images = [im for im in images if not hasattr(im, '_weightimg' if want_weight else '_rmsimg')]
    if not images:
        return
    oio = get_io()
    torch, eng, L = oio.torch, oio.engine, oio.engine.L
    check = _lib.check
    want = []
    for im in images:
        want += [(im, 'f32'), (im.mask_image, 'mask'), (getattr(im, '_rmsimg', None), 'f32')]
    planes = oio.planes(want)
    eng.set_stream(oio.stream.cuda_stream)
    items = []
    for k, image in enumerate(images):
        img, mask, rms = planes[3 * k:3 * k + 3]
        oio.cache_put(image.local_path, 'f32', img)
        oio.cache_put(image.mask_image.local_path, 'mask', mask)
        ny, nx = img.shape
        header, comments = dict(image.header), dict(image.header_comments or {})
        base = image.local_path
        mt = _lib.MASKTYPE_I16 if mask.dtype == torch.int16 else _lib.MASKTYPE_I32
        with torch.cuda.stream(oio.stream):
            bpm = torch.empty((ny, nx), dtype=torch.uint8, device=oio.device)
            if rms is None:
                fw = torch.empty((ny, nx), dtype=torch.float32, device=oio.device)
                border = MASK_BORDER if image.basename.endswith('sciimg.fits') else 0
                check(L.zm_false_weight_dev(eng.ctx, mask.data_ptr(), mt, BAD_SUM, border, nx, ny, fw.data_ptr(),
                                            bpm.data_ptr()), 'zm_false_weight_dev')
                rms = torch.empty((ny, nx), dtype=torch.float32, device=oio.device)
                check(L.zm_background_dev(eng.ctx, img.data_ptr(), fw.data_ptr(), nx, ny, int(BKG_BOX_SIZE), 3, None,
                                          rms.data_ptr(), None, None), 'zm_background_dev')
                items.append((image, '_rmsimg', base.replace('.fits', '.rms.fits'), rms, header, comments))
            else:
                check(L.zm_false_weight_dev(eng.ctx, mask.data_ptr(), mt, BAD_SUM, 0, nx, ny, None, bpm.data_ptr()),
                      'zm_false_weight_dev')
            if want_weight:
                wgt = torch.empty((ny, nx), dtype=torch.float32, device=oio.device)
                satur = 0.9 * float(header['SATURATE']) if 'SATURATE' in header else 0.0
                check(L.zm_weight_from_rms_dev(eng.ctx, rms.data_ptr(), bpm.data_ptr(),
                                               img.data_ptr() if satur else None, satur, img.numel(), wgt.data_ptr()),
                      'zm_weight_from_rms_dev')
                items.append((image, '_weightimg', base.replace('.fits', '.weight.fits'), wgt, header, comments))
    oio.save_all([(path, t, header, comments) for _, _, path, t, header, comments in items])
    for image, attr, path, t, header, comments in items:
        im = FITSImage()
        im.basename = os.path.basename(path)
        im.header, im.header_comments = dict(header), dict(comments)
        im.map_to_local_file(path)
        setattr(image, attr, im)
        oio.cache_put(path, 'f32', t)


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
