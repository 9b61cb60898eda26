"""Multi-GPU coaddition: one process per GPU, frames sharded across ranks
(SURVEY.md section 8(e)).

* WEIGHTED / AVERAGE: every rank reduces its frames to the partial sums
  S1 = sum(w v), S0 = sum(w) on the common grid - two planes of ONE buffer, one
  all-reduce (RCCL over xGMI with the ``nccl`` backend: a reduce-scatter and an
  all-gather over the 7 links of a GPU, 2 x 75.5 MB / 8 per link and phase), then S1 / S0.
* CLIPPED / MEDIAN are not sums: the per-pixel median needs every sample, so the
  resampled stacks are transposed from frame-sharded to row-band-sharded with an
  all-to-all (rank g receives rows [b_g, b_{g+1}) of all N frames), each rank
  combines its band exactly, and the bands are all-gathered.

The arithmetic lives behind a small backend interface so that the sharding and
collective logic is testable on CPU with the ``gloo`` backend; the product
backend is :class:`HipBackend` (libzudsmi on torch-allocated HBM).
"""
import ctypes as C

from ._lib import check, wcs_struct


# Tests set this to run every collective even in a process group of one rank (RCCL accepts a
# communicator of one): the one-GPU box then exercises the same calls an 8-GPU node makes.
FORCE_COLLECTIVES = False

# Developer / bench: wall-clock seconds per named exchange step of this rank (`bench.py --gpus N` sets it to a dict
# for ONE extra step behind its timed region and prints it per rank: the first real multi-GPU run should explain
# itself - VERDICT r4 item 8).  Each probed step is bracketed by device synchronisations, so it is never on.
PROBE = None


class _probed(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PROBE is not None:
            import time
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if PROBE is not None:
            import time
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            PROBE[self.name] = PROBE.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


def band_bounds(nrows, world):
    """Row-band boundaries, ``np.array_split`` semantics (as ``zuds/mpi.py:52-60``
    splits job lists)."""
    base, extra = divmod(nrows, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < extra else 0))
    return b


def reduce_masks(acc, accum, finalize, group=None, banded=None):
    """Fold the partial mask coadds of all ranks into ``acc`` (in place) and finalise.

    ``acc``: this rank's partial mask (2-D tensor, -1 where none of its frames covers the
    pixel).  ``accum(acc, m, first)`` folds one partial mask in, ``finalize(acc)`` turns the
    marker into 0 / writes the coverage plane.  AND / OR are associative and commutative, so
    every rank ends with the same mask whatever the order.  NCCL / RCCL have no bitwise
    reductions; two exact schedules:

    * all-gather of the whole masks and a local fold: world x 4 B / px per rank (one collective
      of the most travelled kind; the default below 4 ranks);
    * ``banded`` (the default from 4 ranks on; ``banded=`` or ZM_MASK_BANDED=0 / 1 override): a
      reduce-scatter by hand - rank g receives row band g of every rank (grouped send / recv, as
      the CLIPPED exchange; the bands are row ranges of a C-contiguous plane, sent in place), folds
      it, and the folded bands are all-gathered: 2 x 4 B / px per rank whatever the world size
      (8 ranks: 75 MB instead of 302 MB received per rank).
    """
    import torch
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    if banded is None:
        import os
        env = os.environ.get('ZM_MASK_BANDED', '')
        banded = (world >= 4) if env == '' else env != '0'
    multi = world > 1 or (on and FORCE_COLLECTIVES)
    if multi and not banded:
        parts = [torch.empty_like(acc) for _ in range(world)]
        with _probed('mask_all_gather'):
            dist.all_gather(parts, acc.contiguous(), group=group)
        for r, m in enumerate(parts):
            accum(acc, m, r == 0)
    elif multi:
        ny = acc.shape[0]
        bounds = band_bounds(ny, world)
        peer = (lambda g: g) if group is None else (lambda g: dist.get_global_rank(group, g))
        mine = acc[bounds[rank]:bounds[rank + 1]]
        recv = [torch.empty_like(mine) for _ in range(world)]
        recv[rank].copy_(mine)
        ops = []
        for g in range(world):
            if g == rank:
                continue
            band = acc[bounds[g]:bounds[g + 1]]          # rows of a contiguous plane: contiguous
            if band.numel():
                ops.append(dist.P2POp(dist.isend, band, peer(g), group))
            if recv[g].numel():
                ops.append(dist.P2POp(dist.irecv, recv[g], peer(g), group))
        if ops:
            with _probed('mask_band_exchange'):
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        folded = torch.empty_like(mine)
        for r, m in enumerate(recv):
            if m.numel():
                accum(folded, m, r == 0)
        # bands may differ by one row: gather through padded buffers
        maxr = max(bounds[g + 1] - bounds[g] for g in range(world))
        pad = torch.empty((maxr,) + tuple(acc.shape[1:]), dtype=acc.dtype, device=acc.device)
        pad[:folded.shape[0]] = folded
        allp = [torch.empty_like(pad) for _ in range(world)]
        with _probed('mask_band_gather'):
            dist.all_gather(allp, pad, group=group)
        for g in range(world):
            acc[bounds[g]:bounds[g + 1]] = allp[g][:bounds[g + 1] - bounds[g]]
    finalize(acc)
    return acc


def all_reduce_planes(s1, s0, group=None):
    """Sum-reduce the two partial-sum planes across ranks with ONE collective when they are the
    two halves of one buffer (``HipBackend.partial_sums``, ``DeviceCoadd``), else one each."""
    import torch.distributed as dist
    n = s1.numel()
    if (s1.is_contiguous() and s0.is_contiguous() and s1.dtype == s0.dtype
            and s1.untyped_storage().data_ptr() == s0.untyped_storage().data_ptr()
            and s0.storage_offset() == s1.storage_offset() + n):
        both = s1.new_empty(0).set_(s1.untyped_storage(), s1.storage_offset(), (2 * n,))
        with _probed('all_reduce_planes'):
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    else:
        with _probed('all_reduce_planes'):
            dist.all_reduce(s1, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(s0, op=dist.ReduceOp.SUM, group=group)


class NativeComm(object):
    """The reductions of the frame-sharded WEIGHTED coadd on RCCL called from inside libzudsmi
    (``zm_comm_*``, csrc/comm.hip) instead of through torch.distributed: the same call pattern -
    one all-reduce over the two partial-sum planes, the banded exchange of the partial masks -
    enqueued on the engine's stream.  torch.distributed (any backend) only carries the 128-byte
    RCCL id from rank 0 to the others; a world of one needs no process group at all.
    ``ZM_NATIVE_RCCL=1`` makes ``DeviceCoadd.run_sharded_weighted`` use it."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.engine = engine
        L = engine.L
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            check(L.zm_comm_unique_id(ident), 'zm_comm_unique_id')
        if self.world > 1:
            box = [ident.raw]
            src = 0 if group is None else dist.get_global_rank(group, 0)
            dist.broadcast_object_list(box, src=src, group=group)
            ident = C.create_string_buffer(box[0], 128)
        self._comm = C.c_void_p()
        check(L.zm_comm_init(engine.ctx, self.world, self.rank, ident, C.byref(self._comm)), 'zm_comm_init')

    def all_reduce_planes(self, s1, s0):
        """s1, s0: the two planes of one buffer (``DeviceCoadd``, ``HipBackend.partial_sums``)."""
        n = s1.numel()
        if not (s1.is_contiguous() and s0.is_contiguous() and s0.data_ptr() == s1.data_ptr() + 4 * n):
            raise ValueError('the partial sums must be the two planes of one float32 buffer')
        check(self.engine.L.zm_coadd_reduce_dev(self.engine.ctx, self._comm, s1.data_ptr(), n), 'zm_coadd_reduce_dev')

    def reduce_mask(self, mask, kind, cov=None):
        ny, nx = mask.shape
        check(self.engine.L.zm_mask_reduce_dev(self.engine.ctx, self._comm, mask.data_ptr(), nx, ny, int(kind),
                                               cov.data_ptr() if cov is not None else None), 'zm_mask_reduce_dev')

    def close(self):
        if self._comm:
            self.engine.L.zm_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipBackend(object):
    """libzudsmi on this rank's GPU; tensors are CUDA (HIP) tensors."""

    def __init__(self, wout, params, device=0, engine=None):
        import torch
        from .engine import get_engine
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.engine = engine or get_engine(device)
        self.stream = torch.cuda.Stream(self.device)
        self.engine.set_stream(self.stream.cuda_stream)
        self.wout = wcs_struct(wout)
        self.params = params
        self.shape = (self.wout.naxis[1], self.wout.naxis[0])

    def frames(self, frames):
        from .device import DeviceFrames
        return frames if isinstance(frames, DeviceFrames) else DeviceFrames(frames, self.device)

    def resample_stack(self, frames, want_mask=False):
        """(n, ny, nx, 2) float32 {value, weight} stack of this rank's frames; with
        ``want_mask`` also the partial mask coadd (-1 = not covered) as ``self.partial_mask``."""
        torch = self.torch
        df = self.frames(frames)
        ny, nx = self.shape
        stack = torch.empty((df.n, ny, nx, 2), dtype=torch.float32, device=self.device)
        self.partial_mask = torch.empty((ny, nx), dtype=torch.int32, device=self.device) if want_mask else None
        self.engine.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            check(self.engine.L.zm_resample_stack_dev(
                self.engine.ctx, df.n, df.arr, C.byref(self.wout), C.byref(self.params), stack.data_ptr(),
                self.partial_mask.data_ptr() if want_mask else None), 'zm_resample_stack_dev')
        return stack

    def combine(self, stack):
        """stack (N, rows, nx, 2) -> (img, wgt) of shape (rows, nx)."""
        torch = self.torch
        stack = stack.contiguous()
        n, rows, nx, _ = stack.shape
        self.engine.set_stream(self.stream.cuda_stream)
        # whatever produced `stack` on the caller's stream is complete before the kernels here
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            img = torch.empty((rows, nx), dtype=torch.float32, device=self.device)
            wgt = torch.empty((rows, nx), dtype=torch.float32, device=self.device)
            if rows * nx == 0:
                return img, wgt
            check(self.engine.L.zm_combine_stack_dev(self.engine.ctx, n, stack.data_ptr(),
                                                     rows * nx, rows * nx, C.byref(self.params),
                                                     img.data_ptr(), wgt.data_ptr()),
                  'zm_combine_stack_dev')
        return img, wgt

    def partial_sums(self, frames):
        """(s1, s0): the two planes of one (2, ny, nx) buffer, so that a single collective reduces both."""
        torch = self.torch
        df = self.frames(frames)
        self.engine.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            both = torch.empty((2,) + tuple(self.shape), dtype=torch.float32, device=self.device)
            s1, s0 = both[0], both[1]
            check(self.engine.L.zm_coadd_dev(self.engine.ctx, df.n, df.arr, C.byref(self.wout),
                                             C.byref(self.params), 1, s1.data_ptr(),
                                             s0.data_ptr(), None, None), 'zm_coadd_dev')
        return s1, s0

    def finalize(self, s1, s0):
        self.engine.set_stream(self.stream.cuda_stream)
        with self.torch.cuda.stream(self.stream):
            check(self.engine.L.zm_coadd_finalize_dev(self.engine.ctx, s1.data_ptr(),
                                                      s0.data_ptr(), s1.numel()),
                  'zm_coadd_finalize_dev')
        return s1, s0

    def reduce_mask(self, group=None, cov=None):
        """Fold the partial masks of all ranks (after ``resample_stack(want_mask=True)``) and
        finalise; returns the mask tensor (``cov``: optional float32 coverage plane)."""
        L, ctx = self.engine.L, self.engine.ctx
        m = self.partial_mask
        n, kind = m.numel(), int(self.params.mask_combine)
        self.engine.set_stream(self.stream.cuda_stream)
        with self.torch.cuda.stream(self.stream):
            reduce_masks(m,
                         lambda acc, x, first: check(L.zm_mask_accum_dev(ctx, acc.data_ptr(), x.data_ptr(),
                                                                         acc.numel(), kind, int(first)),
                                                     'zm_mask_accum_dev'),
                         lambda acc: check(L.zm_mask_finalize_dev(ctx, acc.data_ptr(),
                                                                  cov.data_ptr() if cov is not None else None,
                                                                  n), 'zm_mask_finalize_dev'),
                         group)
        return m

    def scope(self):
        return self.torch.cuda.stream(self.stream)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)


class ShardedCoadd(object):
    """Frame-sharded coadd over the default (or a given) process group."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def _peer(self, g):
        import torch.distributed as dist
        return g if self.group is None else dist.get_global_rank(self.group, g)

    def weighted(self, frames):
        """Sum-reduce coadd; every rank returns the full (img, wgt)."""
        import torch.distributed as dist
        rank, world = self._world()
        s1, s0 = self.backend.partial_sums(frames)
        with self.backend.scope():
            if world > 1 or (FORCE_COLLECTIVES and dist.is_initialized()):
                all_reduce_planes(s1, s0, self.group)
        return self.backend.finalize(s1, s0)

    def exact(self, frames, want_mask=False):
        """Exact CLIPPED / MEDIAN (or any combine) through the row-band
        transpose; every rank returns the full (img, wgt).  ``want_mask``: the backend also
        keeps this rank's partial mask coadd (``backend.reduce_mask()`` folds the ranks)."""
        import torch
        import torch.distributed as dist
        rank, world = self._world()
        stack = (self.backend.resample_stack(frames, want_mask=True) if want_mask
                 else self.backend.resample_stack(frames))   # (n_local, ny, nx, 2)
        n_local, ny, nx, _ = stack.shape
        if world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
            return self.backend.combine(stack)
        bounds = band_bounds(ny, world)
        with self.backend.scope():
            counts = torch.tensor([n_local], dtype=torch.int64, device=stack.device)
            allc = [torch.zeros_like(counts) for _ in range(world)]
            dist.all_gather(allc, counts, group=self.group)
            nfr = [int(c.item()) for c in allc]
            my_rows = bounds[rank + 1] - bounds[rank]
            # band g of my stack goes to rank g; I receive my band of every rank's stack.  Rows
            # [b_g, b_g+1) of ONE frame are contiguous in the (n, ny, nx, 2) stack, so every frame
            # is sent in place (no per-destination copy of the stack) and received straight into
            # its slot of the (N, my_rows, nx, 2) band (no concatenation afterwards): the exchange
            # holds the stack and the band, nothing else.  Grouped point-to-point operations
            # (ncclSend / ncclRecv on RCCL; also available on gloo, which has no all-to-all).
            first = [0]
            for g in range(world):
                first.append(first[-1] + nfr[g])
            band = self.backend.empty((first[-1], my_rows, nx, 2))
            if my_rows:
                band[first[rank]:first[rank + 1]].copy_(stack[:, bounds[rank]:bounds[rank + 1]])
            ops = []
            for g in range(world):
                if g == rank:
                    continue
                if bounds[g + 1] > bounds[g]:
                    for i in range(n_local):
                        ops.append(dist.P2POp(dist.isend, stack[i, bounds[g]:bounds[g + 1]], self._peer(g), self.group))
                if my_rows:
                    for i in range(nfr[g]):
                        ops.append(dist.P2POp(dist.irecv, band[first[g] + i], self._peer(g), self.group))
            if ops:
                with _probed('stack_band_exchange'):
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
        with _probed('band_combine'):
            img_b, wgt_b = self.backend.combine(band)
        with self.backend.scope():
            # bands may differ by one row: gather through padded buffers
            maxr = max(bounds[g + 1] - bounds[g] for g in range(world))
            pad = self.backend.empty((2, maxr, nx))
            pad[0, :my_rows] = img_b
            pad[1, :my_rows] = wgt_b
            allp = [torch.empty_like(pad) for _ in range(world)]
            with _probed('band_gather'):
                dist.all_gather(allp, pad, group=self.group)
            img = torch.cat([allp[g][0, :bounds[g + 1] - bounds[g]] for g in range(world)], dim=0)
            wgt = torch.cat([allp[g][1, :bounds[g + 1] - bounds[g]] for g in range(world)], dim=0)
        return img, wgt
