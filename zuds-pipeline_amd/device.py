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
            msk = self._dev(f.get('mask'), torch.int32, device)
            self.tensors.append((img, wgt, msk))
            self.arr[i].img = img.data_ptr()
            self.arr[i].wgt = wgt.data_ptr() if wgt is not None else None
            self.arr[i].mask = msk.data_ptr() if msk is not None else None
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

    def __init__(self, wout, params=None, device=0, engine=None, want_mask=False):
        torch = _torch()
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.engine = engine or get_engine(device)
        # a dedicated non-default stream: its handle is never 0, and torch ops
        # (RCCL collectives, copies) issued inside `with torch.cuda.stream(...)`
        # order against the kernels libzudsmi enqueues on the same stream
        self.stream = torch.cuda.Stream(self.device)
        self.engine.set_stream(self.stream.cuda_stream)
        self.params = params or coadd_params()
        self.wout = wcs_struct(wout)
        onx, ony = self.wout.naxis[0], self.wout.naxis[1]
        self.shape = (ony, onx)
        self.img = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        self.wgt = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        self.mask = self.mask_wgt = None
        if want_mask:
            self.mask = torch.empty(self.shape, dtype=torch.int32, device=self.device)
            self.mask_wgt = torch.empty(self.shape, dtype=torch.float32, device=self.device)

    def run(self, dframes, partial=False):
        """Enqueue resample + combine of ``dframes`` (a DeviceFrames)."""
        L = self.engine.L
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
        self.run(dframes, partial=True)
        with self.torch.cuda.stream(self.stream):
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                dist.all_reduce(self.img, op=dist.ReduceOp.SUM, group=group)
                dist.all_reduce(self.wgt, op=dist.ReduceOp.SUM, group=group)
            check(self.engine.L.zm_coadd_finalize_dev(self.engine.ctx, self.img.data_ptr(),
                                                      self.wgt.data_ptr(), self.img.numel()),
                  'zm_coadd_finalize_dev')
        return self.img, self.wgt
