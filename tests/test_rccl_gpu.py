"""The collectives of the multi-GPU coadds on RCCL itself (backend ``nccl``), in a process group of
one rank - all a one-GPU box can hold - with ``parallel.FORCE_COLLECTIVES`` so that every call an
8-GPU node makes is made: the single all-reduce over the two partial-sum planes, the mask
all-gather and the banded mask exchange, the row-band exchange of the exact coadd, all on the
engine's own stream.  The results must equal the single-process coadd bit for bit (the
multi-rank arithmetic is covered with gloo in test_sharded_gloo.py)."""
import importlib
import os

import numpy as np
import pytest

from util import pkg, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    par = importlib.import_module('zuds-pipeline_amd.parallel')
    par.FORCE_COLLECTIVES = True
    try:
        yield dist
    finally:
        par.FORCE_COLLECTIVES = False
        dist.destroy_process_group()


def stack(n, nx, ny):
    s = synth()
    base = s.ztf_wcs(nx, ny, tpv=True)
    rng = np.random.default_rng(5)
    frames = []
    for i in range(n):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-8, 8), dy=rng.uniform(-8, 8), rot_deg=rng.uniform(-0.1, 0.1))
        frames.append(s.make_frame(nx, ny, 50 + i, w, nstars=40, nbad=300))
    return frames, base


@pytest.mark.parametrize('banded', ['0', '1'])
def test_weighted_coadd_through_rccl_all_reduce_and_mask_exchange(engine, rccl_group, banded):
    import torch
    z = pkg()
    dmod = importlib.import_module('zuds-pipeline_amd.device')
    frames, base = stack(4, 640, 600)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='AND', subtract_back=True, rescale_weights=True)
    want = engine.coadd(frames, base, p, want_mask=True)
    os.environ['ZM_MASK_BANDED'] = banded
    try:
        dc = dmod.DeviceCoadd(base, p, device=0, engine=engine, want_mask=True)
        dfr = dmod.DeviceFrames(frames, dc.device)
        dc.run_sharded_weighted(dfr)
        dc.stream.synchronize()
        rccl_group.barrier(device_ids=[0])
    finally:
        os.environ.pop('ZM_MASK_BANDED', None)
        engine.set_stream(0)
    got = [t.cpu().numpy() for t in (dc.img, dc.wgt, dc.mask)]
    for a, b, name in zip(got, want, ('img', 'wgt', 'mask')):
        assert np.array_equal(a, b, equal_nan=True), name
    assert (got[2] != 0).any() and (got[1] > 0).mean() > 0.9


@pytest.mark.parametrize('kind', ['CLIPPED', 'WEIGHTED'])
def test_sharded_coadd_classes_on_rccl(engine, rccl_group, kind):
    z = pkg()
    par = importlib.import_module('zuds-pipeline_amd.parallel')
    frames, base = stack(5, 520, 500)
    p = z.coadd_params(combine=kind, mask_combine='OR', subtract_back=False, rescale_weights=False)
    want = engine.coadd(frames, base, p, want_mask=True)
    try:
        be = par.HipBackend(base, p, device=0, engine=engine)
        sc = par.ShardedCoadd(be)
        if kind == 'CLIPPED':
            img, wgt = sc.exact(frames, want_mask=True)
            mask = be.reduce_mask()
            be.stream.synchronize()
            assert np.array_equal(mask.cpu().numpy(), want[2])
        else:
            img, wgt = sc.weighted(frames)
        be.stream.synchronize()
    finally:
        engine.set_stream(0)
    assert np.array_equal(img.cpu().numpy(), want[0], equal_nan=True)
    assert np.array_equal(wgt.cpu().numpy(), want[1], equal_nan=True)


def test_native_rccl_layer_of_libzudsmi(engine):
    """zm_comm_* (csrc/comm.hip): librccl opened from inside libzudsmi, a communicator of one rank,
    the all-reduce over the two partial-sum planes and the mask reduce on the engine's stream; the
    coadd equals the single-call coadd bit for bit (ZM_NATIVE_RCCL=1 in run_sharded_weighted)."""
    z = pkg()
    dmod = importlib.import_module('zuds-pipeline_amd.device')
    par = importlib.import_module('zuds-pipeline_amd.parallel')
    frames, base = stack(4, 640, 600)
    p = z.coadd_params(combine='WEIGHTED', mask_combine='OR', subtract_back=True, rescale_weights=True)
    want = engine.coadd(frames, base, p, want_mask=True)
    os.environ['ZM_NATIVE_RCCL'] = '1'
    try:
        dc = dmod.DeviceCoadd(base, p, device=0, engine=engine, want_mask=True)
        dfr = dmod.DeviceFrames(frames, dc.device)
        dc.run_sharded_weighted(dfr)
        dc.stream.synchronize()
        assert isinstance(dc._native, par.NativeComm) and dc._native.world == 1
        got = [t.cpu().numpy() for t in (dc.img, dc.wgt, dc.mask, dc.mask_wgt)]
        dc._native.close()
    finally:
        os.environ.pop('ZM_NATIVE_RCCL', None)
        engine.set_stream(0)
    for a, b, name in zip(got, want, ('img', 'wgt', 'mask', 'mask coverage')):
        assert np.array_equal(a, b, equal_nan=True), name
    # an all-reduce that really runs: twice the planes in, the same planes out of a world of one
    import torch
    nc = par.NativeComm(engine)
    both = torch.arange(2 * 1000, dtype=torch.float32, device='cuda').reshape(2, 1000)
    nc.all_reduce_planes(both[0], both[1])
    engine.synchronize()
    assert torch.equal(both.cpu().ravel(), torch.arange(2000, dtype=torch.float32))
    nc.close()
