"""Concurrent subtractions on one GPU (nightly.SubtractionPool; BASELINE config 5:
scripts/donightly.py:21-40 + scripts/dophot.py:94-156 of the reference run one process per
job): J chains side by side give exactly the products of one chain at a time, which in turn
are the products of ``SingleEpochSubtraction.from_images`` (tests/test_device_chain_gpu.py);
the forced photometry of the pool equals ``raw_aperture_photometry`` on the same planes."""
import importlib

import numpy as np
import pytest

from util import pkg, synth

pytestmark = pytest.mark.gpu


def make_jobs(torch, z, s, njob, nx, ny, nreg_side, kws, seed=900, variables=0):
    base = s.ztf_wcs(nx, ny, tpv=True)
    rng = np.random.default_rng(seed)
    nst = int(nx * ny / 2500)
    xs, ys = rng.uniform(-10, nx + 10, nst), rng.uniform(-10, ny + 10, nst)
    fl = np.exp(rng.uniform(np.log(3e3), np.log(8e4), nst))
    ra, dec = base.all_pix2world(xs, ys, 0)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to('cuda:0')
    # the reference: a deep, clean frame on the base grid
    rf = s.make_frame(nx, ny, seed, base, star_sky=(ra, dec, fl), fwhm=2.0, noise=1.0, nbad=20)
    ref = dict(img=dev(rf['img'], np.float32), rms=dev(np.full((ny, nx), 1.0), np.float32),
               mask=dev(rf['mask'], np.int32), wcs=base, flxscale=1.0)
    pra, pdec = base.all_pix2world(rng.uniform(20, nx - 20, 60), rng.uniform(20, ny - 20, 60), 0)
    jobs = []
    for i in range(njob):
        w = s.ztf_wcs(nx, ny, dx=rng.uniform(-6, 6), dy=rng.uniform(-6, 6), rot_deg=rng.uniform(-0.05, 0.05))
        fli = fl.copy()
        if variables:
            # a few of the brighter stars changed their flux since the reference was taken: their stamps
            # fail the merit test and the fit needs rejection rounds, more in some jobs than in others
            # (bright, yet below the saturation limit of the command line - 5e3 at the peak - after the change)
            cand = np.flatnonzero((fl > 4e3) & (fl < 1e4))
            pick = rng.choice(cand, size=min((i % 3) * variables, cand.size), replace=False)
            fli[pick] *= rng.uniform(1.8, 2.5, pick.size)
        f = s.make_frame(nx, ny, seed + 1 + i, w, star_sky=(ra, dec, fli), fwhm=2.4, sky=180.0 + 10 * i, nbad=30)
        sci = dict(img=dev(f['img'], np.float32), rms=dev(np.full((ny, nx), 5.0), np.float32),
                   mask=dev(f['mask'], np.int32), wgt=dev(f['wgt'], np.float32), wcs=w, seeing=2.4)
        nm = importlib.import_module('zuds-pipeline_amd.nightly')
        jobs.append(nm.SubtractionJob(sci, ref, radec=(pra, pdec), nreg_side=nreg_side, hotpants_kws=kws, tag=i))
    return jobs


def test_pool_of_four_equals_one_at_a_time(engine):
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    jobs = make_jobs(torch, z, s, 6, 640, 600, 2, {'ko': 1, 'bgo': 0})
    one = nm.SubtractionPool(1)
    a = one.map(jobs)
    one.close()
    four = nm.SubtractionPool(4, batch=1)                 # four separate chains, one host thread each
    assert four.batch == 0 and four.njobs == 4
    b = four.map(jobs)
    c = four.map(jobs[::-1])[::-1]                        # another assignment of jobs to workers
    four.close()
    auto = nm.SubtractionPool(6)                          # round 6: six in flight = two lanes of three batched fits
    assert (auto.njobs, auto.batch) == (2, 3)
    two, twelve = nm.SubtractionPool(4), nm.SubtractionPool(12)
    assert (two.njobs, two.batch, twelve.njobs, twelve.batch) == (1, 0, 3, 4)      # (no thread or engine exists before map)
    two.close()
    twelve.close()
    e = auto.map(jobs)
    auto.close()
    for x, y, y2, y3 in zip(a, b, c, e):
        assert x['tag'] == y['tag'] == y2['tag'] == y3['tag']
        assert x['info'] == y['info'] == y2['info'] == y3['info'] and x['info']['status'] == 0
        for k in ('diff', 'noise', 'mask'):
            assert torch.equal(x[k], y[k]) and torch.equal(x[k], y2[k]) and torch.equal(x[k], y3[k]), k
        for k in ('flux', 'fluxerr', 'flags'):
            assert np.array_equal(x['phot'][k], y['phot'][k], equal_nan=True), k
    assert len({float(r['info']['kernel_sum']) for r in a}) > 1       # different jobs, really
    # the photometry of the pool is the host entry point on the same planes
    r = a[2]
    flux, err, flags = engine.aperture_photometry(r['diff'].cpu().numpy(), r['phot']['x'], r['phot']['y'],
                                                  rms=r['noise'].cpu().numpy(), mask=r['mask'].cpu().numpy())
    assert np.array_equal(flux, r['phot']['flux'], equal_nan=True)
    assert np.array_equal(err, r['phot']['fluxerr'], equal_nan=True)
    assert np.array_equal(flags, r['phot']['flags'])


def test_pool_at_the_reference_parameters(engine):
    """3 x 3 regions, ko = 4 (722 unknowns per region).  A lone job factors on the many-workgroup form
    of the solver, the jobs of a pool on the one-workgroup-per-region form: same bits."""
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    jobs = make_jobs(torch, z, s, 3, 1280, 1240, 3, {}, seed=950)
    for j in jobs:
        j.sci['seeing'] = 3.0
    one = nm.SubtractionPool(1)
    a = one.map(jobs)
    one.close()
    three = nm.SubtractionPool(3, batch=1)
    b = three.map(jobs)
    three.close()
    for x, y in zip(a, b):
        assert x['info'] == y['info'] and x['info']['status'] == 0 and x['info']['ncoeff'] == 722
        for k in ('diff', 'noise', 'mask'):
            assert torch.equal(x[k], y[k]), k


def test_fresh_wcs_objects_per_batch(engine):
    """scripts/donightly.py builds new sci dicts and WCS objects per batch and drops the old ones, so
    CPython hands the addresses of dead WCS objects to new ones (the worker once cached the
    WCS structs under id()).  Two batches of the same frame size and different geometry through ONE
    worker, the first batch garbage-collected in between: the second equals a fresh pool's result."""
    import gc
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    pool = nm.SubtractionPool(1)
    first = make_jobs(torch, z, s, 2, 640, 600, 1, {'ko': 1, 'bgo': 0}, seed=1300)
    ids = {id(j.sci['wcs']) for j in first}
    pool.map(first, keep=False)
    del first
    gc.collect()
    second = reused = None
    for attempt in range(20):                            # until an address really is handed out again
        second = make_jobs(torch, z, s, 2, 640, 600, 1, {'ko': 1, 'bgo': 0}, seed=1400 + attempt)
        reused = ids & {id(j.sci['wcs']) for j in second}
        if reused:
            break
        ids |= {id(j.sci['wcs']) for j in second}
        gc.collect()
    got = pool.map(second)
    pool.close()
    fresh = nm.SubtractionPool(1)
    want = fresh.map(second)
    fresh.close()
    for x, y in zip(got, want):
        assert x['info'] == y['info'] and x['info']['status'] == 0
        for k in ('diff', 'noise', 'mask'):
            assert torch.equal(x[k], y[k]), k
        assert np.array_equal(x['phot']['flux'], y['phot']['flux'], equal_nan=True)


def test_share_limits(engine):
    z = pkg()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    with pytest.raises(ValueError):
        nm.SubtractionPool(0)
    with pytest.raises(ValueError):
        nm.SubtractionPool(65)
    with pytest.raises(z.ZMError):
        engine.set_share(0)
    with pytest.raises(z.ZMError):
        engine.set_share(65)
    engine.set_share(1)


def test_differential_fuzz_across_solver_layouts():
    """40 random scenes and parameter sets (frame size, kernel half width, 1 - 9 regions, cells, ko 0 - 3,
    bgo 0 - 1) through contexts that own the GPU, share it 3 ways and 9 ways (26, 8 and 2 workgroups per
    region in the fused Cholesky): difference, noise and fit summary agree bit for bit
    (tools/fuzz_subtract.py; 150 cases were run once: no mismatch).  Round 4: every case also runs three scenes of
    its shape as ONE batch (zm_subtract_batch) against the three one at a time (560 cases run once: no mismatch)."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location(
        'fuzz_subtract', pathlib.Path(__file__).resolve().parent.parent / 'tools' / 'fuzz_subtract.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(40, 2026, verbose=False) == 0


def _same(torch, x, y):
    assert x['tag'] == y['tag']
    assert x['info'] == y['info'], (x['info'], y['info'])
    for k in ('diff', 'noise', 'mask'):
        assert torch.equal(x[k], y[k]), k
    for k in ('flux', 'fluxerr', 'flags'):
        assert np.array_equal(x['phot'][k], y['phot'][k], equal_nan=True), k


def test_batched_fits_equal_one_at_a_time(engine):
    """``SubtractionPool(J, batch=B)``: the kernel fits of B jobs as one chain of launches with the job as a
    grid dimension (``zm_subtract_batch_dev``).  Same bits as one job at a time, whatever the batch size, the
    number of lanes and the order of the jobs; jobs that converge in different rounds ride along."""
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    jobs = make_jobs(torch, z, s, 7, 640, 600, 2, {'ko': 1, 'bgo': 0}, seed=900, variables=8)
    one = nm.SubtractionPool(1)
    a = one.map(jobs)
    one.close()
    print('rounds per job:', [r['info']['niter'] for r in a], 'status', [r['info']['status'] for r in a])
    assert sum(r['info']['status'] == 0 for r in a) >= 5
    assert len({r['info']['niter'] for r in a}) > 1            # not every job needs the same rounds
    p = nm.SubtractionPool(1, batch=4)                          # batches of 4 and 3
    b = p.map(jobs)
    p.close()
    p = nm.SubtractionPool(2, batch=3)                          # two lanes: 3 + 3 + 1 (a batch of one: the lone path)
    c = p.map(jobs[::-1])[::-1]
    p.close()
    for x, y, y2 in zip(a, b, c):
        _same(torch, x, y)
        _same(torch, x, y2)


def test_batched_fits_at_the_reference_parameters(engine):
    """3 x 3 regions, ko = 4: 722 unknowns per region, 27 factorisations per launch of the batch."""
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    jobs = make_jobs(torch, z, s, 3, 1280, 1240, 3, {}, seed=950)
    for j in jobs:
        j.sci['seeing'] = 3.0
    one = nm.SubtractionPool(1)
    a = one.map(jobs)
    one.close()
    p = nm.SubtractionPool(1, batch=3)
    b = p.map(jobs)
    p.close()
    for x, y in zip(a, b):
        assert x['info']['ncoeff'] == 722
        _same(torch, x, y)


def test_batches_are_formed_per_fit_shape(engine):
    """Jobs whose fits differ in shape (regions, orders, seeing -> half widths) never share a batch; the C entry
    point refuses such a batch instead of fitting the wrong operator."""
    import ctypes as C
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    jobs = make_jobs(torch, z, s, 4, 640, 600, 2, {'ko': 1, 'bgo': 0}, seed=1700)
    jobs[1].nreg_side = 1
    jobs[2].hotpants_kws = {'ko': 2, 'bgo': 0}
    jobs[3].sci['seeing'] = 2.9
    assert len({nm._fit_key(j) for j in jobs}) == 4
    one = nm.SubtractionPool(1)
    a = one.map(jobs)
    one.close()
    p = nm.SubtractionPool(1, batch=4)
    b = p.map(jobs)
    p.close()
    for x, y in zip(a, b):
        _same(torch, x, y)
    # the entry point itself: two jobs, two shapes
    L = z._lib.lib()
    ny, nx = 600, 640
    f = torch.ones((ny, nx), dtype=torch.float32, device='cuda:0')
    out = torch.empty((4, ny, nx), dtype=torch.float32, device='cuda:0')
    hp = importlib.import_module('zuds-pipeline_amd.engine').hp_params
    from_kw = importlib.import_module('zuds-pipeline_amd.hotpants').job_params
    p0 = hp(**from_kw(2.4, nx, ny, 2, 0.0, 0.0, {'ko': 1}))
    p1 = hp(**from_kw(2.4, nx, ny, 2, 0.0, 0.0, {'ko': 2}))
    arr = (z._lib.zm_sub_job * 2)()
    for k, pp in enumerate((p0, p1)):
        arr[k] = z._lib.zm_sub_job(f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), None, C.pointer(pp),
                                   out[2 * k].data_ptr(), out[2 * k + 1].data_ptr())
    infos = (z._lib.zm_hp_info * 2)()
    rc = L.zm_subtract_batch_dev(engine.ctx, 2, arr, nx, ny, infos)
    assert rc != 0 and b'different fit' in L.zm_last_error()
    assert L.zm_subtract_batch_dev(engine.ctx, 0, arr, nx, ny, infos) != 0


def test_batch_with_an_unsolved_region_and_without_bad_pixel_maps(engine):
    """A job whose fit loses a region (status ZM_HP_UNSOLVED: fill value, bit 17) rides in a batch without touching
    the others, and its own products are those of the lone path; jobs straight through the C entry point with
    bpm = NULL and host-side data limits (no limits_dev) work too."""
    import ctypes as C
    import torch
    z, s = pkg(), synth()
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    jobs = make_jobs(torch, z, s, 5, 640, 600, 2, {'ko': 1, 'bgo': 0}, seed=2100, variables=8)
    # one quadrant of job 1 has no usable pixel: that region has no stamps
    jobs[1].sci['mask'][:300, :320] = 1
    one = nm.SubtractionPool(1)
    a = one.map(jobs)
    one.close()
    assert a[1]['info']['status'] & 1 and a[1]['info']['nunsolved'] >= 1
    p = nm.SubtractionPool(1, batch=5)
    b = p.map(jobs)
    p.close()
    for x, y in zip(a, b):
        _same(torch, x, y)
    # the entry point on plain planes: no bad-pixel map, limits from the host
    L = z._lib.lib()
    hp = importlib.import_module('zuds-pipeline_amd.engine').hp_params
    from_kw = importlib.import_module('zuds-pipeline_amd.hotpants').job_params
    ny, nx = 600, 640
    n = 3
    planes, arr, ps = [], (z._lib.zm_sub_job * n)(), []
    for k in range(n):
        sci, ref = jobs[k].sci, jobs[k].ref
        out = torch.empty((4, ny, nx), dtype=torch.float32, device='cuda:0')
        pp = hp(**from_kw(2.4, nx, ny, 2, 20.0 + k, 10.0, {'ko': 1, 'bgo': 0}))
        ps.append(pp)
        planes.append(out)
        arr[k] = z._lib.zm_sub_job(sci['img'].data_ptr(), sci['rms'].data_ptr(), ref['img'].data_ptr(),
                                   ref['rms'].data_ptr(), None, C.pointer(pp), out[0].data_ptr(), out[1].data_ptr())
    infos = (z._lib.zm_hp_info * n)()
    z._lib.check(L.zm_subtract_batch_dev(engine.ctx, n, arr, nx, ny, infos), 'batch')
    for k in range(n):
        info = z._lib.zm_hp_info()
        z._lib.check(L.zm_subtract_dev(engine.ctx, arr[k].sci, arr[k].sci_rms, arr[k].ref, arr[k].ref_rms, None, nx, ny,
                                       C.byref(ps[k]), planes[k][2].data_ptr(), planes[k][3].data_ptr(), C.byref(info)),
                     'lone')
        torch.cuda.synchronize()
        assert torch.equal(planes[k][0], planes[k][2]) and torch.equal(planes[k][1], planes[k][3])
        for f, _ in z._lib.zm_hp_info._fields_:
            assert getattr(info, f) == getattr(infos[k], f), f
