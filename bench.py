#!/usr/bin/env python
"""bench.py - Mpix/s through resample -> coadd -> subtract on 3072 x 3072 frames.

One step = one pass of the hot path over one batch of synthetic frames already
resident in HBM: BASELINE.json configs[1] (32 ZTF-CCD-sized TPV frames, Lanczos-3
resample + WEIGHTED coadd) followed by configs[2] (one science frame subtracted
against that coadd) once the subtraction kernels are built in.  Pixel accounting
(SURVEY.md section 8(d)): Mpix = (frames resampled + frames subtracted) x 9.437184.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling, every
rank resamples its own 32 frames of a 32 N deep stack, the two partial-sum planes
go through one RCCL all-reduce each, every rank then subtracts its own frame.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
RESAMPLE_BYTES_PER_OUTPX = 16  # img+var read, img+var write
MASK_BYTES_PER_OUTPX = 8       # + 4 B in / 4 B out when int32 masks ride along


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--size', type=int, default=3072)
    ap.add_argument('--combine', default='WEIGHTED')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-subtract', action='store_true')
    ap.add_argument('--no-mask', action='store_true', help='skip the mask coadd (dev only)')
    ap.add_argument('--seeing', type=float, default=4.0,
                    help='science FWHM in pixels: r = 2.5 seeing, rss = 6 seeing')
    ap.add_argument('--cpu-frames', type=int, default=8,
                    help='full-size frames the CPU baseline resamples and coadds')
    return ap.parse_args()


def make_device_frames(synth, torch, n, size, seed0, device):
    """Config-2 frames: star fields rendered on the host once, sky + noise added
    on the device (keeps set-up to seconds; the values follow synth.config2)."""
    base = synth.ztf_wcs(size, size, tpv=True)
    rng = np.random.default_rng(seed0 - 1)
    nstars = max(int(3000 * (size / 3072.0) ** 2), 10)
    xs = rng.uniform(-20, size + 20, nstars)
    ys = rng.uniform(-20, size + 20, nstars)
    fl = np.exp(rng.uniform(np.log(1e3), np.log(1e5), nstars))
    ra, dec = base.all_pix2world(xs, ys, 0)
    frames = []
    g = torch.Generator(device=device)
    for i in range(n):
        r = np.random.default_rng(seed0 + i)
        w = synth.ztf_wcs(size, size, dx=r.uniform(-15, 15), dy=r.uniform(-15, 15),
                          rot_deg=r.uniform(-0.1, 0.1), tpv=True)
        sky = r.uniform(100, 300)
        magzp = r.uniform(25.8, 26.6)
        fwhm = r.uniform(1.8, 2.6)
        stars = np.zeros((size, size), dtype=np.float64)
        px, py = w.all_world2pix(ra, dec, 0)
        synth.add_stars(stars, px, py, fl * 10 ** (0.4 * (magzp - 25.0)), fwhm)
        g.manual_seed(seed0 + i)
        img = torch.from_numpy(stars.astype(np.float32)).to(device)
        img += sky + torch.randn((size, size), generator=g, device=device) * float(np.sqrt(sky / 6.2))
        bad = torch.rand((size, size), generator=g, device=device) < 1e-3
        mask = torch.where(bad, 256, 0).to(torch.int32)
        wgt = torch.where(bad, 0.0, 6.2 / sky).to(torch.float32)
        frames.append(dict(img=img, wgt=wgt, mask=mask, wcs=w,
                           flxscale=10 ** (-0.4 * (magzp - 25.0))))
    return base, frames


def pmc_traffic(args):
    """HBM bytes per k_resample launch from the committed rocprofv3 --pmc passes
    (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is); None when the
    profile does not match this workload."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_resample.json')
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    if d.get('size') != args.size or bool(d.get('mask')) != (not args.no_mask):
        return None
    return d.get('hbm_bytes_per_launch')


def cpu_baseline(synth, size, combine, nframes=8, sub_size=640):
    """CPU restatement timed on this box's host cores (a port, NOT SWarp / hotpants, which
    are not installed): the C / OpenMP port of the oracle (oracle/cport, all cores) on the
    resample -> coadd leg of `nframes` config-2 frames of the full size, plus - reported in
    the sample text only - the numpy hotpants restatement on one small frame."""
    from oracle import cport
    from oracle.wcs import WCS as OWCS

    def ow(w):
        return OWCS(w.crpix, w.crval, w.cd, w.pv1, w.pv2, w.naxis)
    c = cport.load(native=True)
    c.set_threads(int(os.environ.get('ZM_CPU_THREADS', 0)) or cport.host_cores())
    base = synth.ztf_wcs(size, size, tpv=True)
    frames = []
    for i in range(nframes):
        r = np.random.default_rng(2000 + i)
        w = synth.ztf_wcs(size, size, dx=r.uniform(-15, 15), dy=r.uniform(-15, 15),
                          rot_deg=r.uniform(-0.1, 0.1), tpv=True)
        frames.append(synth.make_frame(size, size, 2000 + i, w, nstars=100, nbad=size))
    t0 = time.perf_counter()
    vals, wgts = [], []
    for f in frames:
        px, py = c.positions(ow(base), ow(f['wcs']), size, size)
        o, w_, _ = c.resample(f['img'], f['wgt'], px, py, 3, f['flxscale'], f['mask'])
        vals.append(o)
        wgts.append(w_)
    ref, refw = c.combine(np.array(vals), np.array(wgts), combine)
    t1 = time.perf_counter()
    mpix = nframes * size * size / 1e6
    # one subtraction against a small coadd (numpy hotpants restatement, one core)
    from oracle import hotpants as ohp
    from oracle import resample as oresample
    ss = sub_size
    sbase = synth.ztf_wcs(ss, ss, tpv=True)
    sf = synth.make_frame(ss, ss, 2100, synth.ztf_wcs(ss, ss, dx=2.0, dy=-3.0, tpv=True), nstars=100, nbad=ss)
    spx, spy = oresample.positions(ow(sbase), ow(sf['wcs']), ss, ss)
    sv, sw, _ = oresample.resample(sf['img'], sf['wgt'], spx, spy, oresample.LANCZOS3, 1.0)
    srms = np.where(sw > 0, 1.0 / np.sqrt(np.where(sw > 0, sw, 1)), np.sqrt(50000.0))
    t2 = time.perf_counter()
    ohp.subtract(sv + 150.0, sv + 150.0, srms, srms, (sw <= 0).astype(np.uint8),
                 r=5.0, rss=12.0, nsx=max(ss // 100, 1), nsy=max(ss // 100, 1), ko=2, bgo=0,
                 tu=5e3, iu=5e3, tl=-100.0, il=-100.0)
    t3 = time.perf_counter()
    return {'value': mpix / (t1 - t0), 'unit': 'Mpix/s', 'cores': c.threads(), 'kind': 'port',
            'sample': f'{nframes} frames {size}x{size}: per-pixel TPV inverse map + Lanczos-3 resample '
                      f'(image, variance, mask) + {combine} combine in {t1 - t0:.1f} s, C / OpenMP port of '
                      f'the oracle on {c.threads()} threads ({os.cpu_count()} host CPUs visible; gcc -O3 '
                      f'-march=native); separately the numpy hotpants restatement, one core, one '
                      f'{ss}x{ss} subtraction r=5 ko=2: {t3 - t2:.1f} s = {ss * ss / 1e6 / (t3 - t2):.2f} Mpix/s. '
                      f'CPU restatement, not SWarp / hotpants (not installed)'}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # one rank per GPU over RCCL ('nccl').  Rehearsal on a one-GPU box: ZM_DIST_BACKEND=gloo puts
    # several ranks on the same card (local rank modulo the device count).
    backend = os.environ.get('ZM_DIST_BACKEND', 'nccl')
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f'warning: --gpus {args.gpus} but WORLD_SIZE {world}', file=sys.stderr)

    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    dev = importlib.import_module('zuds-pipeline_amd.device')

    import ctypes as C
    eng = z.Engine(local)
    base, frames = make_device_frames(synth, torch, args.frames + 1, args.size,
                                      2000 + 1000 * rank, device)
    sci = frames.pop()            # the science epoch of configs[2]
    # detector defects of the science frame: 300 clustered 3x3 blobs instead of
    # isolated pixels (a 69 x 69 substamp box must be clean to be usable)
    g = torch.Generator(device='cpu')
    g.manual_seed(77 + rank)
    bx = torch.randint(2, args.size - 2, (300,), generator=g)
    by = torch.randint(2, args.size - 2, (300,), generator=g)
    smask = torch.zeros((args.size, args.size), dtype=torch.int32)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            smask[by + dy, bx + dx] = 256
    sci['mask'] = smask.to(device)
    sci['wgt'] = torch.where(sci['mask'] != 0, 0.0, float(sci['wgt'].max())).to(torch.float32)
    sci_rms = torch.where(sci['wgt'] > 0, 1.0 / torch.sqrt(sci['wgt'].clamp_min(1e-20)),
                          float(np.sqrt(50000.0))).to(torch.float32)
    params = z.coadd_params(combine=args.combine, subtract_back=True,
                            rescale_weights=True)
    dframes = dev.DeviceFrames(frames, device)
    coadd = dev.DeviceCoadd(base, params, device=local, engine=eng, want_mask=not args.no_mask)
    sub = dev.DeviceSubtraction(sci['wcs'], base, device=local, engine=eng,
                                stream=coadd.stream)
    ref_rms = torch.empty_like(coadd.wgt)
    npx = args.size * args.size
    L = eng.L

    sum_type = args.combine.upper() in ('WEIGHTED', 'AVERAGE')
    sharded = None
    if world > 1 and not sum_type:
        # exact CLIPPED / MEDIAN of the 32 N deep stack: row-band exchange (BASELINE config 4)
        par = importlib.import_module('zuds-pipeline_amd.parallel')
        sharded = par.ShardedCoadd(par.HipBackend(base, params, device=local, engine=eng))

    # --no-mask: the reference has no mask coadd; the subtraction still takes a (zero) reference mask
    no_ref_mask = torch.zeros((args.size, args.size), dtype=torch.int32, device=device) if args.no_mask else None

    def step():
        # ScienceCoadd / ReferenceImage.from_images: science + mask coadds
        if sharded is not None:
            img, wgt = sharded.exact(dframes, want_mask=coadd.mask is not None)
            with torch.cuda.stream(coadd.stream):
                coadd.stream.wait_stream(sharded.backend.stream)
                coadd.img.copy_(img)
                coadd.wgt.copy_(wgt)
                if coadd.mask is not None:
                    m = sharded.backend.reduce_mask(cov=coadd.mask_wgt)
                    coadd.stream.wait_stream(sharded.backend.stream)
                    coadd.mask.copy_(m)
        elif world > 1:
            coadd.run_sharded_weighted(dframes)
        else:
            coadd.run(dframes)
        with torch.cuda.stream(coadd.stream):
            if coadd.mask is not None:
                z._lib.check(L.zm_mask_flag_dev(eng.ctx, coadd.mask.data_ptr(),
                                                coadd.mask_wgt.data_ptr(), 0.0, 1 << 16, npx))
            z._lib.check(L.zm_add_scalar_dev(eng.ctx, coadd.img.data_ptr(), 150.0, npx))
            z._lib.check(L.zm_rms_from_weight_dev(eng.ctx, coadd.wgt.data_ptr(), None, npx,
                                                  float(np.sqrt(50000.0)), ref_rms.data_ptr()))
        if not args.no_subtract:
            # SingleEpochSubtraction.from_images with the reference's defaults
            sub.run(sci['img'], sci_rms, sci['mask'], sci['wgt'], coadd.img, ref_rms,
                    coadd.mask if coadd.mask is not None else no_ref_mask, seeing=args.seeing, nreg_side=3)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier(device_ids=[local]) if backend == 'nccl' else dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync()
    # Timed region: only the roofline kernel carries HIP-event timers (two event records per
    # launch cost dispatch latency: 0.7 ms per step with every scope timed).  The per-kernel
    # table comes from one more, untimed-for-throughput step with every scope timed.
    eng.timing(True, only='resample')
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    eng.timing(False)
    rs_ms, rs_cnt = eng.timing_read('resample')
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.timing_reset()
    eng.timing(True)
    step()
    sync()
    eng.timing(False)

    frames_per_step = (args.frames + (0 if args.no_subtract else 1)) * world
    mpix_per_step = frames_per_step * args.size * args.size / 1e6
    value = mpix_per_step * args.steps / dt

    if rank == 0:
        names = ['resample', 'mask_box', 'resample_mask', 'median_mad', 'prep', 'mesh_stats', 'mesh_filter', 'bk_expand',
                 'combine', 'lattice', 'hp_masks', 'hp_cells', 'hp_vectors', 'hp_gram',
                 'hp_solve', 'hp_apply']
        kt = {}
        for nme in names:
            ms, cnt = eng.timing_read(nme)
            if cnt:
                kt[nme] = {'ms_per_step': ms, 'launches_per_step': cnt, 'avg_us': 1e3 * ms / cnt}
        dom = max(kt, key=lambda k: kt[k]['ms_per_step']) if kt else None
        if rs_cnt:      # the roofline kernel: from the timed region itself
            kt['resample'] = {'ms_per_step': rs_ms / args.steps, 'launches_per_step': rs_cnt // args.steps,
                              'avg_us': 1e3 * rs_ms / rs_cnt}
        roofline = None
        if 'resample' in kt:
            avg_s = kt['resample']['avg_us'] * 1e-6
            bpp = RESAMPLE_BYTES_PER_OUTPX + (0 if args.no_mask else MASK_BYTES_PER_OUTPX)
            bytes_per_launch = bpp * args.size * args.size
            ach = bytes_per_launch / avg_s / 1e9
            roofline = {'bound': 'hbm',
                        'kernel': 'k_resample<LANCZOS3, mask fused>' if not args.no_mask
                        else 'k_resample<LANCZOS3>',
                        'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': ach / HBM_PEAK_GBS, 'traffic': pmc_traffic(args),
                        'avg_launch_us': kt['resample']['avg_us'],
                        'algorithmic_bytes_per_launch': bytes_per_launch,
                        'dominant_by_time': dom}
        out = {
            'metric': 'Mpix/s resample->coadd->subtract, 3072x3072 frames',
            'value': value, 'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'configs[1]+[2]: {args.frames}x {args.size}x{args.size} '
                                   f'TPV frames/GPU, mesh background + weight rescale + '
                                   f'Lanczos-3 resample + {args.combine} coadd (+ AND mask coadd)'
                                   + ((', RCCL all-reduce of the partial sums' if sum_type else ', row-band exchange over RCCL') if world > 1 else '')
                                   + ('' if args.no_subtract else
                                      '; then 1 science frame/GPU: align ref, hotpants 3x3 regions '
                                      'x 10x10 stamps, r=10, ko=4, subtract'),
                       'frames_per_gpu': args.frames, 'size': args.size,
                       'combine': args.combine, 'subtract': not args.no_subtract,
                       'hotpants': None if args.no_subtract else
                       {k: getattr(sub.info, k) for k, _ in sub.info._fields_}},
            'kernels': kt,      # one extra step with every scope timed ('resample': the timed region)
            'roofline': roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(synth, args.size, args.combine, args.cpu_frames)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
