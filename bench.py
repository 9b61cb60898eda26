#!/usr/bin/env python
"""bench.py - Mpix/s through resample -> coadd -> subtract on 3072 x 3072 frames.

One step = one pass of the hot path over one batch of synthetic frames already
resident in HBM: BASELINE.json configs[1] (32 ZTF-CCD-sized TPV frames, Lanczos-3
resample + WEIGHTED coadd) followed by configs[2] (one science frame subtracted
against that coadd) once the subtraction kernels are built in.  Pixel accounting
(SURVEY.md section 8(d)): Mpix = (frames resampled + frames subtracted) x 9.437184.

N > 1: one rank per GPU over RCCL, weak scaling - every rank resamples its own 32 frames
of a 32 N deep stack, the partial sums are reduced across ranks, every rank then subtracts
its own frame.  Either the driver starts the ranks (torch.distributed.run: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or `python bench.py --gpus N` starts them itself:
the parent never imports torch nor touches a GPU, it spawns N children of this script with
the rendezvous environment set, relays rank 0's JSON line and exits non-zero when a child
fails (the reference's analogue is `srun -n 64` + get_my_share_of_work,
nersc/controller.py:101, zuds/mpi.py:36-64).
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

# hardware queues for the many-subtractions leg (one process on the GPU: zuds-pipeline_amd/nightly.py);
# read by the HIP runtime at its first call.  Not for N > 1 (no such leg) nor for the rehearsal
# with several ranks on one card, where more queues per process only oversubscribe the hardware.
if int(os.environ.get('WORLD_SIZE', '1')) == 1:
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBS = 6290.0    # ... and what that guide measures for a float4 copy (79 % of the spec)
VALU_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector (packed FMA), spec
# fp64 matrix peak: 256 CUs x 4 SIMDs x one v_mfma_f64_16x16x4_f64 (2 x 16 x 16 x 4 flop) per 64 cycles (the
# instruction cost measured for k_chol_fused) x 2.4 GHz = 78.6 TFLOP/s = AMD's FP64 matrix figure
MFMA_F64_PEAK_TFLOPS = 78.6
RESAMPLE_BYTES_PER_OUTPX = 16  # img+var read, img+var write
MASK_BYTES_PER_OUTPX = 8       # + 4 B in / 4 B out when int32 masks ride along
PMC_PROFILES = ['r06_pmc.json', 'r05_pmc.json']     # newest first; each stamped with the hash of the kernel sources it measured


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
    ap.add_argument('--mask-dtype', default='int16', choices=['int16', 'int32'],
                    help='input masks in HBM: int16 = a ZTF mask as its BITPIX 16 file holds it (zm_dframe.mask_type), '
                         'int32 = the widened copy of rounds 1 - 3')
    ap.add_argument('--seeing', type=float, default=4.0,
                    help='science FWHM in pixels: r = 2.5 seeing, rss = 6 seeing')
    ap.add_argument('--no-secondary', action='store_true', help='skip the CLIPPED secondary line')
    ap.add_argument('--no-clocks', action='store_true', help='skip the PCIe / FITS clocks')
    ap.add_argument('--no-nightly', action='store_true', help='skip the concurrent-subtraction leg')
    ap.add_argument('--no-pipelined', action='store_true', help='skip the software-pipelined rate')
    ap.add_argument('--pipelined-share', type=int, default=1, help='developer: zm_ctx_set_share of the subtraction contexts')
    ap.add_argument('--pipelined-priority', type=int, default=0, help='developer: 1 = the subtraction chains of the pipelined leg on high-priority streams')
    ap.add_argument('--pipelined-depth', type=int, default=4, help='subtractions in flight in the pipelined leg')
    ap.add_argument('--nightly-jobs', type=int, default=32, help='subtractions of the concurrent leg')
    ap.add_argument('--nightly-pools', default='1,2,4,8,16', help='jobs in flight to time in the concurrent leg')
    ap.add_argument('--nightly-batches', default='1x8,1x16,2x8,2x16,3x8,3x11',
                    help='lanes x batch of the batched pools to time in the concurrent leg (SubtractionPool(J, batch=B): '
                         'the kernel fits of B jobs as one chain of launches); empty: none')
    ap.add_argument('--nightly-files', default='2x8x16',
                    help='lanes x fit batch x frames per file batch of the FITS-inclusive nightly clock (scripts/donightly.py)')
    ap.add_argument('--dump-coadd', default=None,
                    help='developer / tests: rank 0 saves the coadd planes [img, wgt] (.npy) after the run')
    ap.add_argument('--emulate-ranks', type=int, default=1,
                    help='developer / tests: ONE process coadds the frames N ranks would hold (their seeds), '
                         'the reference a multi-rank run is compared with')
    ap.add_argument('--cpu-frames', type=int, default=2,
                    help='full-size frames the CPU baseline resamples and coadds')
    return ap.parse_args()


def make_device_frames(synth, torch, n, size, seed0, device, mask_dtype='int16'):
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
        mask = torch.where(bad, 256, 0).to(getattr(torch, mask_dtype))
        wgt = torch.where(bad, 0.0, 6.2 / sky).to(torch.float32)
        frames.append(dict(img=img, wgt=wgt, mask=mask, wcs=w,
                           flxscale=10 ** (-0.4 * (magzp - 25.0))))
    return base, frames


def kernel_sources_sha16():
    """Hash of the kernel sources of this tree (the function of tools/make_pmc_json.py)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'zuds-pipeline_amd', 'csrc')
    for f in sorted(f for f in os.listdir(d) if f.endswith(('.hip', '.h'))):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'zudsmi.h'), 'rb').read())
    return h.hexdigest()[:16]


def pmc_profile(args):
    """The counter profile of this tree's kernels (tools/make_pmc_json.py, one tools/gpu_round.sh call): per
    bench command ('weighted', 'clipped') and kernel the mean of FETCH_SIZE (doubled: gfx950 tallies 128-B
    requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE per launch and the SQ instruction counters.  Counters
    cannot be collected inside a bench run (separate --pmc passes), so they are quoted - but only from a
    profile measured on THESE kernel sources (`kernel_sources_sha16`) at this size / depth / mask type;
    otherwise {'stale': why} and the line carries null for traffic / valu figures."""
    sha = kernel_sources_sha16()
    for name in PMC_PROFILES:
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', name)))
        except (OSError, ValueError):
            continue
        if d.get('kernel_sources_sha16') != sha:
            return {'stale': f'profiles/{name} was measured on kernel sources {d.get("kernel_sources_sha16")}, '
                             f'this tree is {sha}: counter figures not quoted'}
        if d.get('size') != args.size or d.get('frames') != args.frames or \
                d.get('mask_dtype') != (None if args.no_mask else args.mask_dtype):
            return {'stale': f'profiles/{name} was measured at size {d.get("size")} x {d.get("frames")} frames, masks '
                             f'{d.get("mask_dtype")}: not this run'}
        d['file'] = f'profiles/{name}'
        return d
    return {'stale': 'no counter profile committed for this round'}


def pmc_traffic(pmc, section, prefixes):
    """HBM bytes per launch summed over the kernels whose names start with one of `prefixes`, or None."""
    ks = (pmc.get(section) or {}).get('kernels') or {}
    hits = [v for k, v in ks.items() if any(k.startswith(p) for p in prefixes) and v.get('hbm_bytes_per_launch') is not None]
    # (per launch of each kernel x its launches per coadd: the pre-pass kernels run once per stack like the fused one)
    return int(sum(v['hbm_bytes_per_launch'] * v.get('launches_per_coadd', 1) for v in hits)) if hits else None




def hotpants_job(synth, size, nreg_side):
    """A full-size subtraction at the bench's parameters for the CPU baseline: size^2 frame pair, nreg_side^2 regions,
    the stamps, kernel and orders of the step (r = 10, rss = 24, ko = 4: 722 unknowns per region)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(3000)
    ref = np.full((size, size), 150.0)
    ns_ = max(int(3000 * (size / 3072.0) ** 2), 50)
    synth.add_stars(ref, rng.uniform(10, size - 10, ns_), rng.uniform(10, size - 10, ns_),
                    np.exp(rng.uniform(np.log(3e3), np.log(8e4), ns_)), 4.0)
    sci = 1.2 * gaussian_filter(ref, 1.0) + 20.0 + rng.normal(0, 3.0, ref.shape)
    ref = ref + rng.normal(0, 0.5, ref.shape)
    nst = max(int(size / 100.0 / nreg_side), 1)
    kw = dict(r=10.0, rss=24.0, nsx=nst, nsy=nst, nrx=nreg_side, nry=nreg_side, ko=4, bgo=0, tu=5e3, iu=5e3, tl=-100.0, il=-100.0)
    return (sci, ref, np.full(ref.shape, 3.0), np.full(ref.shape, 0.5), np.zeros(ref.shape, np.uint8)), kw, nst


def cpu_baseline(synth, size, combine, nframes=8, steps_frames=32, nreg_side=3):
    """The whole metric on this box's host cores, by a CPU restatement (a port, NOT SWarp / hotpants /
    SExtractor, which are not installed): per frame the mesh background of the image and of its variance
    map (SUBTRACT_BACK Y, RESCALE_WEIGHTS Y), background off, per-pixel TPV inverse map, Lanczos-3 resample
    of image / variance / mask, then the combine - all in the C / OpenMP port of the oracle
    (oracle/cport, every core) on `nframes` full-size frames; and the hotpants restatement in C as well (round 6:
    oracle/cport/zm_hotpants.c, validated against oracle/hotpants.py in tests/test_oracle_cport.py) on ONE full-size
    subtraction at the bench's parameters (3 x 3 regions, r = 10, rss = 24, 10 x 10 stamps, ko = 4: 722 unknowns
    per region).  `value` composes the stage times into the bench step - `steps_frames` frames resampled and
    coadded + one full-size subtraction - with the metric's pixel accounting; every stage is also timed on one thread (SWarp's NTHREADS 1,
    zuds/astromatic/makecoadd/default.swarp:115)."""
    from oracle import cport
    from oracle.wcs import WCS as OWCS

    def ow(w):
        return OWCS(w.crpix, w.crval, w.cd, w.pv1, w.pv2, w.naxis)
    c = cport.load(native=True)
    ncores = int(os.environ.get('ZM_CPU_THREADS', 0)) or cport.host_cores()
    base = synth.ztf_wcs(size, size, tpv=True)
    frames = []
    for i in range(nframes):
        r = np.random.default_rng(2000 + i)
        w = synth.ztf_wcs(size, size, dx=r.uniform(-15, 15), dy=r.uniform(-15, 15),
                          rot_deg=r.uniform(-0.1, 0.1), tpv=True)
        frames.append(synth.make_frame(size, size, 2000 + i, w, nstars=100, nbad=size))

    def coadd_leg(fr):
        t0 = time.perf_counter()
        vals, wgts = [], []
        for f in fr:
            img, wgt = f['img'].astype(np.float64), f['wgt']
            bkg, _, _, bsig, _, _ = c.background(img, wgt, 128, 3)
            with np.errstate(divide='ignore'):
                var = np.where(wgt > 1e-30, 1.0 / np.where(wgt > 0, wgt, 1), 0.0)
            _, _, level, _, _, _ = c.background(var, wgt, 128, 3, want_images=False)
            scale = bsig * bsig / level if (level > 0 and bsig > 0) else 1.0
            px, py = c.positions(ow(base), ow(f['wcs']), size, size)
            o, w_, _ = c.resample((img - bkg).astype(np.float32), (wgt / scale).astype(np.float32), px, py, 3,
                                  f['flxscale'], f['mask'])
            vals.append(o)
            wgts.append(w_)
        t1 = time.perf_counter()
        c.combine(np.array(vals), np.array(wgts), combine)
        return t1 - t0, time.perf_counter() - t1
    c.set_threads(ncores)
    t_fr, t_cb = coadd_leg(frames)
    c.set_threads(1)
    t_fr1, t_cb1 = coadd_leg(frames[:1])
    c.set_threads(ncores)
    # one full-size subtraction (nreg_side^2 regions) in the C restatement: every core, then one thread
    hp_args, hp_kw, nst = hotpants_job(synth, size, nreg_side)
    nreg = nreg_side * nreg_side
    t2 = time.perf_counter()
    _, _, hinfo = c.hotpants(*hp_args, **hp_kw)
    t_sub = time.perf_counter() - t2
    c.set_threads(1)
    t2 = time.perf_counter()
    c.hotpants(*hp_args, only_region=nreg // 2, **hp_kw)          # (one region on one thread, x the regions)
    t_reg1 = time.perf_counter() - t2
    c.set_threads(ncores)
    hreg = [r for r in hinfo['regions'] if r is not None]
    npx = size * size / 1e6
    per_frame, per_frame1 = t_fr / nframes, t_fr1 / 1
    comb = t_cb * steps_frames / nframes                       # one read of every sample: linear in the depth
    comb1 = t_cb1 * steps_frames / 1
    t_sub_one = t_reg1 * nreg
    t_step = steps_frames * per_frame + comb + t_sub
    t_step1 = steps_frames * per_frame1 + comb1 + t_sub_one
    mpix_step = (steps_frames + 1) * npx
    return {'value': mpix_step / t_step, 'unit': 'Mpix/s', 'cores': ncores, 'kind': 'port',
            'value_one_thread': mpix_step / t_step1,
            'stages': {'background_rescale_resample_s_per_frame': per_frame, 'same_on_one_thread': per_frame1,
                       f'combine_{combine}_s_per_{steps_frames}_frames': comb,
                       'hotpants_s_per_subtraction': t_sub, 'hotpants_s_per_region_one_thread': t_reg1,
                       'hotpants_s_per_subtraction_one_thread': t_sub_one,
                       'coadd_leg_mpix_s': steps_frames * npx / (steps_frames * per_frame + comb),
                       'subtract_leg_mpix_s': npx / t_sub,
                       'hotpants': {'regions_solved': len(hreg), 'regions': nreg,
                                    'stamps_used': [r['nstamps_used'] for r in hreg], 'rounds': [r['niter'] for r in hreg],
                                    'unknowns': hreg[0]['ncoeff'] if hreg else None,
                                    'restatement': 'oracle/cport/zm_hotpants.c (C / OpenMP; separable basis convolutions, '
                                                   'Cholesky, direct block convolution), fp64'}},
            'sample': f'all three stages of the metric on {size}x{size} frames: {nframes} frames through mesh background '
                      f'(image + variance map) + weight rescale + per-pixel TPV inverse map + Lanczos-3 resample '
                      f'(image, variance, mask) in {t_fr:.1f} s and their {combine} combine in {t_cb:.1f} s (C / OpenMP port '
                      f'of the oracle, {ncores} threads; gcc -O3 -march=native; {os.cpu_count()} host CPUs visible); '
                      f'one subtraction of {nreg} regions with r=10 rss=24 {nst}x{nst} stamps ko=4 (722 unknowns per region) in '
                      f'{t_sub:.2f} s on {ncores} threads, {t_sub_one:.1f} s on one (C restatement of oracle/hotpants.py)' +
                      f'. value = ({steps_frames} + 1) frames x {npx:.2f} Mpix / ({steps_frames} x per-frame '
                      f'time + combine scaled to {steps_frames} frames + the subtraction) = the bench step on every core; '
                      f'value_one_thread: every stage on 1 thread (SWarp NTHREADS 1). '
                      f'CPU restatement, not SWarp / hotpants / SExtractor (not installed)'}


def launch(args):
    """`--gpus N` without a rendezvous environment: start the N ranks ourselves.  Nothing in
    this process may initialise the GPU (a process that has must not exec or be replaced;
    children are plain subprocesses)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   ZM_BENCH_LAUNCHER='bench.py')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    # rank 0's JSON line goes to stdout; whatever else a child library wrote there (gloo prints
    # its connection banner on stdout) goes to stderr
    for line in out.decode().splitlines():
        print(line, file=sys.stdout if line.startswith('{') else sys.stderr)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f'bench.py: ranks failed (rank, exit code): {bad}', file=sys.stderr)
        return 1
    return 0


def copy_ceiling(z, eng, torch, stream, device):
    """the rate a plain float4 copy kernel reaches on this GPU (read + write), the
    practical ceiling beside the nominal 8 TB/s.  1 GiB each way, 10 launches between two events on the
    engine's stream."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    src.random_()            # (not zeros: the chip clocks higher on trivial data, MI355X_MICROARCH.md DVFS)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(3):
            z._lib.check(eng.L.zm_copy_probe_dev(eng.ctx, src.data_ptr(), dst.data_ptr(), n))
        a.record(stream)
        for _ in range(10):
            z._lib.check(eng.L.zm_copy_probe_dev(eng.ctx, src.data_ptr(), dst.data_ptr(), n))
        b.record(stream)
    b.synchronize()
    ms = a.elapsed_time(b) / 10
    del src, dst
    return {'kernel': 'k_copy4 (float4, four non-temporal loads in flight per thread, non-temporal stores, 8 workgroups per CU)', 'bytes_each_way': n,
            'avg_us': 1e3 * ms, 'GBs_read_plus_write': 2 * n / (ms * 1e-3) / 1e9}


def hotpants_command(paths, r, rss, nsx, nsy, nreg_side, big_rms, tu, tl):
    """The command line zuds/hotpants.py:77-93 composes (flag for flag), on the files in `paths`."""
    return (f'hotpants -inim {paths["sci"]} -hki -n i -c t -tmplim {paths["ref"]} -outim {paths["out"]} '
            f'-tu {tu} -iu {tu} -tl {tl} -il {tl} -r {r} -rss {rss} -tni {paths["ref_rms"]} '
            f'-ini {paths["sci_rms"]} -imi {paths["mask"]} -v 0 -oni {paths["out_rms"]} '
            f'-fin {big_rms} -nsx {nsx / nreg_side} -nsy {nsy / nreg_side} -nrx {nreg_side} -nry {nreg_side} '
            f'-bgo 0 -ko 4')


def sextractor_command(paths):
    """zuds/sextractor.py:67-98 with the keys of zuds/astromatic/sextractor.conf that reach the
    background maps spelled out as flags (BACK_SIZE 128 = BKG_BOX_SIZE, BACK_FILTERSIZE 3, MAP_WEIGHT,
    WEIGHT_THRESH 1e-30); no catalogue, no detection filter file."""
    return (f'sex {paths["img"]} -CHECKIMAGE_TYPE BACKGROUND,BACKGROUND_RMS '
            f'-CHECKIMAGE_NAME {paths["bkg"]},{paths["rms"]} -CATALOG_TYPE NONE -PARAMETERS_NAME {paths["param"]} '
            f'-FILTER N -BACK_SIZE 128 -BACK_FILTERSIZE 3 -WEIGHT_IMAGE {paths["wgt"]} -WEIGHT_TYPE MAP_WEIGHT '
            f'-WEIGHT_THRESH 1e-30 -VERBOSE_TYPE QUIET')


def rel_diff(ref, mine, good, floor):
    rel = np.abs(ref[good].astype(np.float64) - mine[good]) / np.maximum(np.abs(ref[good]), floor)
    return {'median_rel_diff': float(np.median(rel)), 'p99_rel_diff': float(np.percentile(rel, 99)),
            'pixels': int(good.sum())}


def reference_tools(workdir, files, hip_products):
    """when the reference's own binaries exist on this box, run the commands the
    reference would run (zuds/swarp.py:68-78, zuds/hotpants.py:77-93, zuds/sextractor.py:67-98) on
    FITS files, time them and report the pixel disagreement with the HIP products: `swarp` on the
    bench's own frames against the coadd, `hotpants` on an aligned synthetic 1024 x 1024 pair against
    zm_subtract at the same parameters, `sex` on one frame against zm_background.  None of them is
    installed on the pool images seen so far; the probe result is part of the line either way."""
    import shutil
    import subprocess
    found = {t: shutil.which(t) for t in ('swarp', 'hotpants', 'sex')}
    rep = {'found': {k: v for k, v in found.items() if v}, 'probed': sorted(found)}
    z = importlib.import_module('zuds-pipeline_amd')
    if found['hotpants'] or found['sex']:
        try:
            rep.update(tool_probes_small(workdir, found, z, subprocess))
        except Exception as e:                               # noqa: a probe must not fail the bench
            rep['probe_error'] = repr(e)
    if not found['swarp'] or files is None:
        return rep
    try:
        inlist, wlist = os.path.join(workdir, 'images.in'), os.path.join(workdir, 'weight.in')
        open(inlist, 'w').write('\n'.join(files['sci']) + '\n')
        open(wlist, 'w').write('\n'.join(files['wgt']) + '\n')
        out = os.path.join(workdir, 'swarp.coadd.fits')
        # default.swarp of the reference spelled out as flags (zuds/astromatic/makecoadd/default.swarp)
        cmd = (f'swarp @{inlist} -BACK_SIZE 128 -IMAGEOUT_NAME {out} -VMEM_DIR {workdir} '
               f'-RESAMPLE_DIR {workdir} -WEIGHT_IMAGE @{wlist} -WEIGHTOUT_NAME {out[:-5]}.weight.fits '
               f'-COMBINE_TYPE {files["combine"]} -WEIGHT_TYPE MAP_WEIGHT -RESCALE_WEIGHTS Y '
               f'-WEIGHT_THRESH 1e-30 -CLIP_AMPFRAC 0.3 -CLIP_SIGMA 4.0 -CELESTIAL_TYPE NATIVE '
               f'-PROJECTION_TYPE TPV -CENTER_TYPE ALL -PIXELSCALE_TYPE MEDIAN -IMAGE_SIZE 0 '
               f'-RESAMPLE Y -RESAMPLING_TYPE LANCZOS3 -OVERSAMPLING 0 -INTERPOLATE N '
               f'-FSCALASTRO_TYPE FIXED -FSCALE_KEYWORD FLXSCALE -SUBTRACT_BACK Y -BACK_TYPE AUTO '
               f'-BACK_FILTERSIZE 3 -NTHREADS 1 -VERBOSE_TYPE QUIET')
        t0 = time.perf_counter()
        subprocess.check_call(cmd.split())
        dt = time.perf_counter() - t0
        ref = z.fits.read(out)[0]
        mine = hip_products['coadd']
        rep['swarp'] = {'seconds': dt, 'mpix_s': len(files['sci']) * mine.size / 1e6 / dt, 'nthreads': 1,
                        'shape': list(ref.shape), 'hip_shape': list(mine.shape)}
        if ref.shape == mine.shape:
            rep['swarp'].update(rel_diff(ref, mine, (ref != 0) & (mine != 0), 1e-3))
    except Exception as e:                                   # noqa: a probe must not fail the bench
        rep['swarp_error'] = repr(e)
    return rep


def tool_probes_small(workdir, found, z, subprocess):
    """hotpants / sex, if installed, on a synthetic 1024 x 1024 scene against the HIP path at the same settings."""
    from scipy.ndimage import gaussian_filter
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    rep = {}
    n = 1024
    rng = np.random.default_rng(77)
    ref = np.full((n, n), 150.0)
    nst = 1200
    synth.add_stars(ref, rng.uniform(10, n - 10, nst), rng.uniform(10, n - 10, nst),
                    np.exp(rng.uniform(np.log(3e3), np.log(8e4), nst)), 2.0)
    sci = 1.3 * gaussian_filter(ref, 0.9, mode='nearest') + 20.0 + rng.normal(0, 3.0, ref.shape)
    ref = ref + rng.normal(0, 0.5, ref.shape)
    bpm = np.zeros((n, n), np.int16)
    for _ in range(30):
        bx, by = rng.integers(20, n - 20, 2)
        bpm[by:by + 3, bx:bx + 3] = 1
    sci, ref = sci.astype(np.float32), ref.astype(np.float32)
    srms, rrms = np.full((n, n), 3.0, np.float32), np.full((n, n), 0.5, np.float32)
    wcs_hdr = synth.ztf_wcs(n, n, tpv=False).to_header()
    pth = {k: os.path.join(workdir, f'probe.{k}.fits') for k in
           ('sci', 'ref', 'sci_rms', 'ref_rms', 'mask', 'out', 'out_rms', 'img', 'wgt', 'bkg', 'rms')}
    for k, a in (('sci', sci), ('ref', ref), ('sci_rms', srms), ('ref_rms', rrms), ('mask', bpm)):
        z.fits.write(pth[k], a, dict(wcs_hdr, NAXIS1=n, NAXIS2=n))
    eng = z.Engine(0)
    try:
        if found['hotpants']:
            big = float(np.sqrt(50000.0))
            kw = dict(r=10.0, rss=24.0, nsx=10, nsy=10, nrx=1, nry=1, ko=4, bgo=0, tu=1e6, iu=1e6, tl=-1e3, il=-1e3)
            cmd = hotpants_command(pth, kw['r'], kw['rss'], kw['nsx'], kw['nsy'], 1, big, kw['tu'], kw['tl'])
            t0 = time.perf_counter()
            subprocess.check_call(cmd.split())
            dt = time.perf_counter() - t0
            d, _, info = eng.subtract(sci, srms, ref, rrms, (bpm != 0).astype(np.uint8), **kw)
            hp = z.fits.read(pth['out'])[0]
            fill = np.float32(1e-30)
            good = (hp != fill) & (d != fill)
            rep['hotpants'] = {'seconds': dt, 'mpix_s': n * n / 1e6 / dt, 'command': cmd,
                               'fill_pixels_agree': float(np.mean((hp == fill) == (d == fill))),
                               'hip_kernel_sum': info['kernel_sum'], 'hip_stamps_used': info['nstamps_used']}
            # (relative to the magnitudes that enter the difference: |I| + |I - D|, tests/test_subtract_gpu.py)
            scale = np.abs(sci.astype(np.float64)) + np.abs(sci.astype(np.float64) - hp)
            err = np.abs(hp.astype(np.float64) - d)[good] / scale[good]
            rep['hotpants'].update({'median_rel_diff': float(np.median(err)), 'p99_rel_diff': float(np.percentile(err, 99)),
                                    'pixels': int(good.sum())})
        if found['sex']:
            wgt = np.where(bpm != 0, 0.0, 1.0 / 9.0).astype(np.float32)
            z.fits.write(pth['img'], sci, dict(wcs_hdr, NAXIS1=n, NAXIS2=n))
            z.fits.write(pth['wgt'], wgt, dict(wcs_hdr, NAXIS1=n, NAXIS2=n))
            pth['param'] = os.path.join(workdir, 'probe.param')
            open(pth['param'], 'w').write('NUMBER\n')
            cmd = sextractor_command(pth)
            t0 = time.perf_counter()
            subprocess.check_call(cmd.split())
            dt = time.perf_counter() - t0
            bk = eng.background(sci, wgt, mesh=128, filtersize=3, want=('bkg', 'rms'))
            sb, sr = z.fits.read(pth['bkg'])[0], z.fits.read(pth['rms'])[0]
            allpx = np.ones(sb.shape, bool)
            rep['sex'] = {'seconds': dt, 'mpix_s': n * n / 1e6 / dt, 'command': cmd,
                          'background': rel_diff(sb, np.asarray(bk[0]), allpx, 1e-3),
                          'rms': rel_diff(sr, np.asarray(bk[1]), allpx, 1e-3)}
    finally:
        eng.close()
    return rep


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # one rank per GPU over RCCL ('nccl').  Rehearsal on a one-GPU box: ZM_DIST_BACKEND=gloo puts
    # several ranks on the same card (local rank modulo the device count).
    backend = os.environ.get('ZM_DIST_BACKEND', 'nccl')
    ndev = max(torch.cuda.device_count(), 1)
    shared_card = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # ZM_BENCH_FORCE_DIST=1 (developer): a process group of ONE rank with every collective of the
    # multi-GPU step still made - the RCCL calls of an 8-GPU run, rehearsed on a one-GPU box
    multi = world > 1 or bool(os.environ.get('ZM_BENCH_FORCE_DIST'))
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend, rank=rank, world_size=world)
        importlib.import_module('zuds-pipeline_amd.parallel').FORCE_COLLECTIVES = world == 1
    if args.gpus != world and rank == 0 and world > 1:
        print(f'warning: --gpus {args.gpus} but WORLD_SIZE {world}', file=sys.stderr)

    z = importlib.import_module('zuds-pipeline_amd')
    synth = importlib.import_module('zuds-pipeline_amd.synth')
    dev = importlib.import_module('zuds-pipeline_amd.device')
    check = z._lib.check

    eng = z.Engine(local)
    base, frames = make_device_frames(synth, torch, args.frames + 1, args.size,
                                      2000 + 1000 * rank, device, args.mask_dtype)
    sci = frames.pop()            # the science epoch of configs[2]
    for r in range(1, args.emulate_ranks if world == 1 else 1):
        frames += make_device_frames(synth, torch, args.frames + 1, args.size, 2000 + 1000 * r, device,
                                     args.mask_dtype)[1][:-1]
    # detector defects of the science frame: 300 clustered 3x3 blobs instead of
    # isolated pixels (a 69 x 69 substamp box must be clean to be usable)
    g = torch.Generator(device='cpu')
    g.manual_seed(77 + rank)
    bx = torch.randint(2, args.size - 2, (300,), generator=g)
    by = torch.randint(2, args.size - 2, (300,), generator=g)
    smask = torch.zeros((args.size, args.size), dtype=getattr(torch, args.mask_dtype))
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            smask[by + dy, bx + dx] = 256
    sci['mask'] = smask.to(device)
    sci['wgt'] = torch.where(sci['mask'] != 0, 0.0, float(sci['wgt'].max())).to(torch.float32)
    sci['rms'] = torch.where(sci['wgt'] > 0, 1.0 / torch.sqrt(sci['wgt'].clamp_min(1e-20)),
                             float(np.sqrt(50000.0))).to(torch.float32)
    params = z.coadd_params(combine=args.combine, subtract_back=True,
                            rescale_weights=True)
    dframes = dev.DeviceFrames(frames, device)
    coadd = dev.DeviceCoadd(base, params, device=local, engine=eng, want_mask=not args.no_mask)
    sub = dev.DeviceSubtraction(sci['wcs'], base, device=local, engine=eng,
                                stream=coadd.stream, overlap=os.environ.get('ZM_SUB_OVERLAP', '1') != '0')
    ref_rms = torch.empty_like(coadd.wgt)
    npx = args.size * args.size
    L = eng.L
    big_rms = float(np.sqrt(50000.0))
    sub_async = os.environ.get('ZM_SUB_ASYNC', '1') != '0'     # (0: every subtraction waits for its summary, rounds 1 - 5)

    sum_type = args.combine.upper() in ('WEIGHTED', 'AVERAGE')
    sharded = None
    if multi and not sum_type:
        # exact CLIPPED / MEDIAN of the 32 N deep stack: row-band exchange (BASELINE config 4)
        par = importlib.import_module('zuds-pipeline_amd.parallel')
        sharded = par.ShardedCoadd(par.HipBackend(base, params, device=local, engine=eng))

    # --no-mask: the reference has no mask coadd; the subtraction still takes a (zero) reference mask
    no_ref_mask = torch.zeros((args.size, args.size), dtype=torch.int32, device=device) if args.no_mask else None

    def coadd_leg(co, dfr):
        # ScienceCoadd / ReferenceImage.from_images: science + mask coadds, bit 16, pedestal, rms map
        if sharded is not None:
            img, wgt = sharded.exact(dfr, want_mask=co.mask is not None)
            with torch.cuda.stream(co.stream):
                co.stream.wait_stream(sharded.backend.stream)
                co.img.copy_(img)
                co.wgt.copy_(wgt)
                if co.mask is not None:
                    m = sharded.backend.reduce_mask(cov=co.mask_wgt)
                    co.stream.wait_stream(sharded.backend.stream)
                    co.mask.copy_(m)
        elif multi:
            co.run_sharded_weighted(dfr)
        else:
            co.run(dfr)
        with torch.cuda.stream(co.stream):
            if co.mask is not None:
                check(L.zm_mask_flag_dev(eng.ctx, co.mask.data_ptr(), co.mask_wgt.data_ptr(), 0.0, 1 << 16, npx))
            check(L.zm_add_scalar_dev(eng.ctx, co.img.data_ptr(), 150.0, npx))
            check(L.zm_rms_from_weight_dev(eng.ctx, co.wgt.data_ptr(), None, npx, big_rms, ref_rms.data_ptr()))

    def sub_leg(co, sc, resident=None):
        # SingleEpochSubtraction.from_images with the reference's defaults (the tested object:
        # tests/test_device_chain_gpu.py).  resident: the science planes are the bench's resident ones (nothing on any
        # stream still writes them): the background of the science frame may then start at once, beside the coadd
        # (sci_ready=False); planes that are being copied in or decoded (the clocks) are waited for on the stream.
        if resident is None:
            resident = sc is sci
        def one():
            # wait=False (round 6): the call returns when the fit's last round has been seen; the next step's coadd
            # is enqueued while this step's convolution runs (same stream: nothing overlaps on the GPU), the summary
            # of step k is read - and its checks made - when step k + 1 gets here, or by `sync` below
            sub.run(sc['img'], sc['rms'], sc['mask'], sc['wgt'], co.img, ref_rms,
                    co.mask if co.mask is not None else no_ref_mask, seeing=args.seeing, nreg_side=3,
                    sci_ready=False if (sub.overlap and resident) else None, wait=not sub_async)
        if not shared_card:
            return one()
        # rehearsal with several ranks on one card: the fused Cholesky sizes its grid for a GPU of
        # its own, so the ranks of a card take turns
        for r in range(world):
            if r == rank:
                one()
                torch.cuda.synchronize(device)
            dist.barrier()

    def step(co=coadd, dfr=dframes, sc=sci):
        coadd_leg(co, dfr)
        if not args.no_subtract:
            sub_leg(co, sc)

    def sync():
        torch.cuda.synchronize(device)
        sub.result()                                    # (the last subtraction's summary and checks, if it was left pending)
        if multi:
            dist.barrier(device_ids=[local]) if backend == 'nccl' else dist.barrier()
            torch.cuda.synchronize(device)

    local_dt = {}

    def timed(fn, n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        dt = time.perf_counter() - t0
        local_dt['last'] = dt              # this rank's own clock (the line carries the maximum over the ranks)
        if multi:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if os.environ.get('ZM_BENCH_TRACE'):               # developer: per-step wall times from a cold start
        for i in range(30):
            sync()
            t0 = time.perf_counter()
            step()
            sync()
            print(f'step {i}: {1e3 * (time.perf_counter() - t0):.2f} ms', file=sys.stderr)
    if os.environ.get('ZM_BENCH_HOSTPROBE'):           # developer: how far ahead of the GPU the host enqueues a step
        for i in range(12):
            sync()
            t0 = time.perf_counter()
            coadd_leg(coadd, dframes)
            t1 = time.perf_counter()
            sub_leg(coadd, sci)
            t2 = time.perf_counter()
            sync()
            t3 = time.perf_counter()
            print(f'step {i}: coadd enqueued in {1e3 * (t1 - t0):.3f} ms, subtraction returned after '
                  f'{1e3 * (t2 - t0):.3f} ms, drained after {1e3 * (t3 - t0):.3f} ms', file=sys.stderr)
    # Runtime spin-up, before the W warm-up steps the contract asks for: a fresh process stalls
    # once, for ~40 ms, at its fourth step (HIP runtime state that is set up lazily - seen with
    # ZM_BENCH_TRACE=1: 15, 11, 11, 53, 11, 11, ... ms); with a small W that stall would land in
    # the timed region.  Untimed, like the allocation of the inputs above.
    for _ in range(6):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    # Timed region: only the roofline kernel carries HIP-event timers (two event records per
    # launch cost dispatch latency: 0.7 ms per step with every scope timed).  The per-kernel
    # table comes from one more, untimed-for-throughput step with every scope timed.
    # WEIGHTED / AVERAGE stacks run the fused kernel (frames looped inside the output tile);
    # CLIPPED / MEDIAN materialise the stack through the same kernel in STACK mode (ZM_COADD_FUSED=0:
    # k_resample frame by frame)
    fused = os.environ.get('ZM_COADD_FUSED', '1') != '0'
    roof_scope = 'coadd_fused' if fused else 'resample'
    eng.timing(True, only=roof_scope)
    eng.timing_reset()
    dt = timed(step, args.steps)
    my_ms_per_step = 1e3 * local_dt['last'] / args.steps
    eng.timing(False)
    rs_ms, rs_cnt = eng.timing_read(roof_scope)
    eng.timing_reset()
    eng.timing(True)
    step()
    sync()
    eng.timing(False)
    names = ['coadd_fused', 'ff_headers', 'resample', 'mask_box', 'resample_mask', 'median_mad', 'prep', 'mesh_stats', 'mesh_filter',
             'bk_expand', 'bk_rows', 'combine', 'lattice', 'hp_masks', 'hp_cells', 'hp_vectors', 'hp_gram',
             'hp_solve', 'hp_chol', 'hp_apply']
    kt = {}
    for nme in names:
        ms, cnt = eng.timing_read(nme)
        if cnt:
            kt[nme] = {'ms_per_step': ms, 'launches_per_step': cnt, 'avg_us': 1e3 * ms / cnt}
    if sub.info.status != 0 and not args.no_subtract:
        print(f'rank {rank}: subtraction status {sub.info.status}', file=sys.stderr)
        sys.exit(3)

    frames_per_step = (args.frames + (0 if args.no_subtract else 1)) * world
    mpix_per_step = frames_per_step * args.size * args.size / 1e6
    value = mpix_per_step * args.steps / dt

    # per-leg rates (same objects, same inputs; each leg bracketed by its own sync)
    legs = {}
    dt_c = timed(lambda: coadd_leg(coadd, dframes), args.steps)
    legs['coadd_ms'] = 1e3 * dt_c / args.steps
    legs['coadd_mpix_s'] = args.frames * world * npx / 1e6 * args.steps / dt_c
    if not args.no_subtract:
        dt_s = timed(lambda: sub_leg(coadd, sci), args.steps)
        legs['subtract_ms'] = 1e3 * dt_s / args.steps
        legs['subtract_mpix_s'] = world * npx / 1e6 * args.steps / dt_s

    # who ran where: the world RCCL / gloo actually formed
    me = {'rank': rank, 'device': local, 'name': torch.cuda.get_device_name(local), 'pid': os.getpid(),
          'ms_per_step': my_ms_per_step}
    if multi:
        # one more step with the exchange steps of this rank on a wall clock of their own (parallel.PROBE: each
        # bracketed by device synchronisations - never inside the timed region): what a first run on a real
        # multi-GPU node needs to explain its scaling
        par_mod = importlib.import_module('zuds-pipeline_amd.parallel')
        par_mod.PROBE = {}
        try:
            coadd_leg(coadd, dframes)
            sync()
        finally:
            me['exchange_ms'] = {k: 1e3 * v for k, v in par_mod.PROBE.items()}
            par_mod.PROBE = None
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    secondary = None
    clocks = None
    tools = None
    pipelined = None
    # (round 5: the PCIe / FITS clocks run FIRST of the extra legs.  Behind the pipelined and nightly legs - dozens of
    # engines, each with two streams, all mapped onto GPU_MAX_HW_QUEUES hardware queues - the copy stream created here
    # shared a queue with a compute stream and the H2D copies of step k + 1 queued behind the kernels of step k:
    # with_pcie_ms 56 -> 69 ms in the round-4 line, 1.00 x the copy alone again when measured on a fresh process.)
    order = os.environ.get('ZM_BENCH_ORDER', 'pipelined,clocks').split(',')
    for leg in order:
        if leg == 'clocks' and world == 1 and rank == 0 and not args.no_clocks:
            clocks, tools = data_movement_clocks(args, z, dev, eng, torch, base, frames, sci, coadd, sub,
                                                 ref_rms, step, timed, 1e3 * dt / args.steps)
        if leg == 'pipelined' and world == 1 and rank == 0 and not args.no_subtract and not args.no_pipelined and sum_type:
            pipelined = pipelined_leg(args, z, dev, torch, base, dframes, sci, coadd, eng, no_ref_mask, npx, local, main_sub=sub)
    nightly = None
    if world == 1 and rank == 0 and not args.no_subtract and not args.no_nightly:
        nightly = nightly_leg(args, z, torch, base, frames, coadd, ref_rms, no_ref_mask, npx, local)
    if world == 1 and rank == 0:
        if sum_type and not args.no_secondary:
            secondary = secondary_clipped(args, z, dev, eng, base, dframes, local, timed, npx,
                                          full_step=None if args.no_subtract else (coadd_leg, sub_leg, sci), stream=coadd.stream)

    if rank == 0 and args.dump_coadd:
        torch.cuda.synchronize(device)
        np.save(args.dump_coadd, torch.stack([coadd.img, coadd.wgt]).cpu().numpy())
    if rank == 0:
        dom = max((k for k in kt if k != 'hp_chol'), key=lambda k: kt[k]['ms_per_step']) if kt else None   # (hp_chol lies inside hp_solve)
        if rs_cnt:      # the roofline kernel: from the timed region itself
            kt[roof_scope] = {'ms_per_step': rs_ms / args.steps, 'launches_per_step': rs_cnt // args.steps,
                              'avg_us': 1e3 * rs_ms / rs_cnt}
            if not fused:
                # (the 'resample' scope also holds the two align launches of the subtraction)
                pass
        roofline = None
        pmc = pmc_profile(args)
        # which form of the fused kernel the launcher chose for this stack (zm_ctx_query: the owner-staged one where
        # every planned footprint fits its fixed slot, else the LDS-DMA staged one)
        ff_kernel = {1: 'k_coadd_fused_dma', 2: 'k_coadd_fused_own'}.get(eng.query('fused_form'), 'k_coadd_fused_own')
        m = 0 if args.no_mask else 1
        mask_in = 0 if not m else (2 if args.mask_dtype == 'int16' else 4)      # bytes per input pixel of a mask plane
        if roof_scope in kt:
            avg_s = kt[roof_scope]['avg_us'] * 1e-6
            if fused and sum_type:
                # What THIS kernel moves: per input pixel image + weight (8 B) + the 2-byte
                # box-OR entry of the mask (the mask words themselves are read by k_mask_box_rows, its own
                # launch); per output pixel coadd + weight (+ int32 mask coadd).  SURVEY.md 8(d)'s fused figure.
                bytes_per_launch = (args.frames * (8 + 2 * m) + (8 + 4 * m)) * npx
                kname = ff_kernel + '<LANCZOS3' + (', mask coadd>' if m else '>')
                units = f'{args.frames} frames x {args.size}^2 px per launch'
            elif fused:
                # the materialised stack out of the same kernel (STACK mode): 8 B in (+ 2 B box-OR) per input
                # pixel, 8 B {value, weight} out per output pixel and frame, + 4 B partial mask coadd per stack
                bytes_per_launch = (args.frames * (16 + 2 * m) + 4 * m) * npx
                kname = ff_kernel + '<LANCZOS3, stack' + (', mask coadd>' if m else '>')
                units = f'{args.frames} frames x {args.size}^2 px per launch'
            else:
                bytes_per_launch = (RESAMPLE_BYTES_PER_OUTPX + MASK_BYTES_PER_OUTPX * m) * npx
                kname = 'k_resample<LANCZOS3' + (', mask fused>' if m else '>')
                units = f'1 frame x {args.size}^2 px per launch'
            ach = bytes_per_launch / avg_s / 1e9
            section = 'weighted' if sum_type else 'clipped'
            kern_pmc = (pmc.get(section) or {}).get('kernels') or {}
            kp = (kern_pmc.get('k_coadd_fused') or kern_pmc.get('k_coadd_fused_dma')) if fused else None
            roofline = {'bound': 'hbm', 'kernel': kname, 'units_per_launch': units,
                        'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': ach / HBM_PEAK_GBS, 'traffic': kp.get('hbm_bytes_per_launch') if kp else None,
                        'frac_of_achievable': ach / HBM_ACHIEVABLE_GBS, 'achievable': HBM_ACHIEVABLE_GBS,
                        'avg_launch_us': kt[roof_scope]['avg_us'],
                        'us_per_frame': kt[roof_scope]['avg_us'] / (args.frames if fused else 1),
                        'algorithmic_bytes_per_launch': bytes_per_launch,
                        'algorithmic_bytes': f'per input px: image + weight 8 B{" + box-OR entry 2 B" if m else ""}; per output px: '
                                             f'coadd + weight 8 B{" + int32 mask coadd 4 B" if m else ""} - what this launch reads and writes',
                        'dominant_by_time': dom,
                        'counters_from': pmc.get('stale') or f'{pmc.get("file")} (kernel sources {pmc.get("kernel_sources_sha16")})'}
            if kp:
                # vector issue, not HBM, bounds this kernel: wave-instructions per output pixel and frame
                roofline['traffic_over_algorithmic'] = roofline['wasted'] = kp['hbm_bytes_per_launch'] / bytes_per_launch
                roofline['valu_insts_per_px'] = kp.get('valu_insts_per_px')
                roofline['lds_insts_per_px'] = kp.get('lds_insts_per_px')
                roofline['valu_busy_frac'] = kp.get('valu_busy_frac')
                # what BINDS this kernel, beside the HBM fraction it is priced against - the share of
                # its wave cycles in which a SIMD issues a vector instruction (SQ_ACTIVE_INST_VALU x 4 waves /
                # SQ_WAVE_CYCLES); the exact 6 x 6 two-plane filter in fp32 leaves no way to 0.60 of HBM
                roofline['valu_issue_frac'] = kp.get('valu_busy_frac')
                roofline['binding'] = 'vector issue (valu_issue_frac), not HBM (frac)'
            if fused and sum_type and 'coadd_ms' in legs:
                # the whole coadd leg against the same roof: + the estimation read of the mesh statistics (image +
                # weight, 8 B per input pixel) and the mask words the box pre-pass reads
                leg_bytes = (args.frames * (8 + 8 + mask_in + 0) + (8 + 4 * m)) * npx
                leg_ach = leg_bytes / (legs['coadd_ms'] * 1e-3) / 1e9
                roofline['leg'] = {'what': f'coadd leg: mesh statistics (8 B / input px) + box-OR pre-pass ({mask_in} B / input px of mask) + '
                                           f'fused resample -> coadd (8 B / input px + products)',
                                   'algorithmic_bytes': leg_bytes, 'ms': legs['coadd_ms'], 'achieved': leg_ach,
                                   'leg_frac': leg_ach / HBM_PEAK_GBS,
                                   'leg_traffic': pmc_traffic(pmc, 'weighted', ('k_mesh_stats', 'k_mask_box', 'k_coadd_fused'))}
            if world == 1:
                try:
                    cc = copy_ceiling(z, eng, torch, coadd.stream, device)
                    roofline['copy_ceiling'] = cc
                    roofline['frac_of_copy_ceiling'] = ach / cc['GBs_read_plus_write']
                    cc['guide_float4_copy_GBs'] = HBM_ACHIEVABLE_GBS
                except Exception as e:                       # noqa: a probe must not fail the bench
                    roofline['copy_ceiling_error'] = repr(e)
        # the dominant scope by time is the kernel fit's solver: the fused Cholesky against the fp64 matrix rate
        solve_roof = None
        if 'hp_chol' in kt and not args.no_subtract:
            nunk, nreg = int(sub.info.ncoeff), 9
            # (the launches of a step: one per rejection round that ran + the one enqueued ahead of the host that
            # found nothing to do and returned at its guard - the factorisation's own time is per round that ran)
            nl, niter = kt['hp_chol']['launches_per_step'], max(int(sub.info.niter), 1)
            us = kt['hp_chol']['ms_per_step'] * 1e3 / min(nl, niter)
            flop = nreg * nunk ** 3 / 3.0
            form = os.environ.get('ZM_CHOL_FORM', 'df')
            kname = {'df': 'k_chol_df', 'lat': 'k_chol_fused', 'tp': 'k_chol_tp'}.get(form, 'k_chol_df')
            kc = ((pmc.get('weighted') or {}).get('kernels') or {}).get(kname)
            solve_roof = {'bound': 'mfma', 'kernel': kname, 'avg_us': us, 'launches_per_step': nl, 'rounds': niter,
                          'flop_per_launch': flop, 'what': f'{nreg} Cholesky factorisations of {nunk}^2 (n^3 / 3 each) per launch, fp64',
                          'achieved': flop / (us * 1e-6) / 1e12, 'peak': MFMA_F64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': flop / (us * 1e-6) / 1e12 / MFMA_F64_PEAK_TFLOPS,
                          'traffic': kc.get('hbm_bytes_per_launch') if kc else None,
                          'traffic_note': 'HBM bytes per launch by the counters; the algorithmic minimum is the unscaled matrix in and the '
                                          f'factor out: {2 * nreg * nunk * nunk * 8} B',
                          'wasted': (kc['hbm_bytes_per_launch'] / (2.0 * nreg * nunk * nunk * 8)) if kc and kc.get('hbm_bytes_per_launch') else None,
                          'note': 'latency-bound: 23 dependent block steps (diagonal factor by one wave, panel chains, '
                                  'one hand-over between workgroups per 64 columns); tiles stay in LDS'}
        # SURVEY.md 8(d), the exception to the HBM bound: the convolution of the subtraction (25 B and
        # 2 * 2 * (2r + 1)^2 flop per pixel) against both the HBM peak and the fp32 vector peak
        apply_roof = None
        if 'hp_apply' in kt and not args.no_subtract:
            us = kt['hp_apply']['avg_us']
            hw = int(2.5 * args.seeing)
            flop = 4.0 * (2 * hw + 1) ** 2 * npx
            apply_roof = {'kernel': f'k_hp_apply_w<{hw}> (+ k_hp_kbasis, k_hp_ktable: the block kernels)', 'avg_us': us, 'algorithmic_bytes': 25 * npx,
                          'achieved_GBs': 25 * npx / (us * 1e-6) / 1e9, 'hbm_frac': 25 * npx / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          'flop': flop, 'achieved_TFLOPs': flop / (us * 1e-6) / 1e12,
                          'valu_frac': flop / (us * 1e-6) / 1e12 / VALU_F32_PEAK_TFLOPS,
                          'valu_peak_TFLOPs': VALU_F32_PEAK_TFLOPS,
                          'what': 'template and template variance convolved with the spatially varying kernel, '
                                  '(2r+1)^2 taps each; bound by fp32 vector issue, not HBM (SURVEY 8(d))'}
        out = {
            'metric': 'Mpix/s resample->coadd->subtract, 3072x3072 frames',
            'value': value, 'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'configs[1]+[2]: {args.frames}x {args.size}x{args.size} '
                                   f'TPV frames/GPU, mesh background + weight rescale + '
                                   f'Lanczos-3 resample + {args.combine} coadd (+ AND mask coadd, {args.mask_dtype} masks)'
                                   + ((', RCCL reduce of the partial sums' if sum_type else ', row-band exchange over RCCL') if world > 1 else '')
                                   + ('' if args.no_subtract else
                                      '; then 1 science frame/GPU: align ref, hotpants 3x3 regions '
                                      'x 10x10 stamps, r=10, ko=4, subtract'),
                       'frames_per_gpu': args.frames, 'size': args.size,
                       'mask_dtype': None if args.no_mask else args.mask_dtype,
                       'combine': args.combine, 'subtract': not args.no_subtract,
                       'hotpants': None if args.no_subtract else
                       {k: getattr(sub.info, k) for k, _ in sub.info._fields_}},
            'world': {'backend': backend if multi else None, 'native_rccl': os.environ.get('ZM_NATIVE_RCCL') == '1', 'world_size': world,
                      'launcher': os.environ.get('ZM_BENCH_LAUNCHER', 'external' if world > 1 else 'none'),
                      'ranks': ranks},
            'legs': legs,
            'solve_roofline': solve_roof,
            'apply_roofline': apply_roof,
            'kernels': kt,      # one extra step with every scope timed ('resample': the timed region)
            'roofline': roofline,
        }
        if nightly is not None:
            out['nightly'] = nightly
        if pipelined is not None:
            out['pipelined'] = pipelined
        if secondary is not None:
            out['clipped'] = secondary          # (rounds 2 - 3 called it `secondary`)
        if clocks is not None:
            out['clocks'] = clocks
        if tools is not None:
            out['reference_tools'] = tools
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(synth, args.size, args.combine, args.cpu_frames, args.frames)
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


def pipelined_leg(args, z, dev, torch, base, dframes, sci, coadd, eng, no_ref_mask, npx, local, main_sub=None):
    """The same step with consecutive steps software-pipelined, reported beside `value`, never as
    it: the subtractions of steps k, k - 1, ... (D contexts, each with its own stream and host
    thread) run while the coadd of step k + 1 is computed - step k + 1's coadd does not depend on
    step k's subtraction (another field, another quadrant: BASELINE config 5).  Every step still
    makes one full coadd and one full subtraction against THAT coadd (its products are snapshotted
    into one of D + 1 buffer sets; events order the streams).  With D >= 2 the subtraction contexts
    use the one-workgroup-per-region form of the kernel fit's factorisation (zm_ctx_set_share), which
    leaves the CUs to the coadd kernels; D = 1 keeps the many-workgroup form."""
    from concurrent.futures import ThreadPoolExecutor
    check = z._lib.check
    D = max(1, args.pipelined_depth)
    L = eng.L
    big = float(np.sqrt(50000.0))
    A = coadd.stream
    engs, subs = [], []
    pool = ThreadPoolExecutor(max_workers=D)
    try:
        if D >= 2:
            eng.set_share(D + 1)                     # the coadd context too: its fused kernel yields CU slots
        for _ in range(D):
            st = torch.cuda.Stream(coadd.device, priority=-1 if args.pipelined_priority else 0)
            e = z.Engine(local, stream=st.cuda_stream)
            e.set_share(max(args.pipelined_share, D))
            engs.append(e)
            subs.append(dev.DeviceSubtraction(sci['wcs'], base, device=local, engine=e, stream=st))
        nset = D + 1
        snap = [dict(img=torch.empty_like(coadd.img), rms=torch.empty_like(coadd.img),
                     mask=torch.empty((args.size, args.size), dtype=torch.int32, device=coadd.img.device))
                for _ in range(nset)]
        ready = [torch.cuda.Event() for _ in range(nset)]
        freed = [None] * nset
        futs = {}

        def enqueue_coadd(k):
            s = snap[k % nset]
            if freed[k % nset] is not None:
                A.wait_event(freed[k % nset])        # the subtraction that read this buffer set is done
            coadd.run(dframes)
            with torch.cuda.stream(A):
                if coadd.mask is not None:
                    check(L.zm_mask_flag_dev(eng.ctx, coadd.mask.data_ptr(), coadd.mask_wgt.data_ptr(), 0.0, 1 << 16, npx))
                    s['mask'].copy_(coadd.mask)
                else:
                    s['mask'].copy_(no_ref_mask)
                check(L.zm_add_scalar_dev(eng.ctx, coadd.img.data_ptr(), 150.0, npx))
                check(L.zm_rms_from_weight_dev(eng.ctx, coadd.wgt.data_ptr(), None, npx, big, s['rms'].data_ptr()))
                s['img'].copy_(coadd.img)
                ready[k % nset].record(A)

        def subtract(k):
            torch.cuda.set_device(local)
            s, sub = snap[k % nset], subs[k % D]
            sub.stream.wait_event(ready[k % nset])
            sub.run(sci['img'], sci['rms'], sci['mask'], sci['wgt'], s['img'], s['rms'], s['mask'],
                    seeing=args.seeing, nreg_side=3)
            ev = torch.cuda.Event()
            ev.record(sub.stream)
            freed[k % nset] = ev
            return sub.info.status

        state = {'k': 0, 'bad': 0}
        enqueue_coadd(0)

        def pstep():
            k = state['k']
            if k - D in futs:                         # context k % D is free again
                state['bad'] += futs.pop(k - D).result() != 0
            enqueue_coadd(k + 1)
            futs[k] = pool.submit(subtract, k)
            state['k'] = k + 1

        def drain():
            for k in sorted(futs):
                state['bad'] += futs.pop(k).result() != 0
            torch.cuda.synchronize()

        for _ in range(3 + D):
            pstep()
        drain()
        t0 = time.perf_counter()
        # (at least 20 steps: the drain at the end is D subtractions deep and is shared by the steps timed - with the
        # five steps of a default run the leg read 6.0 - 6.2 ms where twenty give 5.5 - 5.6)
        nsteps = max(args.steps, 20)
        for _ in range(nsteps):
            pstep()
        drain()
        dt = time.perf_counter() - t0
        ok = state['bad'] == 0
        subs.clear()
    finally:
        pool.shutdown(wait=True)
        eng.set_share(1)
        eng.set_stream(A.cuda_stream)
        for e in engs:
            e.close()
    return {'ms_per_step': 1e3 * dt / nsteps,
            'mpix_s': (args.frames + 1) * npx / 1e6 * nsteps / dt, 'steps': nsteps, 'status_ok': ok,
            'subtractions_in_flight': D,
            'what': 'steps software-pipelined: the subtractions of steps k, k - 1, ... beside the coadd of step k + 1 '
                    '(one stream, context and host thread per subtraction in flight, D + 1 sets of coadd products); '
                    'one full coadd and one full subtraction against it per step'}


def nightly_leg(args, z, torch, base, frames, coadd, ref_rms, no_ref_mask, npx, local):
    """BASELINE config 5 in small: the epochs of the stack subtracted against its coadd, J at a
    time on one GPU (nightly.SubtractionPool: J engines / streams / host threads; the reference
    runs one process per job, nersc/controller.py:101), forced r = 3 px photometry at 500 fixed
    sky positions on every difference image (scripts/dophot.py:94-156).  Same products for every J
    (tests/test_nightly_gpu.py)."""
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    njobs = min(args.nightly_jobs, len(frames))
    rng = np.random.default_rng(5)
    ra, dec = base.all_pix2world(rng.uniform(50, args.size - 50, 500), rng.uniform(50, args.size - 50, 500), 0)
    ref = dict(img=coadd.img, rms=ref_rms, mask=coadd.mask if coadd.mask is not None else no_ref_mask,
               wcs=base, flxscale=1.0)
    big = float(np.sqrt(50000.0))
    jobs = []
    g = torch.Generator(device='cpu')
    for i, f in enumerate(frames[:njobs]):
        # detector defects as on the science frame of the main step: 300 clustered 3 x 3 blobs
        # (isolated bad pixels at 1e-3 would leave no clean 69 x 69 substamp box)
        g.manual_seed(177 + i)
        bx = torch.randint(2, args.size - 2, (300,), generator=g)
        by = torch.randint(2, args.size - 2, (300,), generator=g)
        m = torch.zeros((args.size, args.size), dtype=torch.int32)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                m[by + dy, bx + dx] = 256
        m = m.to(f['img'].device)
        wgt = torch.where(m != 0, 0.0, float(f['wgt'].max())).to(torch.float32)
        rms = torch.where(wgt > 0, 1.0 / torch.sqrt(wgt.clamp_min(1e-20)), big).to(torch.float32)
        sci = dict(img=f['img'], rms=rms, mask=m, wgt=wgt, wcs=f['wcs'], seeing=args.seeing)
        jobs.append(nm.SubtractionJob(sci, ref, radec=(ra, dec), nreg_side=3))
    out = {'jobs': njobs, 'photometry_positions': 500, 'pools': {}, 'batched': {}}

    def timed_pool(J, batch):
        pool = nm.SubtractionPool(J, device=local, batch=batch)
        shape = {'lanes': pool.njobs, 'fit_batch': pool.batch}       # (J in flight without a batch: the pool picks lanes x batch)
        try:
            pool.map(jobs[:min(max(J * max(batch, 1), 1), len(jobs))], keep=False)      # allocations, code objects
            torch.cuda.synchronize()
            reps = []
            for _ in range(2):                             # (host threads: the faster of two passes)
                t0 = time.perf_counter()
                res = pool.map(jobs, keep=False)
                torch.cuda.synchronize()
                reps.append(time.perf_counter() - t0)
            dt = min(reps)
        finally:
            pool.close()
        # (a job that raised comes back as {'tag', 'error'} without 'info': nightly.SubtractionPool._run)
        bad = [r for r in res if 'error' in r or r['info']['status'] != 0]
        rec = {'ms_per_subtraction': 1e3 * dt / njobs, 'subtract_mpix_s': njobs * npx / 1e6 / dt, **shape,
               'passes_ms': [1e3 * t / njobs for t in reps], 'failed': len(bad)}
        errs = [r['error'] for r in bad if 'error' in r]
        if errs:
            rec['first_error'] = str(errs[0])[:200]
        return rec

    for J in [int(v) for v in args.nightly_pools.split(',')]:
        if J <= njobs:
            out['pools'][str(J)] = timed_pool(J, 0)
    # the batched form: `lanes x batch` - the fits of `batch` jobs are one chain of launches (zm_subtract_batch_dev)
    for spec in [v for v in args.nightly_batches.split(',') if v]:
        J, B = (int(v) for v in spec.split('x'))
        if B <= njobs:
            out['batched'][spec] = timed_pool(J, B)
    if out['batched']:
        bb = max(out['batched'].items(), key=lambda kv: kv[1]['subtract_mpix_s'])
        out['batched_best'] = {'lanes_x_batch': bb[0], 'ms_per_subtraction': bb[1]['ms_per_subtraction'],
                               'over_one_worker': out['pools']['1']['ms_per_subtraction'] / bb[1]['ms_per_subtraction']
                               if '1' in out['pools'] else None}
    best = max(list(out['pools'].values()) + list(out['batched'].values()), key=lambda v: v['subtract_mpix_s'])
    out['subtract_mpix_s'] = best['subtract_mpix_s']
    if not args.no_clocks:
        try:
            out['with_fits'] = nightly_files_clock(args, z, torch, base, jobs, ref, (ra, dec), npx, local)
        except Exception as e:                               # noqa: report, do not fail the bench
            out['with_fits'] = {'error': repr(e)}
    return out


def nightly_files_clock(args, z, torch, base, jobs, ref, radec, npx, local):
    """The same night on the clock a user of scripts/donightly.py lives on: science frames, masks and weight maps as
    FITS files in, three products + a photometry table per subtraction out, through the driver's own loop
    (`donightly.run_night`: the ring of fitsring.FITSRing reads batch b + 1 and writes batch b - 1 while the pool
    subtracts batch b).  One untimed pass (allocations, pinned rings), the products removed, one timed pass."""
    import importlib.util
    import shutil
    import tempfile
    dev = importlib.import_module('zuds-pipeline_amd.device')
    ringmod = importlib.import_module('zuds-pipeline_amd.fitsring')
    nm = importlib.import_module('zuds-pipeline_amd.nightly')
    spec = importlib.util.spec_from_file_location('donightly', os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                            'scripts', 'donightly.py'))
    script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(script)
    lanes, fitb, batch = (int(v) for v in args.nightly_files.split('x'))
    d = tempfile.mkdtemp(prefix='zmnight_', dir=os.environ.get('TMPDIR') or None)
    ring = pool = None
    try:
        need = len(jobs) * npx * (4 + 4 + 2 + 2 * 12)
        if shutil.disk_usage(d).free < 1.5 * need:
            raise OSError(f'not enough free space under {d} for {need / 1e9:.1f} GB of FITS files')
        ring = ringmod.FITSRing(local)
        imgs = []
        for i, job in enumerate(jobs):
            sc = job.sci
            hdr = dict(sc['wcs'].to_header(), NAXIS1=args.size, NAXIS2=args.size, MAGZP=26.0, SEEING=float(sc['seeing']),
                       OBSJD=2458800.5 + 0.25 * i, FIELDID=651, CCDID=3, QID=1, FILTERID=1)
            fn = os.path.join(d, f'ztf_n{i:03d}_000651_zg_c03_o_q1_sciimg.fits')
            ring.save(fn, sc['img'], hdr)
            ring.save(fn.replace('sciimg', 'mskimg'), sc['mask'], hdr, bitpix=16)       # (a ZTF mask file: BITPIX 16)
            ring.save(fn.replace('.fits', '.weight.fits'), sc['wgt'], hdr)
            imgs.append(fn)
        ring.flush()
        # a night is longer than 32 frames: the same files under second names (hard links - no more bytes on disk, the
        # same bytes through the page cache, PCIe and the kernels), so that two of four batches run in steady state
        for i, fn in enumerate(list(imgs)):
            g = os.path.join(d, f'ztf_n{len(jobs) + i:03d}_000651_zg_c03_o_q1_sciimg.fits')
            for a, b in zip(script.science_files(fn), script.science_files(g)):
                os.link(a[0], b[0])
            imgs.append(g)
        refname = os.path.join(d, 'ref.000651_c03_q1_zg.fits')
        rh = dict(base.to_header(), NAXIS1=args.size, NAXIS2=args.size, MAGZP=25.0)
        ring.save(refname, ref['img'], rh)
        wgt = torch.where(ref['rms'] < 200.0, 1.0 / (ref['rms'] * ref['rms']), torch.zeros_like(ref['rms']))
        ring.save(refname.replace('.fits', '.weight.fits'), wgt, rh)
        ring.save(refname.replace('.fits', '.mask.fits'), ref['mask'], rh)
        ring.flush()
        io = dev.FITSDeviceIO(local, engine=z.Engine(local))
        rf = script.load_reference(io, refname)
        pool = nm.SubtractionPool(lanes, device=local, batch=fitb)

        def night():
            t0 = time.perf_counter()
            done = script.run_night(imgs, rf, pool, io, ring, radec, batch=batch, nreg_side=3)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, done
        import contextlib
        import io as _io
        with contextlib.redirect_stdout(_io.StringIO()) as log:
            _, done = night()
            for out in done:
                for sfx in ('.fits', '.rms.fits', '.mask.fits', '.phot.txt'):
                    os.remove(out.replace('.fits', sfx))
            dt, done = night()
        bytes_in = sum(os.path.getsize(p) for fn in imgs for p, _ in script.science_files(fn))
        bytes_out = sum(os.path.getsize(out.replace('.fits', sfx)) for out in done for sfx in ('.fits', '.rms.fits', '.mask.fits'))
        return {'ms_per_subtraction': 1e3 * dt / max(len(done), 1), 'subtract_mpix_s': len(done) * npx / 1e6 / dt,
                'subtractions': len(done), 'of': len(imgs), 'lanes_x_fitbatch_x_filebatch': args.nightly_files,
                'files_in_per_subtraction': 3, 'files_out_per_subtraction': 4, 'bytes_in': bytes_in, 'bytes_out': bytes_out,
                'in_GBs': bytes_in / dt / 1e9, 'out_GBs': bytes_out / dt / 1e9,
                'readers': ring.nreaders, 'writers': ring.nwriters, 'page_cache': 'warm',
                'driver': 'scripts/donightly.py run_night: reads of batch b + 1 and writes of batch b - 1 under the '
                          'subtractions of batch b',
                'log_tail': log.getvalue().strip().splitlines()[-(14 if os.environ.get('ZM_NIGHT_TRACE') else 2):]}
    finally:
        if pool is not None:
            pool.close()
        if ring is not None:
            ring.close()
        shutil.rmtree(d, ignore_errors=True)


def secondary_clipped(args, z, dev, eng, base, dframes, local, timed, npx, full_step=None, stream=None):
    """The reference's DEFAULT operator in the headline's shadow: configs[1] with the science
    COMBINE_TYPE every from_images call runs unless told otherwise (CLIPPED 4.0 / 0.3,
    zuds/astromatic/makecoadd/default.swarp:24-31) - the resident-stack path: k_coadd_fused_dma in STACK mode
    (samples stored, not summed) + k_combine<32> - as a coadd leg and, with `full_step`, as the whole step
    (this coadd + the same subtraction against it), each of the two kernels with its own HBM roofline."""
    p = z.coadd_params(combine='CLIPPED', subtract_back=True, rescale_weights=True)
    # (on the stream of the headline's coadd: one more stream would change which streams share a hardware queue)
    co = dev.DeviceCoadd(base, p, device=local, engine=eng, want_mask=not args.no_mask, stream=stream)
    m = 0 if args.no_mask else 1
    pmc = pmc_profile(args)
    kern = (pmc.get('clipped') or {}).get('kernels') or {}
    co.run(dframes)
    out = {'combine': 'CLIPPED', 'what': 'COMBINE_TYPE CLIPPED, the reference default (default.swarp:24-31): resident stack'}
    scopes = {}
    for scope in ('coadd_fused', 'combine'):
        eng.timing(True, only=scope)
        eng.timing_reset()
        dt = timed(lambda: co.run(dframes), args.steps)
        eng.timing(False)
        scopes[scope] = eng.timing_read(scope)
    out['coadd_ms'] = 1e3 * dt / args.steps
    out['coadd_mpix_s'] = args.frames * npx / 1e6 * args.steps / dt
    ms, cnt = scopes['coadd_fused']
    if cnt:
        us = 1e3 * ms / cnt
        byt = (args.frames * (16 + 2 * m) + 4 * m) * npx   # 8 B + 2 B box-OR in per input px, 8 B {value, weight} out per px and frame
        kp = kern.get('k_coadd_fused') or kern.get('k_coadd_fused_dma')
        out['stack_roofline'] = {'bound': 'hbm', 'kernel': (kp or {}).get('name', 'k_coadd_fused_own / _dma') + ' (LANCZOS3, stack' + (', mask coadd)' if m else ')'),
                                 'avg_launch_us': us, 'algorithmic_bytes_per_launch': byt, 'achieved': byt / (us * 1e-6) / 1e9,
                                 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 'traffic': kp.get('hbm_bytes_per_launch') if kp else None,
                                 'wasted': kp['hbm_bytes_per_launch'] / byt if kp and kp.get('hbm_bytes_per_launch') else None,
                                 'valu_insts_per_px': kp.get('valu_insts_per_px') if kp else None}
    ms, cnt = scopes['combine']
    if cnt:
        us = 1e3 * ms / cnt
        byt = (8 * args.frames + 8) * npx          # one read of every sample + one write
        kp = kern.get('k_combine')
        out['combine_roofline'] = {'bound': 'hbm', 'kernel': f'k_combine<{args.frames}> CLIPPED', 'avg_launch_us': us,
                                   'algorithmic_bytes_per_launch': byt, 'achieved': byt / (us * 1e-6) / 1e9,
                                   'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   'frac_of_achievable': byt / (us * 1e-6) / 1e9 / HBM_ACHIEVABLE_GBS,
                                   'traffic': kp.get('hbm_bytes_per_launch') if kp else None,
                                   'wasted': kp['hbm_bytes_per_launch'] / byt if kp and kp.get('hbm_bytes_per_launch') else None}
        out['combine_kernel'] = {'avg_us': us, 'algorithmic_bytes': byt, 'achieved_GBs': byt / (us * 1e-6) / 1e9,
                                 'frac_of_hbm_peak': byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    out['counters_from'] = pmc.get('stale') or f'{pmc.get("file")} (kernel sources {pmc.get("kernel_sources_sha16")})'
    if full_step is not None:
        coadd_leg, sub_leg, sci = full_step

        def step_c():
            coadd_leg(co, dframes)
            sub_leg(co, sci)
        step_c()
        dts = timed(step_c, args.steps)
        out['ms_per_step'] = 1e3 * dts / args.steps
        out['value_mpix_s'] = (args.frames + 1) * npx / 1e6 * args.steps / dts
        out['step'] = 'the bench step with COMBINE_TYPE CLIPPED: mesh background + rescale + resample to the resident stack + ' \
                      'clipped combine (+ AND mask coadd), then the same subtraction against that coadd'
    del co
    # BASELINE configs[3]: what each of 8 ranks combines after the row-band exchange of a 256-frame stack -
    # all 256 samples of its 384 rows (k_combine_wide<4>); synthetic samples, 2 % of them without weight
    import ctypes as C
    import torch
    depth, rows = 8 * args.frames, args.size // 8
    if depth <= 512 and rows >= 1:
        g = torch.Generator(device='cuda')
        g.manual_seed(9)
        device = torch.device('cuda', local)
        stack = torch.empty((depth, rows, args.size, 2), dtype=torch.float32, device=device)
        stack[..., 0] = torch.randn((depth, rows, args.size), generator=g, device=device) * 5 + 100
        stack[..., 1] = torch.where(torch.rand((depth, rows, args.size), generator=g, device=device) < 0.02, 0.0, 0.04)
        o1 = torch.empty((rows, args.size), dtype=torch.float32, device=device)
        o2 = torch.empty_like(o1)
        bpx = rows * args.size

        def band():
            z._lib.check(eng.L.zm_combine_stack_dev(eng.ctx, depth, stack.data_ptr(), bpx, bpx, C.byref(p),
                                                    o1.data_ptr(), o2.data_ptr()))
        band()
        dtb = timed(band, args.steps)
        byt = (8 * depth + 8) * bpx
        out['band_combine'] = {'depth': depth, 'rows': rows, 'ms': 1e3 * dtb / args.steps,
                               'achieved_GBs': byt * args.steps / dtb / 1e9,
                               'frac_of_hbm_peak': byt * args.steps / dtb / 1e9 / HBM_PEAK_GBS,
                               'what': 'CLIPPED combine of one rank\'s row band of an 8-rank stack (every frame of the stack)'}
        del stack
    return out


def object_api_clock(z, d, files, sci_paths, args, nref=8):
    """The reference's own call sites on the FITS files of the step: what a user of
    scripts/makeref.py / dostack.py / dosub.py waits for - `ReferenceImage.from_images` of `nref` frames
    (zuds/coadd.py:25-236, defaults: COMBINE_TYPE CLIPPED, mesh background, rescale, mask coadd, bit 16,
    pedestal, seeing) and `SingleEpochSubtraction.from_images` of one science frame against it
    (zuds/subtraction.py:57-226), wall clock incl. every file read and written, on the device route of the
    object layer (objdev) and on the host-pointer route it replaced (ZM_OBJECT_API=host)."""
    nref = min(nref, len(files['sci']))

    def objects(paths_sci, paths_wgt, paths_msk):
        out = []
        for ps, pw, pm in zip(paths_sci, paths_wgt, paths_msk):
            im = z.ScienceImage.from_file(ps)
            im._weightimg = z.FITSImage.from_file(pw)
            im.mask_image = z.MaskImage.from_file(pm)
            out.append(im)
        return out
    res = {'frames': nref, 'size': args.size}
    for route in ('device', 'host'):
        old = os.environ.get('ZM_OBJECT_API')
        os.environ['ZM_OBJECT_API'] = route
        try:
            tc, ts = [], []
            for rep in range(2):                         # the faster of two (allocations, page cache)
                refname = os.path.join(d, f'ref_{route}{rep}.000651_c03_q1_zg.fits')
                ims = objects(files['sci'][:nref], files['wgt'][:nref], files['msk'][:nref])
                t0 = time.perf_counter()
                ref = z.ReferenceImage.from_images(ims, refname)
                t1 = time.perf_counter()
                sci = objects([sci_paths[0]], [sci_paths[1]], [sci_paths[2]])[0]
                t2 = time.perf_counter()
                sub = z.SingleEpochSubtraction.from_images(sci, ref, tmpdir=d)
                t3 = time.perf_counter()
                tc.append(t1 - t0)
                ts.append(t3 - t2)
                for sfx in ('.fits', '.rms.fits', '.mask.fits'):
                    os.remove(sub.local_path.replace('.fits', sfx))
            res[route] = {'reference_from_images_ms': 1e3 * min(tc), 'subtraction_from_images_ms': 1e3 * min(ts),
                          'coadd_mpix_s': nref * args.size ** 2 / 1e6 / min(tc), 'subtract_mpix_s': args.size ** 2 / 1e6 / min(ts)}
            # a science frame without a SEEING card (zuds/hotpants.py:38-44 measures it and goes on): round 6 keeps it on
            # the device route (stars and moments on the planes in HBM); the host route measures on host arrays
            tn = []
            for rep in range(2):
                sci = objects([sci_paths[0]], [sci_paths[1]], [sci_paths[2]])[0]
                sci.header.pop('SEEING', None)
                t2 = time.perf_counter()
                sub = z.SingleEpochSubtraction.from_images(sci, ref, tmpdir=d)
                tn.append(time.perf_counter() - t2)
                for sfx in ('.fits', '.rms.fits', '.mask.fits'):
                    os.remove(sub.local_path.replace('.fits', sfx))
            res[route]['subtraction_without_seeing_card_ms'] = 1e3 * min(tn)
            res[route]['measured_seeing_px'] = float(sub.header['SEEING'])
            if route == 'device':
                # ... and COLD: the frames as ZTF delivers them - science image + mask, no
                # .weight.fits / .rms.fits.  from_images asks every frame for its weight map (zuds/swarp.py:43-51),
                # dosub.py asks the science frame for its rms map first (scripts/dosub.py:35-47): mesh background,
                # 1 / rms^2 with the bad-bit and SATURATE rules, the derived maps written next to the inputs - all
                # on the device planes (objdev.derive_maps).  The derived files and the plane cache are dropped
                # between the repeats, so every repeat starts cold.
                objdev = importlib.import_module('zuds-pipeline_amd.objdev')

                def cold_objects(paths_sci, paths_msk):
                    out = []
                    for ps, pm in zip(paths_sci, paths_msk):
                        for sfx in ('.rms.fits', '.weight.fits'):
                            if os.path.exists(ps.replace('.fits', sfx)):
                                os.remove(ps.replace('.fits', sfx))
                        im = z.ScienceImage.from_file(ps)
                        im.mask_image = z.MaskImage.from_file(pm)
                        out.append(im)
                    return out
                tc, ts, td = [], [], []
                for rep in range(2):
                    objdev.get_io().cache_clear()
                    refname = os.path.join(d, f'ref_cold{rep}.000651_c03_q1_zg.fits')
                    ims = cold_objects(files['sci'][:nref], files['msk'][:nref])
                    t0 = time.perf_counter()
                    ref = z.ReferenceImage.from_images(ims, refname)
                    t1 = time.perf_counter()
                    sci = cold_objects([sci_paths[0]], [sci_paths[2]])[0]
                    t2 = time.perf_counter()
                    _ = sci.rms_image
                    t3 = time.perf_counter()
                    sub = z.SingleEpochSubtraction.from_images(sci, ref, tmpdir=d)
                    t4 = time.perf_counter()
                    tc.append(t1 - t0)
                    td.append(t3 - t2)
                    ts.append(t4 - t2)
                    for sfx in ('.fits', '.rms.fits', '.mask.fits'):
                        os.remove(sub.local_path.replace('.fits', sfx))
                cold_objects(files['sci'][:nref] + [sci_paths[0]], files['msk'][:nref] + [sci_paths[2]])   # (removes the maps)
                res['cold'] = {'reference_from_images_ms': 1e3 * min(tc), 'subtraction_from_images_ms': 1e3 * min(ts),
                               'of_which_rms_image_ms': 1e3 * min(td),
                               'what': 'device route, inputs without .weight.fits / .rms.fits: the maps are derived on the '
                                       'device (mesh background rms, 1 / rms^2) and written next to the inputs; the '
                                       'subtraction clock includes sci.rms_image (scripts/dosub.py:35-47)',
                               'h2d_bytes_per_cold_frame_px': 6, 'derived_files_written': 2 * nref + 1}
        finally:
            if old is None:
                os.environ.pop('ZM_OBJECT_API', None)
            else:
                os.environ['ZM_OBJECT_API'] = old
    return res


def data_movement_clocks(args, z, dev, eng, torch, base, frames, sci, coadd, sub, ref_rms, step, timed,
                         device_ms):
    """the same step on three clocks - inputs resident in HBM (the headline),
    + H2D of every input plane and D2H of the products over PCIe, + FITS files on local disk in
    and out (raw data blocks to the GPU, decode / encode kernels: device.FITSDeviceIO)."""
    import shutil
    import tempfile
    device = coadd.device
    clocks = {'device_ms': device_ms, 'workload': 'the bench step'}
    tools = None
    planes = [(f, k) for f in frames for k in ('img', 'wgt', 'mask')] + \
             [(sci, k) for k in ('img', 'wgt', 'mask', 'rms')]
    products = lambda: [coadd.img, coadd.wgt] + ([coadd.mask] if coadd.mask is not None else []) + \
        ([] if args.no_subtract else [sub.diff, sub.noise, sub.submask])
    native16 = sci['mask'].dtype == torch.int16              # int16 masks in HBM: no widening behind the copy
    try:
        # Host side: every input plane in pinned memory, masks as int16 (what a ZTF mask file holds:
        # half the bytes of the int32 the kernels read; widened on the device behind the copy).
        def host_plane(f, k):
            t = f[k].cpu()
            return (t.to(torch.int16) if k == 'mask' else t).pin_memory()
        pinned_in = [host_plane(f, k) for f, k in planes]
        pinned_out = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in products()]
        in_bytes = sum(t.numel() * t.element_size() for t in pinned_in)
        out_bytes = sum(t.numel() * t.element_size() for t in pinned_out)
        # Device side: TWO sets of input planes.  The copy stream fills set (k + 1) % 2 while the compute
        # stream works on set k % 2 - H2D of step k + 1 under the kernels of step k - and a third stream
        # returns the products of step k - 1 (PCIe is full duplex).  Events order the three.
        def clone_set():
            fr2 = [dict(f, img=torch.empty_like(f['img']), wgt=torch.empty_like(f['wgt']),
                        mask=torch.empty_like(f['mask'])) for f in frames]
            sc2 = dict(sci, img=torch.empty_like(sci['img']), wgt=torch.empty_like(sci['wgt']),
                       mask=torch.empty_like(sci['mask']), rms=torch.empty_like(sci['rms']))
            return fr2, sc2
        fr_b, sc_b = clone_set()
        sets = [dict(frames=frames, sci=sci, dfr=dev.DeviceFrames(frames, device)),
                dict(frames=fr_b, sci=sc_b, dfr=dev.DeviceFrames(fr_b, device))]
        for S in sets:
            S['planes'] = [(f, k) for f in S['frames'] for k in ('img', 'wgt', 'mask')] + \
                          [(S['sci'], k) for k in ('img', 'wgt', 'mask', 'rms')]
            S['m16'] = {id(f): torch.empty(f['mask'].shape, dtype=torch.int16, device=device)
                        for f in S['frames'] + [S['sci']]}
            S['free'] = None                         # event: the compute that read this set is done
        # (high-priority streams: the runtime gives them hardware queues of their own, so a copy never waits in a
        # queue behind a kernel of the step it is meant to overlap with)
        cs, ds = torch.cuda.Stream(device, priority=-1), torch.cuda.Stream(device, priority=-1)
        state = {'k': 0, 'd2h': None}

        def enqueue_copies(S):
            with torch.cuda.stream(cs):
                if S['free'] is not None:
                    cs.wait_event(S['free'])
                for (f, k), h in zip(S['planes'], pinned_in):
                    (S['m16'][id(f)] if (k == 'mask' and not native16) else f[k]).copy_(h, non_blocking=True)
                S['arrived'] = cs.record_event()

        def pcie_step():
            # (the engine calls below block the host - the subtraction reads back its rejection flags - so
            # the copies of the NEXT step are put on the copy stream first)
            S = sets[state['k'] % 2]
            state['k'] += 1
            enqueue_copies(sets[state['k'] % 2])
            with torch.cuda.stream(coadd.stream):
                coadd.stream.wait_event(S['arrived'])
                if not native16:
                    for f in S['frames'] + [S['sci']]:
                        f['mask'].copy_(S['m16'][id(f)])         # int16 -> int32 on the device
                if state['d2h'] is not None:
                    coadd.stream.wait_event(state['d2h'])        # the products of the previous step have left
            step(coadd, S['dfr'], S['sci'])
            with torch.cuda.stream(coadd.stream):
                S['free'] = coadd.stream.record_event()
            with torch.cuda.stream(ds):
                ds.wait_event(S['free'])
                for h, t in zip(pinned_out, products()):
                    h.copy_(t, non_blocking=True)
                state['d2h'] = ds.record_event()

        def copies_only():
            with torch.cuda.stream(cs):
                for (f, k), h in zip(sets[1]['planes'], pinned_in):
                    (sets[1]['m16'][id(f)] if (k == 'mask' and not native16) else f[k]).copy_(h, non_blocking=True)
        enqueue_copies(sets[0])
        pcie_step()
        pcie_step()
        nrep = 4
        dt = timed(pcie_step, nrep) / nrep
        copies_only()
        dt_copy = timed(copies_only, 2) / 2
        clocks['with_pcie_ms'] = 1e3 * dt
        clocks['pcie'] = {'h2d_bytes': in_bytes, 'd2h_bytes': out_bytes, 'host_memory': 'pinned',
                          'h2d_copy_alone_ms': 1e3 * dt_copy, 'h2d_GBs': in_bytes / dt_copy / 1e9,
                          'masks_over_pcie': 'int16, read as int16 by the kernels' if native16 else 'int16, widened on the device',
                          'overlap': 'H2D of step k + 1 (copy stream, second set of input planes) under the kernels '
                                     'of step k; D2H of the products of step k - 1 on a third stream; events '
                                     'between the three',
                          'steps_timed': nrep,
                          'ratio_to_max_of_copy_and_device': dt / max(dt_copy, device_ms * 1e-3)}
        del sets, fr_b, sc_b
        torch.cuda.empty_cache()
    except Exception as e:                                   # noqa: report, do not fail the bench
        clocks['with_pcie_ms'] = None
        clocks['pcie_error'] = repr(e)

    if os.environ.get('ZM_BENCH_PCIE_ONLY') == '1':          # developer (tools/pcie_bisect.sh): the PCIe clock alone
        return clocks, tools
    d = tempfile.mkdtemp(prefix='zmbench_', dir=os.environ.get('TMPDIR') or None)
    try:
        need = sum(t.numel() * t.element_size() for f, k in planes for t in [f[k]])
        if shutil.disk_usage(d).free < 2 * need:
            raise OSError(f'not enough free space under {d} for {need / 1e9:.1f} GB of FITS files')
        files = {'sci': [], 'wgt': [], 'msk': [], 'combine': args.combine}
        for i, f in enumerate(frames + [sci]):
            hdr = dict(f['wcs'].to_header(), NAXIS1=args.size, NAXIS2=args.size,
                       MAGZP=25.0 - 2.5 * float(np.log10(f['flxscale'])), SEEING=args.seeing,
                       OBSMJD=58800.0 + 0.25 * i, FIELDID=651, CCDID=3, QID=1, FILTERID=1)
            for lst, key, suf, cast in (('sci', 'img', 'sciimg', None), ('wgt', 'wgt', 'weight', None),
                                        ('msk', 'mask', 'mskimg', np.int16)):
                path = os.path.join(d, f'f{i:02d}.{suf}.fits')
                a = f[key].cpu().numpy()
                z.fits.write(path, a.astype(cast) if cast else a, hdr)
                files[lst].append(path)
        sci_paths = [files[k].pop() for k in ('sci', 'wgt', 'msk')]
        io = dev.FITSDeviceIO(device.index, engine=eng, stream=coadd.stream)
        big_rms = float(np.sqrt(50000.0))
        names = ['coadd.fits', 'coadd.weight.fits'] + (['coadd.mask.fits'] if coadd.mask is not None else []) + \
            ([] if args.no_subtract else ['sub.fits', 'sub.rms.fits', 'sub.mask.fits'])
        hdr_out = base.to_header()

        def fits_step_serial():
            # round 5's form, kept as the byte-for-byte yardstick and as `serial_ms`: read all, compute, write all
            dfr, _ = io.load_frames(files['sci'], files['wgt'], files['msk'])
            sc = dict(wcs=sci['wcs'])
            sc['img'], _ = io.load(sci_paths[0], 'f32')
            sc['wgt'], _ = io.load(sci_paths[1], 'f32')
            sc['mask'], _ = io.load(sci_paths[2], 'mask' if native16 else 'i32')
            sc['rms'] = torch.empty_like(sc['img'])
            with torch.cuda.stream(coadd.stream):
                z._lib.check(eng.L.zm_rms_from_weight_dev(eng.ctx, sc['wgt'].data_ptr(), None, sc['img'].numel(),
                                                          big_rms, sc['rms'].data_ptr()))
            step(coadd, dfr, sc)
            for nme, t in zip(names, products()):
                io.save(os.path.join(d, nme), t, hdr_out)
        fits_step_serial()                                   # page cache, allocations
        dt_serial = timed(fits_step_serial, 1)
        import hashlib

        def digest(path):
            with open(path, 'rb') as fh:
                return hashlib.sha256(fh.read()).hexdigest()
        serial_sha = {nme: digest(os.path.join(d, nme)) for nme in names}

        # The steady state of a night: the 99 files of step k + 1 are read by the ring's reader
        # threads into pinned buffers and sent + decoded on its copy stream while step k computes; the six products
        # of step k are encoded behind its kernels, copied back on a third stream and written by writer threads
        # while step k + 1 computes (zuds-pipeline_amd/fitsring.py).  N reads, N computes, N writes inside the clock;
        # the clock stops when the last file is on disk.
        ringmod = importlib.import_module('zuds-pipeline_amd.fitsring')
        ring = ringmod.FITSRing(device.index)
        sci_extra = [(sci_paths[0], 'f32'), (sci_paths[1], 'f32'), (sci_paths[2], 'mask' if native16 else 'i32')]
        outdirs = [os.path.join(d, f'out{i}') for i in range(3)]
        for od in outdirs:
            os.makedirs(od, exist_ok=True)
        st = {'k': 0, 'ticket': None}

        def prefetch():
            return ring.prefetch_frames(files['sci'], files['wgt'], files['msk'], extra=sci_extra)

        def fits_step():
            t, st['ticket'] = st['ticket'], prefetch()       # step k + 1's reads start before step k is enqueued
            dfr, _, extra = ring.frames(t, coadd.stream)
            sc = dict(wcs=sci['wcs'], img=extra[0][0], wgt=extra[1][0], mask=extra[2][0])
            eng.set_stream(coadd.stream.cuda_stream)
            with torch.cuda.stream(coadd.stream):
                sc['rms'] = torch.empty_like(sc['img'])
                z._lib.check(eng.L.zm_rms_from_weight_dev(eng.ctx, sc['wgt'].data_ptr(), None, sc['img'].numel(),
                                                          big_rms, sc['rms'].data_ptr()))
            step(coadd, dfr, sc)
            od = outdirs[st['k'] % len(outdirs)]
            st['k'] += 1
            for nme, tns in zip(names, products()):
                ring.save(os.path.join(od, nme), tns, hdr_out, engine=eng, stream=coadd.stream)

        nfits = 6
        try:
            st['ticket'] = prefetch()
            fits_step()
            fits_step()                                      # pinned rings, allocator pools
            ring.flush()
            sync_ = lambda: (ring.flush(), torch.cuda.synchronize(device))
            sync_()
            t0 = time.perf_counter()
            for _ in range(nfits):
                fits_step()
            sync_()
            dt = (time.perf_counter() - t0) / nfits
            st['ticket'].result()                            # (the read ahead of a step that is not run)
            same = all(digest(os.path.join(od, nme)) == serial_sha[nme] for od in outdirs for nme in names)
        finally:
            ring.close()
        clocks['with_fits_ms'] = 1e3 * dt
        clocks['fits'] = {'files_in': 3 * (len(frames) + 1), 'files_out': len(names),
                          'bytes_in': sum(os.path.getsize(p) for k in ('sci', 'wgt', 'msk') for p in files[k]) + sum(os.path.getsize(p) for p in sci_paths),
                          'bytes_out': sum(os.path.getsize(os.path.join(d, nme)) for nme in names),
                          'page_cache': 'warm (the inputs were written by this process just before)',
                          'decode': 'on the device (zm_fits_decode_dev)', 'steps_timed': nfits,
                          'readers': ring.nreaders, 'writers': ring.nwriters, 'host_cores': len(os.sched_getaffinity(0)),
                          'pipeline': 'reads + H2D + decode of step k + 1 and D2H + writes of step k - 1 under the kernels of '
                                      'step k (fitsring.FITSRing); the clock stops when the last product is on disk',
                          'serial_ms': 1e3 * dt_serial, 'serial_is': 'round 5: read all -> compute -> write all, one step',
                          'products_byte_identical_to_serial': bool(same),
                          'ratio_to_with_pcie': (1e3 * dt / clocks['with_pcie_ms']) if clocks.get('with_pcie_ms') else None}
        mpix = (len(frames) + (0 if args.no_subtract else 1)) * args.size * args.size / 1e6
        for k in ('device_ms', 'with_pcie_ms', 'with_fits_ms'):
            if clocks.get(k):
                clocks[k.replace('_ms', '_mpix_s')] = mpix / (clocks[k] * 1e-3)
        tools = reference_tools(d, files, {'coadd': z.fits.read(os.path.join(d, 'coadd.fits'))[0]})
        try:
            clocks['object_api_ms'] = object_api_clock(z, d, files, sci_paths, args)
        except Exception as e:                               # noqa: report, do not fail the bench
            clocks['object_api_error'] = repr(e)
    except Exception as e:                                   # noqa: report, do not fail the bench
        clocks['with_fits_ms'] = None
        clocks['fits_error'] = repr(e)
        tools = reference_tools(d, None, None)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # the step objects go back to the resident inputs
    eng.set_stream(coadd.stream.cuda_stream)
    return clocks, tools


if __name__ == '__main__':
    main()
