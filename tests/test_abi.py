"""C-ABI surface and host-side (fp64, no GPU) entry points."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import grid as ogrid
from oracle import resample as oresample
from oracle.wcs import map_out_to_in
from util import pkg, synth, to_oracle_wcs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'zudsmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(zm_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    z = pkg()
    L = C.CDLL(str(z._lib.LIBPATH))
    names = header_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(L, name), f'{name} declared in zudsmi.h but not exported'
    # and the ctypes table binds every one of them
    bound = set(z._lib.exported_symbols())
    assert set(names) <= bound | {'zm_debug_lanczos3'}, set(names) - bound


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every POD struct of include/zudsmi.h as a C compiler sees them,
    against the ctypes mirrors the Python layer binds with (a probe compiled from the header
    itself, not hand-written constants)."""
    import os
    import subprocess
    z = pkg()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {'zm_wcs': z._lib.zm_wcs, 'zm_frame': z._lib.zm_frame, 'zm_dframe': z._lib.zm_dframe,
               'zm_coadd_params': z._lib.zm_coadd_params, 'zm_hp_params': z._lib.zm_hp_params,
               'zm_hp_info': z._lib.zm_hp_info, 'zm_mask_plan': z._lib.zm_mask_plan,
               'zm_sub_job': z._lib.zm_sub_job}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "zudsmi.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} . %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'probe.c'
    src.write_text('\n'.join(lines) + '\n')
    exe = tmp_path / 'probe'
    subprocess.check_call(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)])
    seen = 0
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        name, field, value = line.split()
        cls = structs[name]
        want = C.sizeof(cls) if field == '.' else getattr(cls, field).offset
        assert int(value) == want, (name, field, int(value), want)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())


def test_last_error_is_set_on_bad_arguments():
    z = pkg()
    L = z._lib.lib()
    rc = L.zm_autogrid(0, None, None)
    assert rc != 0
    assert b'zm_autogrid' in L.zm_last_error()


@pytest.mark.parametrize('tpv', [False, True])
def test_wcs_roundtrip_and_oracle_agreement(tpv):
    s = synth()
    w = s.ztf_wcs(3072, 3080, dx=4.25, dy=-9.5, rot_deg=0.07, tpv=tpv)
    ow = to_oracle_wcs(w)
    rng = np.random.default_rng(3)
    x = rng.uniform(-50, 3130, 200)
    y = rng.uniform(-50, 3130, 200)
    ra, dec = w.all_pix2world(x, y, 1)
    ora, odec = ow.pix2sky(x, y)
    np.testing.assert_allclose(ra, ora, rtol=0, atol=1e-11)
    np.testing.assert_allclose(dec, odec, rtol=0, atol=1e-11)
    x2, y2 = w.all_world2pix(ra, dec, 1)
    np.testing.assert_allclose(x2, x, rtol=0, atol=1e-7)
    np.testing.assert_allclose(y2, y, rtol=0, atol=1e-7)


def test_wcs_origin_convention():
    s = synth()
    w = s.tan_wcs(100, 80)
    ra1, dec1 = w.all_pix2world(10.0, 20.0, 1)
    ra0, dec0 = w.all_pix2world(9.0, 19.0, 0)
    assert ra1 == ra0 and dec1 == dec0
    # CRPIX maps to CRVAL
    ra, dec = w.all_pix2world(w.crpix[0], w.crpix[1], 1)
    np.testing.assert_allclose([ra, dec], w.crval, atol=1e-12)


def test_map_matches_oracle():
    z = pkg()
    s = synth()
    wout = s.ztf_wcs(3072, 3072, tpv=True)
    win = s.ztf_wcs(3072, 3072, dx=11.3, dy=-6.1, rot_deg=-0.09, tpv=True)
    rng = np.random.default_rng(5)
    xo = rng.uniform(1, 3072, 300)
    yo = rng.uniform(1, 3072, 300)
    xi = np.empty_like(xo)
    yi = np.empty_like(yo)
    a, b = z._lib.wcs_struct(wout), z._lib.wcs_struct(win)
    z._lib.check(z._lib.lib().zm_wcs_map(C.byref(a), C.byref(b), xo.size,
                                         xo.ctypes.data, yo.ctypes.data,
                                         xi.ctypes.data, yi.ctypes.data))
    oxi, oyi = map_out_to_in(to_oracle_wcs(wout), to_oracle_wcs(win), xo, yo)
    np.testing.assert_allclose(xi, oxi, rtol=0, atol=1e-8)
    np.testing.assert_allclose(yi, oyi, rtol=0, atol=1e-8)


def test_flux_scale_matches_oracle():
    z = pkg()
    s = synth()
    win = s.ztf_wcs(512, 512, rot_deg=0.3)
    wout = s.tan_wcs(600, 600, scale=3.0e-4)
    a, b = z._lib.wcs_struct(win), z._lib.wcs_struct(wout)
    v = C.c_double()
    z._lib.check(z._lib.lib().zm_flux_scale(C.byref(a), C.byref(b), 0.37, C.byref(v)))
    ref = oresample.flux_scale(to_oracle_wcs(win), to_oracle_wcs(wout), 0.37)
    assert abs(v.value - ref) < 1e-9 * ref
    # ratio of pixel areas: (3.0e-4)^2 / (~2.8125e-4)^2
    assert abs(v.value / 0.37 - (3.0e-4 / 2.8125e-4) ** 2) < 2e-3


def test_autogrid_matches_oracle_and_covers_inputs():
    z = pkg()
    s = synth()
    ws = [s.ztf_wcs(495, 495, dx=dx, dy=dy, rot_deg=r, tpv=False)
          for dx, dy, r in [(0, 0, 0.0), (24.4, -25.2, 0.05), (-10.0, 8.0, -0.1)]]
    n = len(ws)
    arr = (z._lib.zm_wcs * n)(*[z._lib.wcs_struct(w) for w in ws])
    out = z._lib.zm_wcs()
    z._lib.check(z._lib.lib().zm_autogrid(n, arr, C.byref(out)))
    ref = ogrid.autogrid([to_oracle_wcs(w) for w in ws])
    assert (out.naxis[0], out.naxis[1]) == ref.naxis
    np.testing.assert_allclose(list(out.crpix), ref.crpix, atol=1e-6)
    np.testing.assert_allclose(list(out.crval), ref.crval, atol=1e-10)
    np.testing.assert_allclose(list(out.cd), ref.cd.ravel(), atol=1e-15)
    # every input corner lands inside the output frame
    wo = z.WCS.from_struct(out)
    for w in ws:
        fp = w.calc_footprint()
        x, y = wo.all_world2pix(fp[:, 0], fp[:, 1], 1)
        assert x.min() >= 0.5 - 1e-6 and x.max() <= out.naxis[0] + 0.5 + 1e-6
        assert y.min() >= 0.5 - 1e-6 and y.max() <= out.naxis[1] + 0.5 + 1e-6
    # union of offset frames is larger than one frame, as in the reference's
    # test_stack (495 x 495 inputs -> 544 x 545, zuds/tests/suite/test_stack.py:26-27)
    assert out.naxis[0] > 495 and out.naxis[1] > 495


def test_lanczos3_taps_match_oracle():
    z = pkg()
    L = z._lib.lib()
    out = np.zeros(6, dtype=np.float32)
    worst = 0.0
    for d in np.concatenate([np.linspace(1e-5, 1 - 1e-5, 257), [1e-4, 0.5, 0.999]]):
        L.zm_debug_lanczos3(C.c_float(d), out.ctypes.data)
        ref = oresample.lanczos3_taps(np.float64(np.float32(d)))
        worst = max(worst, np.abs(out - ref).max())
        assert abs(out.sum() - 1.0) < 1e-6
    assert worst < 5e-7, worst


def test_native_band_bounds_equal_array_split_and_the_python_layer():
    """csrc/comm.hip's row bands against np.array_split (what zuds/mpi.py:52-60 shards job lists with)
    and parallel.band_bounds, ranks 1 .. 8 (and 64) x odd and even heights, fewer rows than ranks too."""
    import importlib
    z = pkg()
    par = importlib.import_module('zuds-pipeline_amd.parallel')
    L = z._lib.lib()
    for world in list(range(1, 9)) + [64]:
        for ny in (1, 2, 3, 5, 7, 8, 63, 64, 65, 383, 3072, 3079, 3080, 3081):
            b = (C.c_int32 * (world + 1))()
            assert L.zm_comm_band_bounds(ny, world, b) == 0
            want = np.cumsum([0] + [len(a) for a in np.array_split(np.arange(ny), world)])
            assert list(b) == list(want) == par.band_bounds(ny, world), (world, ny)


@pytest.mark.parametrize('kind', ['AND', 'OR'])
def test_native_mask_plan_replayed_for_every_rank_folds_like_one_process(kind):
    """The banded schedule zm_mask_reduce_dev issues to RCCL (csrc/comm.hip), replayed on the CPU from
    zm_comm_mask_plan of every rank: sends match the peers' receives element for element, slots never
    overlap, and fold + gather leaves every rank with the mask one process would compute.  The
    hardware has only ever run this layer with one rank; this covers its band arithmetic."""
    z = pkg()
    L = z._lib.lib()
    rng = np.random.default_rng(11)
    for world in (2, 3, 4, 5, 7, 8):
        for nx, ny in ((5, 3), (7, 13), (16, 64), (9, 67), (3, 2)):
            plans = []
            for r in range(world):
                P = z._lib.zm_mask_plan()
                assert L.zm_comm_mask_plan(nx, ny, world, r, C.byref(P)) == 0
                plans.append(P)
            part = rng.integers(0, 2 ** 20, (world, ny * nx)).astype(np.int32)
            part[rng.uniform(size=part.shape) < 0.3] = -1              # "no frame of this rank covers the pixel"
            # the exchange: rank r's Send(g) lands in rank g's Recv(r)
            recv = [np.full(world * P.band_px, -7, np.int32) for P in plans]
            for r, P in enumerate(plans):
                assert P.world == world and P.rank == r
                for g in range(world):
                    Q = plans[g]
                    assert P.send_cnt[g] == Q.my_px == Q.recv_cnt[r]
                    assert Q.recv_off[r] + Q.recv_cnt[r] <= (r + 1) * Q.band_px <= len(recv[g])
                    recv[g][Q.recv_off[r]:Q.recv_off[r] + Q.recv_cnt[r]] = part[r][P.send_off[g]:P.send_off[g] + P.send_cnt[g]]
                assert sum(P.send_cnt[g] for g in range(world)) == nx * ny
                assert all(P.send_off[g] + P.send_cnt[g] == P.send_off[g + 1] for g in range(world - 1))
            # the fold (k_mask_accum: -1 is the identity) and the all-gather through slots of the largest band
            gathered = np.full(world * plans[0].band_px, -7, np.int32)
            for g, Q in enumerate(plans):
                acc = np.full(Q.my_px, -1, np.int32)
                for r in range(world):
                    m = recv[g][Q.recv_off[r]:Q.recv_off[r] + Q.my_px]
                    both = (acc != -1) & (m != -1)
                    acc = np.where(both, (acc & m) if kind == 'AND' else (acc | m), np.where(acc == -1, m, acc))
                gathered[Q.gather_off[g]:Q.gather_off[g] + Q.my_px] = acc
            out = np.empty(nx * ny, np.int32)
            P = plans[0]
            for g in range(world):
                out[P.send_off[g]:P.send_off[g] + P.send_cnt[g]] = gathered[P.gather_off[g]:P.gather_off[g] + P.send_cnt[g]]
            want = np.full(nx * ny, -1, np.int32)
            for r in range(world):
                m = part[r]
                both = (want != -1) & (m != -1)
                want = np.where(both, (want & m) if kind == 'AND' else (want | m), np.where(want == -1, m, want))
            assert np.array_equal(out, want), (world, nx, ny)
    P = z._lib.zm_mask_plan()
    assert L.zm_comm_mask_plan(4, 4, 65, 0, C.byref(P)) != 0 and b'at most 64' in L.zm_last_error()
