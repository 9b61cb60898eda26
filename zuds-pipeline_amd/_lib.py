"""ctypes binding of libzudsmi.so (the C-ABI declared in include/zudsmi.h).

This is the only place the product touches native code.  There is no CPU
fallback: if the shared object is missing or a GPU call fails, the caller gets
an exception.
"""
import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIBPATH = HERE / 'lib' / 'libzudsmi.so'

NPV = 40

RESAMPLE = {'NEAREST': 0, 'BILINEAR': 1, 'LANCZOS3': 3}
COMBINE = {'WEIGHTED': 0, 'MEDIAN': 1, 'CLIPPED': 2, 'AVERAGE': 3}
MASKCOMB = {'AND': 0, 'OR': 1}


class ZMError(RuntimeError):
    """A libzudsmi call returned non-zero (message from zm_last_error)."""


class zm_wcs(C.Structure):
    _fields_ = [('crpix', C.c_double * 2), ('crval', C.c_double * 2),
                ('cd', C.c_double * 4), ('pv1', C.c_double * NPV),
                ('pv2', C.c_double * NPV), ('naxis', C.c_int32 * 2),
                ('flags', C.c_int32), ('pad_', C.c_int32)]


MASKTYPE_I32, MASKTYPE_I16 = 0, 1


class zm_frame(C.Structure):
    _fields_ = [('img', C.c_void_p), ('wgt', C.c_void_p), ('mask', C.c_void_p),
                ('wcs', zm_wcs), ('flxscale', C.c_double),
                ('mask_type', C.c_int32), ('pad_', C.c_int32)]


zm_dframe = zm_frame   # same layout; pointers are device addresses


class zm_coadd_params(C.Structure):
    _fields_ = [('combine', C.c_int32), ('mask_combine', C.c_int32),
                ('resample', C.c_int32), ('subtract_back', C.c_int32),
                ('back_size', C.c_int32), ('back_filtersize', C.c_int32),
                ('rescale_weights', C.c_int32), ('pad_', C.c_int32),
                ('clip_sigma', C.c_double), ('clip_ampfrac', C.c_double),
                ('weight_thresh', C.c_double)]


class zm_hp_params(C.Structure):
    _fields_ = [('tu', C.c_double), ('tl', C.c_double), ('iu', C.c_double),
                ('il', C.c_double), ('r', C.c_double), ('rss', C.c_double),
                ('fin', C.c_double), ('fi', C.c_double),
                ('nsx', C.c_int32), ('nsy', C.c_int32),
                ('nrx', C.c_int32), ('nry', C.c_int32),
                ('ko', C.c_int32), ('bgo', C.c_int32),
                ('nss', C.c_int32), ('normalize', C.c_int32),
                ('ft', C.c_double), ('ks', C.c_double),
                ('ngauss', C.c_int32), ('deg', C.c_int32 * 4),
                ('pad_', C.c_int32 * 3),
                ('sigma', C.c_double * 4),
                ('limits_dev', C.c_void_p), ('limits_nsigma', C.c_double),
                ('flag_mask_dev', C.c_void_p), ('flag_bit', C.c_int32), ('async_info', C.c_int32)]


class zm_sub_job(C.Structure):
    """One job of ``zm_subtract_batch_dev`` (include/zudsmi.h): device pointers + its hotpants flags."""
    _fields_ = [('sci', C.c_void_p), ('sci_rms', C.c_void_p), ('ref', C.c_void_p),
                ('ref_rms', C.c_void_p), ('bpm', C.c_void_p),
                ('params', C.POINTER(zm_hp_params)),
                ('out_diff', C.c_void_p), ('out_rms', C.c_void_p)]


class zm_hp_info(C.Structure):
    _fields_ = [('nstamps_total', C.c_int32), ('nstamps_used', C.c_int32),
                ('niter', C.c_int32), ('ncoeff', C.c_int32),
                ('kernel_sum', C.c_double), ('chi2', C.c_double),
                ('nmasked', C.c_int32), ('status', C.c_int32),
                ('nunsolved', C.c_int32), ('retries', C.c_int32)]


COMM_MAX_RANKS = 64


class zm_mask_plan(C.Structure):
    _fields_ = [('world', C.c_int32), ('rank', C.c_int32), ('band_px', C.c_int64), ('my_px', C.c_int64),
                ('send_off', C.c_int64 * COMM_MAX_RANKS), ('send_cnt', C.c_int64 * COMM_MAX_RANKS),
                ('recv_off', C.c_int64 * COMM_MAX_RANKS), ('recv_cnt', C.c_int64 * COMM_MAX_RANKS),
                ('gather_off', C.c_int64 * COMM_MAX_RANKS)]


HP_UNSOLVED, HP_TIMEOUT, HP_PENDING = 1, 2, 4          # zm_hp_info.status bits (include/zudsmi.h)


_P = C.c_void_p
_SIGS = {
    'zm_ctx_create': (C.c_int, [C.c_int, C.POINTER(_P)]),
    'zm_ctx_destroy': (C.c_int, [_P]),
    'zm_ctx_set_stream': (C.c_int, [_P, _P]),
    'zm_ctx_synchronize': (C.c_int, [_P]),
    'zm_ctx_create_on_stream': (C.c_int, [C.c_int, _P, C.POINTER(C.c_void_p)]),
    'zm_ctx_set_share': (C.c_int, [_P, C.c_int]),
    'zm_ctx_query': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    'zm_ctx_set_conventions': (C.c_int, [_P, C.c_int, C.c_int]),
    'zm_last_error': (C.c_char_p, []),
    'zm_version': (C.c_char_p, []),
    'zm_autogrid': (C.c_int, [C.c_int, C.POINTER(zm_wcs), C.POINTER(zm_wcs)]),
    'zm_wcs_pix2sky': (C.c_int, [C.POINTER(zm_wcs), C.c_int, _P, _P, _P, _P]),
    'zm_wcs_sky2pix': (C.c_int, [C.POINTER(zm_wcs), C.c_int, _P, _P, _P, _P]),
    'zm_wcs_map': (C.c_int, [C.POINTER(zm_wcs), C.POINTER(zm_wcs), C.c_int,
                             _P, _P, _P, _P]),
    'zm_flux_scale': (C.c_int, [C.POINTER(zm_wcs), C.POINTER(zm_wcs),
                                C.c_double, C.POINTER(C.c_double)]),
    'zm_resample': (C.c_int, [_P, _P, _P, _P, C.POINTER(zm_wcs),
                              C.POINTER(zm_wcs), C.c_int, C.c_double,
                              _P, _P, _P]),
    'zm_align_pair_dev': (C.c_int, [_P, _P, _P, _P, C.POINTER(zm_wcs), C.POINTER(zm_wcs), C.c_int, C.c_double,
                                    C.c_double, _P, _P, _P]),
    'zm_resample_dev': (C.c_int, [_P, _P, _P, _P, C.POINTER(zm_wcs),
                                  C.POINTER(zm_wcs), C.c_int, C.c_double,
                                  _P, _P, _P]),
    'zm_resample_i16': (C.c_int, [_P, _P, _P, _P, C.POINTER(zm_wcs),
                                  C.POINTER(zm_wcs), C.c_int, C.c_double,
                                  _P, _P, _P]),
    'zm_resample_i16_dev': (C.c_int, [_P, _P, _P, _P, C.POINTER(zm_wcs),
                                      C.POINTER(zm_wcs), C.c_int, C.c_double,
                                      _P, _P, _P]),
    'zm_mask_widen_dev': (C.c_int, [_P, _P, C.c_int64, _P]),
    'zm_coadd_params_default': (None, [C.POINTER(zm_coadd_params)]),
    'zm_coadd': (C.c_int, [_P, C.c_int, C.POINTER(zm_frame), C.POINTER(zm_wcs),
                           C.POINTER(zm_coadd_params), _P, _P, _P, _P]),
    'zm_coadd_dev': (C.c_int, [_P, C.c_int, C.POINTER(zm_frame),
                               C.POINTER(zm_wcs), C.POINTER(zm_coadd_params),
                               C.c_int, _P, _P, _P, _P]),
    'zm_coadd_finalize_dev': (C.c_int, [_P, _P, _P, C.c_int64]),
    'zm_resample_stack_dev': (C.c_int, [_P, C.c_int, C.POINTER(zm_frame),
                                        C.POINTER(zm_wcs),
                                        C.POINTER(zm_coadd_params), _P, _P]),
    'zm_combine_stack_dev': (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int64,
                                       C.POINTER(zm_coadd_params), _P, _P]),
    'zm_background': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                _P, _P, _P, _P]),
    'zm_background_dev': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int,
                                    C.c_int, _P, _P, _P, _P]),
    'zm_hp_params_default': (None, [C.POINTER(zm_hp_params)]),
    'zm_subtract': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                              C.POINTER(zm_hp_params), _P, _P,
                              C.POINTER(zm_hp_info)]),
    'zm_subtract_dev': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                  C.POINTER(zm_hp_params), _P, _P,
                                  C.POINTER(zm_hp_info)]),
    'zm_subtract_info': (C.c_int, [_P, C.POINTER(zm_hp_info)]),
    'zm_subtract_batch_dev': (C.c_int, [_P, C.c_int, C.POINTER(zm_sub_job), C.c_int, C.c_int,
                                        C.POINTER(zm_hp_info)]),
    'zm_subtract_batch': (C.c_int, [_P, C.c_int, C.POINTER(zm_sub_job), C.c_int, C.c_int,
                                    C.POINTER(zm_hp_info)]),
    'zm_median_mad': (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    'zm_median_mad_dev': (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
    'zm_find_stars': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                C.c_int, _P, _P, _P, C.POINTER(C.c_int)]),
    'zm_star_fwhm': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    'zm_find_stars_dev': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                    C.c_int, _P, _P, _P, C.POINTER(C.c_int)]),
    'zm_star_fwhm_dev': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    'zm_negpix_test': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_double, C.c_double, _P]),
    'zm_fits_decode_dev': (C.c_int, [_P, _P, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int, _P]),
    'zm_fits_encode_dev': (C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    'zm_mask_accum_dev': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int]),
    'zm_mask_finalize_dev': (C.c_int, [_P, _P, _P, C.c_int64]),
    'zm_comm_unique_id': (C.c_int, [_P]),
    'zm_comm_init': (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(C.c_void_p)]),
    'zm_comm_destroy': (C.c_int, [_P]),
    'zm_coadd_reduce_dev': (C.c_int, [_P, _P, _P, C.c_int64]),
    'zm_mask_reduce_dev': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    'zm_comm_band_bounds': (C.c_int, [C.c_int, C.c_int, _P]),
    'zm_comm_mask_plan': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(zm_mask_plan)]),
    'zm_median_mad2_dev': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_double)]),
    'zm_median_mad2_async_dev': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P]),
    'zm_rms_from_weight_dev': (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, _P]),
    'zm_weight_from_rms_dev': (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int64, _P]),
    'zm_false_weight_dev': (C.c_int, [_P, _P, C.c_int, C.c_int32, C.c_int, C.c_int, C.c_int, _P, _P]),
    'zm_mask_bad_dev': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P, _P]),
    'zm_mask_flag_dev': (C.c_int, [_P, _P, _P, C.c_float, C.c_int32, C.c_int64]),
    'zm_add_scalar_dev': (C.c_int, [_P, _P, C.c_float, C.c_int64]),
    'zm_copy_probe_dev': (C.c_int, [_P, _P, _P, C.c_int64]),
    'zm_aperture_photometry': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P,
                                         C.c_double, _P, _P, _P]),
    'zm_aperture_photometry_dev': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P,
                                             C.c_double, _P, _P, _P]),
    'zm_timing_enable': (C.c_int, [_P, C.c_int]),
    'zm_timing_filter': (C.c_int, [_P, C.c_char_p]),
    'zm_timing_reset': (C.c_int, [_P]),
    'zm_timing_read': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double),
                                 C.POINTER(C.c_int64)]),
    'zm_debug_lanczos3': (None, [C.c_float, _P]),
}

_lib = None


def exported_symbols():
    """Names every entry point of include/zudsmi.h must be found under."""
    return [k for k in _SIGS if k != 'zm_debug_lanczos3']


def _one_hip_runtime():
    """One HIP runtime per process, initialised in one order.

    PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAMEs as
    /opt/rocm).  If libzudsmi pulls in the system runtime first, a later ``import torch`` finds
    "No HIP GPUs"; pre-loading torch's copies fixes that, but a torch that initialises after
    libzudsmi has already run kernels was seen to hang on MI355X (2 of 3 runs).  The one
    arrangement that has been stable is torch first - so when torch is installed it is
    imported here, before libzudsmi is loaded (ZM_NO_TORCH=1 skips this for torch-free use;
    the device-resident classes of device.py / parallel.py need torch anyway)."""
    import importlib.util
    import os
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    if not os.environ.get('ZM_NO_TORCH'):
        try:
            import torch  # noqa: F401
            return
        except Exception:
            pass
    d = os.path.join(os.path.dirname(spec.origin), 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        p = os.path.join(d, name)
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load libzudsmi.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _one_hip_runtime()
    if not LIBPATH.exists():
        raise ZMError(f'{LIBPATH} is missing: run __graft_entry__.build() '
                      f'(python zuds-pipeline_amd/build.py). There is no CPU '
                      f'fallback for this path.')
    L = C.CDLL(str(LIBPATH))
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise ZMError(f'{LIBPATH} does not export {name}; rebuild it '
                          f'(python zuds-pipeline_amd/build.py --force)')
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what=''):
    if rc != 0:
        msg = lib().zm_last_error().decode('utf-8', 'replace')
        raise ZMError(f'{what}: {msg} (rc={rc})' if what else f'{msg} (rc={rc})')


def ptr(a):
    """Raw address of a numpy array (or int / None passthrough)."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    return a.ctypes.data


def as_f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def as_i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def as_mask(a):
    """A mask plane for zm_frame.mask: an int16 array (a ZTF mask as read from its BITPIX 16 file) is
    handed over as it is (MASKTYPE_I16, half the bytes over PCIe and from HBM); anything else as int32.
    Returns (array or None, mask_type)."""
    if a is None:
        return None, MASKTYPE_I32
    a = np.asarray(a)
    if a.dtype == np.int16:
        return np.ascontiguousarray(a), MASKTYPE_I16
    return np.ascontiguousarray(a, dtype=np.int32), MASKTYPE_I32


def wcs_struct(w):
    """zm_wcs from a WCS object (attributes crpix, crval, cd, pv1, pv2, naxis,
    has_pv) or from a FITS header dict."""
    from .wcs import WCS
    if isinstance(w, zm_wcs):
        return w
    if isinstance(w, dict):
        w = WCS.from_header(w)
    s = zm_wcs()
    s.crpix[0], s.crpix[1] = float(w.crpix[0]), float(w.crpix[1])
    s.crval[0], s.crval[1] = float(w.crval[0]), float(w.crval[1])
    cd = np.asarray(w.cd, dtype=np.float64).ravel()
    for i in range(4):
        s.cd[i] = float(cd[i])
    for k in range(NPV):
        s.pv1[k] = float(w.pv1[k])
        s.pv2[k] = float(w.pv2[k])
    s.naxis[0], s.naxis[1] = int(w.naxis[0]), int(w.naxis[1])
    s.flags = 1 if w.has_pv else 0
    return s
