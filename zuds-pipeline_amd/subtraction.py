"""Difference-image products (``zuds/subtraction.py``): same classes, signatures,
naming (``sub.<new>_<ref>.fits`` + ``.rms.fits`` + ``.mask.fits``) and mask-bit
semantics; the SWarp / SExtractor / hotpants processes are libzudsmi calls."""
import os
import uuid
from pathlib import Path

import numpy as np

from .coadd import ScienceCoadd, _coadd_from_images
from .constants import APER_KEY
from .fitsfile import HasWCS
from .image import CalibratableImage, CalibratableImageBase, CalibratedImage, FITSImage
from .mask import MaskImage, MaskImageBase

__all__ = ['sub_name', 'Subtraction', 'SingleEpochSubtraction', 'MultiEpochSubtraction']


def sub_name(frame, template):
    """``sub.<frame[:-5]>_<template[:-5]>.fits`` in the science directory
    (``zuds/subtraction.py:25-37``)."""
    frame = f'{frame}'
    template = f'{template}'
    refp = os.path.basename(template)[:-5]
    newp = os.path.basename(frame)[:-5]
    outdir = os.path.dirname(frame)
    subp = '_'.join([newp, refp])
    return os.path.join(outdir, 'sub.%s.fits' % subp)


def _shallow(obj, cls, mapped=True):
    """In-memory stand-in for the reference's on-disk transaction copy: a new
    object of class ``cls`` sharing ``obj``'s arrays, with its OWN header dicts (the
    reference reloads the copy from disk: a SEEING measured on it never reaches the
    caller's object).  ``mapped=False``
    leaves the copy without a file: whatever derives products from it (check-images,
    rms maps, a measured SEEING) keeps them in memory, as the reference keeps them in
    the transaction directory it deletes (``zuds/subtraction.py:68-99,224``)."""
    new = cls()
    new.basename = obj.basename
    new.header = dict(obj.header or {})
    new.header_comments = dict(obj.header_comments or {})
    new.data = obj.data
    if mapped and obj.ismapped:
        new._path = obj.local_path
    for prop in ('field', 'ccdid', 'qid', 'fid'):
        if hasattr(obj, prop):
            setattr(new, prop, getattr(obj, prop))
    return new


def _subtraction_device(cls, sci, ref, final_out, outmask, nreg_side, subtract_new_back, hotpants_kws):
    """``Subtraction.from_images`` on device planes: the chain of ``device.DeviceSubtraction`` (align the
    reference + mask + rms, OR the masks, mesh background + pedestal, robust limits, kernel fit, convolve,
    subtract, bit 17 - ``zuds/subtraction.py:57-226``, ``zuds/hotpants.py:15-95``) fed from raw FITS blocks and
    writing each of the three products once with its final cards.  The same pixels and cards as the
    host-pointer route (``tests/test_object_route_gpu.py``)."""
    from . import _lib, objdev
    from .constants import BAD_SUM, BIG_RMS
    from .device import DeviceSubtraction
    from .hotpants import info_cards, warn_unsolved
    oio = objdev.get_io()
    torch, eng, L = oio.torch, oio.engine, oio.engine.L
    check = _lib.check
    # the science frame comes with a weight map, or with an rms map (the mesh BACKGROUND_RMS map `sci.rms_image`
    # makes for a frame that was delivered without either, scripts/dosub.py:35-47), or with both
    sci_img, sci_mask, sci_wgt, sci_rms0, ref_img, ref_mask, ref_wgt = oio.planes(
        [(sci, 'f32'), (sci.mask_image, 'mask'), (getattr(sci, '_weightimg', None), 'f32'),
         (getattr(sci, '_rmsimg', None), 'f32'),
         (ref, 'f32'), (ref.mask_image, 'mask'), (ref._weightimg, 'f32')])
    eng.set_stream(oio.stream.cuda_stream)

    def rms_of(img, wgt, mask, header):
        # CalibratableImageBase.rms_image (zuds/image.py:173-208): 1 / sqrt(w), BIG_RMS where the mask is bad
        # (and within 10 % of SATURATE)
        m32 = mask
        if mask.dtype == torch.int16:
            m32 = torch.empty(mask.shape, dtype=torch.int32, device=mask.device)
            check(L.zm_mask_widen_dev(eng.ctx, mask.data_ptr(), mask.numel(), m32.data_ptr()), 'widen')
        bad = torch.empty(mask.shape, dtype=torch.uint8, device=mask.device)
        rms = torch.empty_like(img)
        check(L.zm_mask_bad_dev(eng.ctx, m32.data_ptr(), None, BAD_SUM, m32.numel(), None, bad.data_ptr()), 'bpm')
        check(L.zm_rms_from_weight_dev(eng.ctx, wgt.data_ptr(), bad.data_ptr(), wgt.numel(), float(BIG_RMS),
                                       rms.data_ptr()), 'rms')
        # (zm_rms_from_weight_dev also gives BIG_RMS to a weight <= 0 on a pixel the mask does not flag; the
        # reference's numpy divides there - 1 / sqrt(0) = inf -, and so does the host route: ADVICE r4)
        rms = torch.where((wgt <= 0) & (bad == 0), 1.0 / torch.sqrt(wgt), rms)
        if 'SATURATE' in header:
            rms = torch.where(img >= 0.9 * float(header['SATURATE']), torch.full_like(rms, float(BIG_RMS)), rms)
        return rms, m32

    def weight_of(img, rms, mask, header):
        # CalibratableImageBase.weight_image (zuds/image.py:136-171) from an rms map: 1 / rms^2, 0 where the mask
        # is bad or the pixel is within 10 % of SATURATE - in memory only, as on the host route (the transaction
        # copy of the science frame is unmapped: nothing lands next to the caller's file)
        mt = _lib.MASKTYPE_I16 if mask.dtype == torch.int16 else _lib.MASKTYPE_I32
        ny, nx = img.shape
        bad = torch.empty(mask.shape, dtype=torch.uint8, device=mask.device)
        check(L.zm_false_weight_dev(eng.ctx, mask.data_ptr(), mt, BAD_SUM, 0, nx, ny, None, bad.data_ptr()), 'bpm')
        wgt = torch.empty_like(img)
        satur = 0.9 * float(header['SATURATE']) if 'SATURATE' in header else 0.0
        check(L.zm_weight_from_rms_dev(eng.ctx, rms.data_ptr(), bad.data_ptr(), img.data_ptr() if satur else None,
                                       satur, img.numel(), wgt.data_ptr()), 'weight')
        return wgt
    with torch.cuda.stream(oio.stream):
        if sci_rms0 is not None:
            sci_rms = sci_rms0
            if sci_wgt is None:
                sci_wgt = weight_of(sci_img, sci_rms, sci_mask, sci.header)
        else:
            sci_rms, _ = rms_of(sci_img, sci_wgt, sci_mask, sci.header)
        ref_rms, ref_m32 = rms_of(ref_img, ref_wgt, ref_mask, ref.header)
    sci_header = dict(sci.header)
    sci_comments = dict(sci.header_comments or {})
    if 'SEEING' not in sci_header:
        # zuds/hotpants.py:38-44: a frame without the card gets its seeing measured and the subtraction goes on
        # (round 6: on the planes already in HBM - the stars and moments of seeing.measure_seeing, the bad-pixel map
        # of `mask.boolean` - instead of falling back to the host-pointer route; the card lands on the
        # transaction's header, never on the caller's object or file, as there)
        from .seeing import measure_seeing_dev
        with torch.cuda.stream(oio.stream):
            m32 = sci_mask
            if sci_mask.dtype == torch.int16:
                m32 = torch.empty(sci_mask.shape, dtype=torch.int32, device=sci_mask.device)
                check(L.zm_mask_widen_dev(eng.ctx, sci_mask.data_ptr(), sci_mask.numel(), m32.data_ptr()), 'widen')
            bad = torch.empty(sci_mask.shape, dtype=torch.uint8, device=sci_mask.device)
            check(L.zm_mask_bad_dev(eng.ctx, m32.data_ptr(), None, BAD_SUM, m32.numel(), None, bad.data_ptr()), 'bpm')
            sat = sci_header.get('SATURATE')
            seeing, _ = measure_seeing_dev(sci_img, bad, float(sat) if sat else None, engine=eng)
        sci_header['SEEING'] = float(seeing)
        sci_comments['SEEING'] = 'FWHM of seeing in pixels (Goldstein)'
    chain = DeviceSubtraction(sci.wcs, ref.wcs, device=oio.device.index, engine=eng, stream=oio.stream, overlap=True)
    diff, noise, submask = chain.run(sci_img, sci_rms, sci_mask, sci_wgt, ref_img, ref_rms, ref_m32,
                                     seeing=float(sci_header['SEEING']), nreg_side=nreg_side,
                                     subtract_back=subtract_new_back, hotpants_kws=hotpants_kws,
                                     ref_flxscale=float((ref.header or {}).get('FLXSCALE', 1.0)))
    info = {k: getattr(chain.info, k) for k, _ in chain.info._fields_}
    warn_unsolved(info, final_out)
    # cards: what hotpants.HotpantsCall.run writes, re-read, plus what from_images adds before its save()
    shape = tuple(chain.shape)
    hdr0 = dict(sci_header)
    hdr0.update(info_cards(info))
    hdr, com = objdev.written_header(hdr0, {}, shape, -32)          # (HotpantsCall.run writes the cards without comments)
    mh0 = dict(sci.mask_image.header or {})
    mc0 = dict(sci.mask_image.header_comments or {})
    mh0['BIT17'] = 17
    mc0['BIT17'] = 'MASKED BY HOTPANTS (1e-30) / DG'
    mhdr, mcom = objdev.written_header(mh0, mc0, shape, 32)
    props = {}
    for prop in ('field', 'ccdid', 'qid', 'fid'):
        v = getattr(sci, prop, None)
        props[prop] = v
        if v is not None:
            hdr[prop.upper()] = v
            mhdr[prop.upper()] = v
    hdr['SEEING'] = sci_header['SEEING']
    com['SEEING'] = sci_comments.get('SEEING', '')
    if issubclass(cls, CalibratedImage):
        for key in ('MAGZP', APER_KEY):
            if key in sci.header:
                hdr[key] = sci.header[key]
                com[key] = (sci.header_comments or {}).get(key, '')
    oio.save_all([(final_out, diff, hdr, com), (final_out.replace('.fits', '.rms.fits'), noise, hdr0, {}),
                  (outmask, submask, mhdr, mcom)])
    sub = cls.from_file(final_out, load_others=False) \
        if issubclass(cls, CalibratableImage) else cls.from_file(final_out)
    finalsubmask = MaskImage.from_file(outmask)
    sub._rmsimg = FITSImage.from_file(final_out.replace('.fits', '.rms.fits'))
    for img in (sub, finalsubmask):
        for prop, v in props.items():
            setattr(img, prop, v)
    sub.mask_image = finalsubmask
    finalsubmask.parent_image = sub
    sub.reference_image = ref
    sub.target_image = sci
    sub.hotpants_info = info
    return sub


class Subtraction(HasWCS):

    reference_image = None
    target_image = None

    @property
    def mjd(self):
        return self.target_image.mjd

    @classmethod
    def from_images(cls, sci, ref, data_product=False, tmpdir='/tmp', **kwargs):
        """``zuds/subtraction.py:57-226``: align the reference (and its mask) to
        the science grid, OR the masks, subtract with hotpants' algorithm, flag
        bit 17 where the difference carries the fill value 1e-30."""
        from .hotpants import prepare_hotpants

        subtract_new_back = kwargs.get('subtract_back', True)
        nreg_side = kwargs.get('nreg_side', 3)
        hotpants_kws = kwargs.get('hotpants_kws', {})

        if not (hasattr(sci, '_rmsimg') or hasattr(sci, '_weightimg')):
            raise ValueError('Science image must have a weight map or '
                             'rms map defined prior to subtraction.')
        if getattr(sci, 'mask_image', None) is None or getattr(ref, 'mask_image', None) is None:
            raise ValueError('Science and reference images must have masks.')

        directory = Path(tmpdir) / uuid.uuid4().hex
        final_dir = os.path.dirname(sci.local_path)
        final_out = os.path.join(final_dir, os.path.basename(
            sub_name(sci.local_path, ref.local_path)))
        outmask = final_out.replace('.fits', '.mask.fits')

        from . import objdev
        if objdev.enabled() and hasattr(ref, '_weightimg'):
            # the device route (objdev): raw FITS blocks -> HBM -> DeviceSubtraction -> encoded products; round 5:
            # also for a science frame that carries an rms map instead of a weight map - the cold case of
            # scripts/dosub.py:35-47, whose map `sci.rms_image` has just made on the device (objdev.derive_maps).
            # Round 6: a science frame whose SEEING still has to be measured stays on this route too.
            return _subtraction_device(cls, sci, ref, final_out, outmask, nreg_side, subtract_new_back,
                                       hotpants_kws)

        # The reference works on transaction copies whose masks are plain
        # MaskImageBase objects (zuds/subtraction.py:94-99): aligning such a mask
        # uses COMBINE_TYPE OR but does not add bit 16 (zuds/swarp.py:184-191).
        transact_ref = _shallow(ref, ref.__class__)
        transact_ref.mask_image = _shallow(ref.mask_image, MaskImageBase)
        transact_ref._weightimg = ref.weight_image
        # the science frame too: prepare_hotpants derives a background-subtracted plane,
        # possibly an rms map and a SEEING from it; none of that may land next to the
        # caller's file
        transact_sci = _shallow(sci, CalibratableImageBase, mapped=False)
        transact_sci.mask_image = sci.mask_image
        for attr in ('_rmsimg', '_weightimg'):
            if hasattr(sci, attr):
                setattr(transact_sci, attr, getattr(sci, attr))

        # remapped ref and remapped ref mask on the science grid
        remapped_ref = transact_ref.aligned_to(transact_sci, tmpdir=tmpdir)
        remapped_refmask = remapped_ref.mask_image
        remapped_ref.parent_image = transact_ref

        # a totally new copy of the mask
        submask = MaskImageBase()
        submask.basename = os.path.basename(outmask)
        for prop in ('field', 'ccdid', 'qid', 'fid'):
            setattr(submask, prop, getattr(sci, prop, None))
        submask.map_to_local_file(outmask)
        badpix = remapped_refmask.data.astype(np.int32) | sci.mask_image.data.astype(np.int32)
        submask.data = badpix
        submask.header = dict(sci.mask_image.header or {})
        submask.header_comments = dict(sci.mask_image.header_comments or {})

        call = prepare_hotpants(transact_sci, remapped_ref, final_out, submask.boolean, directory,
                                tmpdir=tmpdir, nreg_side=nreg_side,
                                subtract_new_back=subtract_new_back,
                                hotpants_kws=hotpants_kws)
        try:
            os.rmdir(directory)
        except OSError:
            pass
        sd, _ = call.run()                       # writes final_out and its .rms.fits

        # flip bit 17 where hotpants masked the output (zuds/subtraction.py:167-177)
        hotbad = np.zeros_like(submask.data, dtype=np.int32)
        hotbad[sd == np.float32(1e-30)] = 2 ** 17
        submask.data = submask.data | hotbad
        submask.header['BIT17'] = 17
        submask.header_comments['BIT17'] = 'MASKED BY HOTPANTS (1e-30) / DG'
        submask.save()

        sub = cls.from_file(final_out, load_others=False) \
            if issubclass(cls, CalibratableImage) else cls.from_file(final_out)
        finalsubmask = MaskImage.from_file(outmask)
        sub._rmsimg = FITSImage.from_file(final_out.replace('.fits', '.rms.fits'))
        for img in (sub, finalsubmask):
            for prop in ('field', 'ccdid', 'qid', 'fid'):
                v = getattr(sci, prop, None)
                setattr(img, prop, v)
                if v is not None:
                    img.header[prop.upper()] = v
        sub.mask_image = finalsubmask
        finalsubmask.parent_image = sub
        sub.reference_image = ref
        sub.target_image = sci
        sub.hotpants_info = call.info
        # zuds/subtraction.py:208-209 reads the caller's header; a frame without the card got
        # its SEEING measured on the transaction copy (prepare_hotpants), which is what the
        # difference image was made with
        sub.header['SEEING'] = sci.header.get('SEEING', transact_sci.header['SEEING'])
        sub.header_comments['SEEING'] = (sci.header_comments or {}).get(
            'SEEING', transact_sci.header_comments.get('SEEING', ''))
        if isinstance(sub, CalibratedImage):
            for key in ('MAGZP', APER_KEY):
                if key in sci.header:
                    sub.header[key] = sci.header[key]
                    sub.header_comments[key] = (sci.header_comments or {}).get(key, '')
        sub.save()
        sub.mask_image.save()
        return sub


class SingleEpochSubtraction(Subtraction, CalibratedImage):
    """``zuds/subtraction.py:229-240``."""


class MultiEpochSubtraction(Subtraction, CalibratableImage):
    """A CLIPPED coadd of single-epoch subtractions (``zuds/subtraction.py:261-319``)."""

    input_images = None

    @classmethod
    def from_images(cls, sci, ref, data_product=False, tmpdir='/tmp', **kwargs):
        force_map_subs = kwargs.pop('force_map_subs', True)
        if not isinstance(sci, ScienceCoadd):
            raise TypeError(f'Input science image "{sci.basename}" must be '
                            f'an instance of ScienceCoadd, got {type(sci)}.')
        # without a database the single-epoch subtractions are handed over, or
        # looked up as attributes of the stack inputs
        images = kwargs.pop('single_epoch_subtractions', None)
        if images is None:
            images = [getattr(i, 'single_epoch_subtraction', None) for i in sci.input_images]
            images = [i for i in images if i is not None and i.reference_image is ref]
        if len(images) != len(sci.input_images):
            raise ValueError('Number of single-epoch subtractions != number'
                             f' of stack inputs. Stack inputs: '
                             f'{[i.basename for i in sci.input_images]}, '
                             f'Single-epoch subtractions: '
                             f'{[i.basename for i in images]}')
        outfile_name = sub_name(sci.local_path, ref.local_path)
        coadd = _coadd_from_images(cls, images, outfile_name, sci_swarp_kws=kwargs,
                                   mask_swarp_kws=kwargs, addbkg=False,
                                   calculate_seeing=False, tmpdir=tmpdir)
        coadd.reference_image = ref
        coadd.target_image = sci
        coadd.header['SEEING'] = sci.header['SEEING']
        coadd.save()
        return coadd
